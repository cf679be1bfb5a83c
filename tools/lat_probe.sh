cd /tmp && export TMPDIR=/tmp
for fr in 8 512 4096; do
  rm -rf /tmp/lp; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python3 $GRAFT_REPO_ROOT/tools/kprof.py --frames $fr --steps 50 --nbuf 20 > /tmp/lp.log 2>&1
  echo "frames=$fr"; grep stft_db $(find /tmp/lp -name "*kernel_stats.csv") | sed "s/.*StftKArgs)\",//"
done
