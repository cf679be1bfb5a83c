#!/bin/bash
# round 3, call 1: baseline state on this box + overlap trace + nbuf sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/c01; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log
AB=tools/variants/abbench; LIB=jadespectrogram_amd/libjsg.so
for cfg in c2 c3 c5 x4096; do timeout -k 10 120 $AB --cfg $cfg --reps 200 --rounds 3 $LIB >> $O/ab_base.log 2>&1; done
grep -E "==|us/launch" $O/ab_base.log | cut -c1-220
timeout -k 10 120 $AB --cfg c5 --streams 3 --reps 200 --rounds 3 $LIB > $O/ab_c5_s3.log 2>&1; grep "us/launch" $O/ab_c5_s3.log | cut -c1-220
for nb in 20 40 80; do python bench.py --nbuf $nb --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_nbuf$nb.json 2> $O/bench_c2_nbuf$nb.err; python - <<PY
import json; l=json.loads(open("$O/bench_c2_nbuf$nb.json").read().strip().splitlines()[-1]); print("nbuf $nb value %.4g region_frac %.4f inorder_us %.3f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"]))
PY
done
tools/profile_overlap.sh r03 c2 4 > $O/overlap.log 2>&1; tail -40 $O/overlap.log
python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
for c in ("c3","c5"):
    l=json.loads(open("$O/bench_%s.json"%c).read().strip().splitlines()[-1]); print(c, "value %.4g region_frac %.4f inorder_us %.3f frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"], l["roofline"]["frac"]))
PY
