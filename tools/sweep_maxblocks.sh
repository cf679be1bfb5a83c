# DEV TOOL: bench.py with the grid capped at JSG_STFT_MAX_BLOCKS workgroups (more FFTs per wave, fewer waves per launch)
for mb in 0 512 256 192 128 96 64; do
  JSG_STFT_MAX_BLOCKS=$mb timeout -k 10 300 python bench.py --no-cpu-baseline --blocks-per-cu 8 2>&1 | tail -1 > /tmp/b.json
  python3 -c "
import json; j=json.load(open('/tmp/b.json')); print('max_blocks', $mb, '%.4g' % j['value'], 'in-order us', round(j['roofline']['avg_launch_us'],3), 'conc', round(j['roofline']['concurrent_frac'],3))"
done
