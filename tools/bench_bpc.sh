# DEV TOOL: bench.py across --blocks-per-cu values, interleaved repetitions (REPS, default 3)
for r in $(seq ${REPS:-3}); do for b in 1 2 3 8; do timeout -k 10 300 python bench.py --no-cpu-baseline --blocks-per-cu $b 2>&1 | tail -1 > /tmp/b.json; python3 -c "
import json; j=json.load(open('/tmp/b.json')); print('rep', $r, 'bpc', $b, '%.4g' % j['value'], 'in-order us', round(j['roofline']['avg_launch_us'],3), 'conc', round(j['roofline']['concurrent_frac'],3))"; done; done
