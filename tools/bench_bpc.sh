for b in 1 2 8; do timeout -k 10 300 python bench.py --no-cpu-baseline --blocks-per-cu $b 2>&1 | tail -1 > /tmp/b.json; python3 -c "
import json; j=json.load(open('/tmp/b.json')); print('bpc', $b, j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['concurrent_frac'])"; done
