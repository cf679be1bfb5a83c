B="--no-extra --no-boundary --no-calibration --no-parity --no-cpu-baseline --no-single"
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bare    ', round(j['roofline']['frac'],4), round(j['config']['prewarm_s'],2))"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('full    ', round(j['roofline']['frac'],4))"
  python bench.py --steps 20 --warmup 5 --no-extra --no-boundary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('no-child', round(j['roofline']['frac'],4))"
done
