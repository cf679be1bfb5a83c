#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c13; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
run() { echo "--- $1 ($2 streams $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 2 --reps 200 --rounds 3 $V/libjsg_r02.so $CUR 2>&1 | grep -E "us/launch|vs first" | cut -c1-200; }
run "X=1" c2 4
run "X=1" c3 2
run "X=1" c5 3
for c in c2 c3 c5; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-boundary > $O/bench_$c.json 2> $O/bench_$c.err; python - <<PY
import json; l=json.loads(open("$O/bench_$c.json").read().strip().splitlines()[-1]); print("$c value %.4g region_frac %.4f inorder_us %.3f frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"], l["roofline"]["frac"]), l["parity"]["kernel"])
PY
done
