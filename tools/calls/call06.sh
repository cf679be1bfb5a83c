#!/bin/bash
# round 3, call 6: single-kernel STFT -> ARGB (OUTK = 2): tests + C5 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/c06; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
run() { echo "--- $1 ($2 $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 1 --reps 200 --rounds 3 $CUR 2>&1 | grep -E "us/launch|vs first|fused" | cut -c1-220; }
run "X=1" c5 1
run "JSG_IMAGE_TWO_KERNELS=1" c5 1

for s in 1 2 3 4; do python bench.py --config c5 --streams $s --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_c5_s$s.json 2> $O/bench_c5_s$s.err; python - <<PY
import json; l=json.loads(open("$O/bench_c5_s$s.json").read().strip().splitlines()[-1]); print("c5 streams $s value %.4g us/img %.2f region_frac %.4f inorder_us %.3f" % (l["value"], 1875e6/l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"])); print(json.dumps(l.get("parity"))[:600])
PY
done
JSG_IMAGE_TWO_KERNELS=1 python bench.py --config c5 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_c5_two.json 2> $O/bench_c5_two.err; python - <<PY
import json; l=json.loads(open("$O/bench_c5_two.json").read().strip().splitlines()[-1]); print("c5 two-kernel value %.4g us/img %.2f inorder_us %.3f" % (l["value"], 1875e6/l["value"], l["roofline"]["avg_launch_us"]))
PY
