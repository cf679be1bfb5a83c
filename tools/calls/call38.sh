#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c38; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-250
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $V/libjsg_r02.so $CUR 2>&1 | grep -E "us/launch|fused" | cut -c1-220
for s in 1 3; do python bench.py --config c5 --streams $s --no-cpu-baseline --no-boundary --steps 20 --warmup 5 > $O/bench_c5_s$s.json 2> $O/bench_c5_s$s.err; python - <<PY
import json; l=json.loads(open("$O/bench_c5_s$s.json").read().strip().splitlines()[-1]); print("c5 streams $s value %.4g us/img %.2f region_frac %.4f inorder_us %.3f fused_diff %s" % (l["value"], 1875e6/l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"], l["parity"]["fused_image_pixels_differing_from_two_kernel_image"]))
PY
done
