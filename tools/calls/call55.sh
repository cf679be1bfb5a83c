#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c55; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_colormap.py tests/test_gpu_parity.py -m gpu -x -q -k "image or fused or colo or strided" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
tools/profile_bench.sh r03x c5 > $O/prof_c5.log 2>&1; grep -E "failed" $O/prof_c5.log
python bench.py --config c5 --no-cpu-baseline --no-boundary > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c5 --images-per-launch 1 --no-cpu-baseline --no-boundary > $O/bench_c5_single_3streams.json 2>> $O/bench_c5.err
python bench.py --config c5 --images-per-launch 1 --streams 1 --no-cpu-baseline --no-boundary > $O/bench_c5_single_inorder.json 2>> $O/bench_c5.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    l=json.loads(open(f).read().strip().splitlines()[-1]); r=l["roofline"]
    print(f.split("/")[-1], "value %.4g ms/step %.3f region %.4f inorder_us %.3f events %.4f" % (l["value"], l["ms_per_step"], r["timed_region_frac_of_8p0"], r["avg_launch_us"], r["frac_event_timed"]), l["parity"].get("strided_batch_pixels_differing_from_single_launches"), l["parity"]["fused_image_pixels_differing_from_two_kernel_image"])
h=json.load(open('gpurun_out/profiles_r03x/r03x_c5_hbm_traffic.json')); print('  ', round(h['avg_us'],3), round(h['traffic_over_algorithmic'],4), round(h['frac_of_8p0_from_trace_avg'],4), h['images_per_launch'])
PY
