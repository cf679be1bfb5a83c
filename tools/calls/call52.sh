#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c52; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bench_cli.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
python bench.py --config c5 --no-cpu-baseline --no-boundary > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc=$?"; tail -3 $O/bench_c5.err
python bench.py --config c5 --images-per-launch 1 --no-cpu-baseline --no-boundary > $O/bench_c5_single.json 2>> $O/bench_c5.err; echo "rc=$?"
python - <<PY
import json
for f in ("bench_c5.json","bench_c5_single.json"):
    l=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); r=l["roofline"]
    print(f, "value %.4g ms/step %.3f region %.4f inorder_us %.3f frac %.4f events %.4f ipl %s" % (l["value"], l["ms_per_step"], r["timed_region_frac_of_8p0"], r["avg_launch_us"], r["frac"], r["frac_event_timed"], l["config"]["images_per_launch"]), l["parity"].get("strided_batch_pixels_differing_from_single_launches"), l["config"]["step"])
PY
