#!/bin/bash
# final check of the round-3 tree on a fresh box: what the driver runs (GPU tests, smoke, default bench line)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c63; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); r=l["roofline"]
print("value %.4g %s frac %.4f (%s) traffic %.4g region %.4f cpu %s boundary p50 %s" % (l["value"], l["unit"], r["frac"], r["frac_source"][:20], r["traffic"], r["timed_region_frac_of_8p0"], l["cpu_baseline"]["value"], l["boundary"]["process_block_latency"]["p50_us"]))
print(r["traffic_source"]["matches_this_build"], l["parity"]["kernel"], l["parity"]["fused_image_pixels_differing_from_two_kernel_image"])
PY
