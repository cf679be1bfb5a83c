#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c24; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; python - <<PY
import json; l=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1]); print("c2 value %.4g region_frac %.4f inorder_us %.3f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"])); print(l["config"]["same_region_default_environment"])
PY
