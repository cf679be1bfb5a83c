#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c21; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; python - <<PY
import json; l=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1]); print("c2 value %.4g region_frac %.4f inorder_us %.3f frac %.4f (%s)" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"], l["roofline"]["frac"], l["roofline"]["frac_source"][:40])); print(l["config"]["same_region_default_environment"]); print(l["config"]["issue"][:200]); print(l["roofline"]["traffic_source"])
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary --streams 4 > $O/bench_c2s4.json 2> $O/bench_c2s4.err; python - <<PY
import json; l=json.loads(open("$O/bench_c2s4.json").read().strip().splitlines()[-1]); print("c2 --streams 4 (caller streams) value %.4g region_frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"]))
PY
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary > $O/bench_c3.json 2> $O/bench_c3.err; python - <<PY
import json; l=json.loads(open("$O/bench_c3.json").read().strip().splitlines()[-1]); print("c3 value %.4g region_frac %.4f inorder_us %.3f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"]), l["roofline"]["second_roof"])
PY
