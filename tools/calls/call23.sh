#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c23; mkdir -p $O
for q in 8 16 32; do for cfg in "c2 --streams 4" "c3" "c3 --streams 2" "c5"; do GPU_MAX_HW_QUEUES=$q python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-boundary > $O/b.json 2> $O/b.err; python - <<PY
import json; l=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); print("GPU_MAX_HW_QUEUES=$q $cfg value %.4g region_frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"]))
PY
done; done
