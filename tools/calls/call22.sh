#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c22; mkdir -p $O
for q in 4 6 8 12 16; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-boundary > $O/b.json 2> $O/b.err; python - <<PY
import json; l=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); print("pool, GPU_MAX_HW_QUEUES=$q value %.4g region_frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"]))
PY
done
for q in 4 8; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-boundary --streams 4 > $O/b.json 2> $O/b.err; python - <<PY
import json; l=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); print("caller streams 4, GPU_MAX_HW_QUEUES=$q value %.4g region_frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"]))
PY
done
