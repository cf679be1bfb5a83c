#!/bin/bash
# DEV TOOL (GPU box): whole-job rate of bench.py's c2 region over streams x issue threads (x workgroups per CU).
#   usage: tools/sweep_overlap.sh "<streams...>" "<threads...>" "<bpc...>"
cd ${GRAFT_REPO_ROOT:-.}
for s in ${1:-4 8 12}; do for t in ${2:-2 3 4}; do for b in ${3:-1 2}; do
  python bench.py --streams $s --issue-threads $t --blocks-per-cu $b --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $s (used %d) threads $t bpc $b  value %.4g  region_frac %.4f' % (l['config']['hip_streams_per_gpu'], l['value'], l['roofline']['timed_region_frac_of_8p0']))"
done; done; done
