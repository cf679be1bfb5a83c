#!/bin/bash
# Second data point: the same PMC pass on graph replays at the full launch count (1024 launches per step).
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c18; mkdir -p $O
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config c2 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-boundary"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES --output-format csv -d $O/pmc -- $CMD > $O/pmc.log 2>&1; echo "rc=$?"
grep -v "^W2026\|^E2026.*Opened" $O/pmc.log | tail -30 | cut -c1-300
wc -l $O/pmc/*/*counter_collection.csv 2>/dev/null
find $O -name '*counter_collection.csv' -delete
