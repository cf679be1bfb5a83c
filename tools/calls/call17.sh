#!/bin/bash
# One controlled reproduction of "rocprofv3 counter passes crash on hipGraph replays" (ADVICE r2): small launch count, log kept.
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c17; mkdir -p $O
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config c2 --streams 1 --steps 2 --warmup 1 --launches-per-step 32 --no-cpu-baseline --no-boundary"
echo "== kernel trace of a graph-replay run"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "rc=$?"
grep -c stft_db_kernel $O/trace/*/*kernel_trace.csv 2>/dev/null | head -3
echo "== PMC pass of a graph-replay run"; timeout -k 10 200 rocprofv3 --pmc SQ_WAVES --output-format csv -d $O/pmc -- $CMD > $O/pmc.log 2>&1; echo "rc=$?"
grep -v "^W2026\|^E2026.*Opened" $O/pmc.log | tail -40 | cut -c1-400
ls $O/pmc/*/ 2>/dev/null | head
