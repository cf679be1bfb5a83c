#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of bench.py's OVERLAPPED launch mode (the mode that produces `value`).
#   usage: tools/profile_overlap.sh <tag> [config] [streams]        e.g.  tools/profile_overlap.sh r03 c2 4
# Host-issued launches on S streams, every stream held by a gate kernel while the host enqueues a step (bench.py --gate), so the
# traced dispatches overlap on the GPU as they do when the host keeps up.  Kernel trace only -- no counters in this pass.
set -u
TAG=${1:-r03}; CFG=${2:-c2}; S=${3:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_overlap_${TAG}_${CFG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --streams $S --no-graph --gate --steps 10 --warmup 2 --no-cpu-baseline --no-boundary --no-parity"
echo "$CMD" > $OUT/command.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/stats.log; }
grep -h '"metric"' $OUT/stats.log | head -1 > $OUT/bench_lines.jsonl
python3 $GRAFT_REPO_ROOT/tools/overlap_summarize.py $OUT $TAG $CFG
find $OUT -name '*kernel_trace.csv' -delete
