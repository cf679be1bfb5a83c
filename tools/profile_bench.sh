#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for bench.py's headline kernel.
#   1. --kernel-trace --stats of the bench command           -> gpurun_out/prof_bench/stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (never combined with tracing)
# Summaries are post-processed by tools/profile_summarize.py into profiles/.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --streams 1: every dispatch runs alone, so rocprofv3's per-dispatch duration is the per-kernel number that
# bench.py's `roofline` (its in-order pass) reports; the default multi-stream timed region overlaps dispatches
CMD="python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 5000 --warmup 500 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || echo "write pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1 || echo "sq pass failed"
grep -h '"metric"' $OUT/*.log | head -4
python3 $GRAFT_REPO_ROOT/tools/profile_summarize.py $OUT $TAG
