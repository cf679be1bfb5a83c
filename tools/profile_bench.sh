#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for one bench.py configuration.
#   usage: tools/profile_bench.sh <tag> [config]          e.g.  tools/profile_bench.sh r02 c2
#   1. --kernel-trace --stats of the bench command           -> gpurun_out/prof_bench_<tag>_<cfg>/stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (never combined with tracing)
#   3. one SQ counter pass
# Summaries are post-processed by tools/profile_summarize.py into gpurun_out/profiles_<tag>/ (copy them to profiles/).
set -u
TAG=${1:-r06}; CFG=${2:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench_${TAG}_${CFG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# The timed region of bench.py IS a sequence of back-to-back dispatches of one kernel on one stream, host-issued by plain C calls: the
# tracer sees exactly what the bench times.  The side legs (parity, calibration, one-batch-per-dispatch, child runs) are switched off so
# that the trace holds the main kernel only; otherwise this is the driver's command.
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-boundary --no-parity --no-extra --no-calibration --no-single --no-power"
# counter passes: the same dispatches, fewer of them (the collector serialises every dispatch); counters are per-dispatch means
PMC_CMD="python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 3 --warmup 1 --dispatches-per-step 2 --no-cpu-baseline --no-boundary --no-parity --no-extra --no-calibration --no-single --no-power"
echo "$CMD" > $OUT/command.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $PMC_CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $PMC_CMD > $OUT/write.log 2>&1 || echo "write pass failed"
# SQ counters: three passes of at most eight (the block has eight slots), never combined with tracing
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- $PMC_CMD > $OUT/sq.log 2>&1 || echo "sq pass failed"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/sq2 -- $PMC_CMD > $OUT/sq2.log 2>&1 || echo "sq2 pass failed"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_IFETCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq3 -- $PMC_CMD > $OUT/sq3.log 2>&1 || echo "sq3 pass failed"
grep -h '"metric"' $OUT/stats.log | head -1 > $OUT/bench_lines.jsonl
python3 $GRAFT_REPO_ROOT/tools/profile_summarize.py $OUT $TAG $CFG
# the raw per-dispatch files are large (tens of MB per configuration; gpurun copies back at most 64 MiB): keep the summaries
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
