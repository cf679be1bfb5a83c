#!/bin/bash
# DEV TOOL: collect rocprofv3 PMC passes for one kprof configuration.  usage: tools/pmc_run.sh <tag> [kprof args...]
# (counters in separate passes, never combined with tracing -- see the gpurun rules)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT"
P4="FETCH_SIZE TCC_HIT_sum"
P5="WRITE_SIZE TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/kprof.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stft_db_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("counter means per dispatch of stft_db_kernel:")
for k,v in sorted(agg.items()):
    print(f"  {k:24s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
