#!/bin/bash
# DEV TOOL (run on the GPU box through gpurun): rocprofv3 PMC passes + kernel trace of one tools/abbench configuration.
#   usage: tools/pmc_ab.sh <tag> <cfg> <lib.so> [passes]     e.g.  tools/pmc_ab.sh r02_c3 c3 jadespectrogram_amd/libjsg.so "1 2 3 4 5 T"
# Counters go in separate passes and are never combined with tracing (gpurun rule); the program after `--` is the
# abbench binary itself (no shell / env / python hop between the profiler and the process that touches the GPU).
set -u
TAG=$1; CFG=$2; LIB=$(readlink -f $3); PASSES=${4:-"1 2 3 4 5 T"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
AB=$ROOT/tools/variants/abbench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AB_EAGER=1
declare -A SETS
SETS[1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SETS[2]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
SETS[3]="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT"
SETS[4]="FETCH_SIZE TCC_HIT_sum"
SETS[5]="WRITE_SIZE TCC_MISS_sum"
SETS[6]="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES"
for p in $PASSES; do
  if [ "$p" = T ]; then
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $AB --cfg $CFG --reps 400 --rounds 3 $LIB > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; }
  else
    timeout -k 10 200 rocprofv3 --pmc ${SETS[$p]} --output-format csv -d $OUT/p$p -- $AB --cfg $CFG --reps 60 --rounds 1 $LIB > $OUT/p$p.log 2>&1 || { echo "pass $p failed"; tail -5 $OUT/p$p.log; }
  fi
done
python3 - <<PY
import csv, glob, collections, json, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        k = 'stft_db_kernel' if 'stft_db_kernel' in k else ('stft_image_kernel' if 'stft_image' in k else ('colormap_kernel' if 'colormap' in k else k[:40]))
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {"tag": "$TAG", "cfg": "$CFG", "lib": os.path.basename("$LIB"), "counters_mean_per_dispatch": {}, "kernel_trace": {}}
for k, d in sorted(agg.items()):
    out["counters_mean_per_dispatch"][k] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        out["kernel_trace"][r['Name'][:60]] = {"calls": int(r['Calls']), "avg_ns": float(r['AverageNs']), "min_ns": float(r['MinNs']), "max_ns": float(r['MaxNs'])}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, d in out["counters_mean_per_dispatch"].items():
    print("==", k)
    for c, v in d.items():
        print(f"   {c:28s} {v:16.1f}")
for k, d in out["kernel_trace"].items():
    print(f"trace {k:60s} calls {d['calls']:6d} avg {d['avg_ns']/1e3:8.2f} us  min {d['min_ns']/1e3:8.2f} us")
PY
