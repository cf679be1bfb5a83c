#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ / traffic counters of a probe script, per kernel instantiation.
#   usage: tools/pmc_probe.sh <out dir under gpurun_out> <python script> [VAR=value ...]
# Separate passes of at most eight SQ counters, FETCH_SIZE and WRITE_SIZE on their own, never combined with tracing.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; SCRIPT=$GRAFT_REPO_ROOT/$2; shift 2
for kv in "$@"; do export "$kv"; done
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PP_REPS=${PP_REPS:-2} PP_ROUNDS=${PP_ROUNDS:-2} TP_REPS=${TP_REPS:-2} TP_ROUNDS=${TP_ROUNDS:-2}
run() { timeout -k 10 300 rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $SCRIPT > $OUT/$1.log 2>&1 || echo "$1 pass failed"; }
run sq  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM"
run sq3 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft_db_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    m["dispatches"] = len(next(iter(d.values())))
    res[k[:140]] = m
json.dump(res, open(os.path.join(out, "counters.json"), "w"), indent=1)
for k, m in res.items():
    w = m.get("SQ_WAVES", 0) or 1
    print(k[:100])
    print("   per wave: VALU %.0f  LDS %.0f  VMEM %.0f  SALU %.0f | wave cycles %.0f  valu-active %.0f (%.1f %%)  wait_inst_any %.1f %%  wait_lds %.1f %% | FETCH %.1f MB x2  WRITE %.1f MB | lds bank conflict %.1f %%" % (
        m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_VMEM", 0) / w, m.get("SQ_INSTS_SALU", 0) / w,
        m.get("SQ_WAVE_CYCLES", 0) * 4 / w, m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / w, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)),
        100 * m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)), 100 * m.get("SQ_WAIT_INST_LDS", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)),
        m.get("FETCH_SIZE", 0) / 1024, m.get("WRITE_SIZE", 0) / 1024, 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
find $OUT -name '*counter_collection.csv' -delete
