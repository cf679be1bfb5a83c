#!/bin/bash
# DEV TOOL: address-translation counters of the C2 launch (separate PMC pass, no tracing).  usage: tools/pmc_tlb.sh [kprof args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tlb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/kprof.py "$@" > $OUT/p1.log 2>&1 || { echo "pass failed"; tail -5 $OUT/p1.log; }
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stft_db_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(f"  {k:34s} mean {sum(v)/len(v):14.1f}  min {min(v):12.1f} max {max(v):12.1f} (n={len(v)})")
PY
