#!/bin/bash
# Run ON THE GPU BOX: the tried-tables per JOULE.  Every variant keeps the card busy for PW_SECONDS while socket power and
# core clock are sampled (tools/power_probe.py): rate, watts, MHz and units per joule side by side.
#   product build: C2 with / without the logarithm, reference / tail-plane layout, 2048 points mono three- vs two-stage, the C3 geometry with
#                  the three kernels (three-stage, two-stage "B", pair plan)
#   round-4-tree variants (tools/variants/libjsg_r4_*.so: prefetch depth 1 vs 2, four- vs eight-wave workgroups of the 1024-point plan)
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/power_ab.jsonl}
export PW_SECONDS=${PW_SECONDS:-5}
cd $GRAFT_REPO_ROOT
: > $OUT
if [ -z "${PW_VARIANTS_ONLY:-}" ]; then PW_ONLY=c2,c2lin,c2tail,c2048a,c2048b,c3a,c3,c3p python3 tools/power_probe.py >> $OUT 2>/dev/null; fi
for v in r4_pfd1_w4 r4_pfd2_w4 r4_pfd1_w8; do
    SP_ABI5=1 SP_LIB=tools/variants/libjsg_$v.so PW_ONLY=c2 python3 tools/power_probe.py >> $OUT 2>/dev/null
done
PW_ONLY=c2 python3 tools/power_probe.py >> $OUT 2>/dev/null   # the product's C2 once more at the end (drift of the box over the run)
python3 - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: j = json.loads(l)
    except Exception: continue
    print("%-22s %-88s %8.4f of 8 TB/s  %.4g units/s  %6.0f W  %5.0f MHz  %.4g units/J" % (j.get("build"), j["workload"][:88], j["frac_of_8TBps"], j["units_per_s"], j.get("socket_W_median") or 0, j.get("sclk_MHz_median") or 0, j.get("units_per_joule") or 0))
PY
