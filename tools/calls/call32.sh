#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c32; mkdir -p $O
JSG_2048_PLAN=3 tools/profile_bench.sh r03x c3 > $O/prof_c3.log 2>&1; grep -E "failed" $O/prof_c3.log
python3 -c "
import json
h=json.load(open('gpurun_out/profiles_r03x/r03x_c3_hbm_traffic.json')); print(round(h['avg_us'],3)); print(json.dumps(h.get('derived'),indent=0)[:1500])"
