#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c19; mkdir -p $O
for c in c3 c5; do tools/profile_bench.sh r03 $c > $O/prof_$c.log 2>&1; grep -E "failed" $O/prof_$c.log; python3 -c "
import json; d=json.load(open('gpurun_out/profiles_r03/r03_${c}_hbm_traffic.json')); print('$c', d.get('avg_us'), json.dumps(d.get('derived'), indent=0)[:1500])"; done
