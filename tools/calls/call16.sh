#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c16; mkdir -p $O
for c in c2 c3 c5; do tools/profile_bench.sh r03 $c > $O/prof_$c.log 2>&1; grep -E "failed|frac_of_8p0_from_trace_avg|traffic_over_algorithmic|\"avg_us\"" $O/prof_$c.log; done
tools/profile_overlap.sh r03 c2 4 > $O/ov_c2.log 2>&1; tools/profile_overlap.sh r03 c3 2 > $O/ov_c3.log 2>&1; tools/profile_overlap.sh r03 c5 3 > $O/ov_c5.log 2>&1
python3 -c "
import json
for c in ('c2','c3','c5'):
    d=json.load(open('gpurun_out/profiles_r03/r03_%s_overlap.json'%c)); print(c, {k:(round(v['us_per_launch_wall'],2), round(v['frac_of_8p0'],3), round(v['avg_dispatch_us'],2), round(v['mean_kernels_in_flight'],2), v['dispatches'], v['groups']) for k,v in d.items() if isinstance(v,dict) and 'frac_of_8p0' in v})"
for c in c2 c3 c5; do head -3 gpurun_out/profiles_r03/r03_${c}_kernel_stats.csv | cut -c1-160; done
