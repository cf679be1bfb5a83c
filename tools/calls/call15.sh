#!/bin/bash
# round 3: the committed profile set (profiles/r03_*): per-kernel trace + HBM traffic + SQ counters, overlap traces, nbuf sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/c15; mkdir -p $O
for c in c2 c3 c5; do tools/profile_bench.sh r03 $c > $O/prof_$c.log 2>&1; grep -E "failed|frac_of_8p0_from_trace_avg|traffic_over_algorithmic|\"avg_us\"" $O/prof_$c.log; done
tools/profile_overlap.sh r03 c2 4 > $O/ov_c2.log 2>&1; grep -E "frac_of_8p0|mean_kernels|failed" $O/ov_c2.log
tools/profile_overlap.sh r03 c3 2 > $O/ov_c3.log 2>&1; grep -E "frac_of_8p0|mean_kernels|failed" $O/ov_c3.log
tools/profile_overlap.sh r03 c5 3 > $O/ov_c5.log 2>&1; grep -E "frac_of_8p0|mean_kernels|failed" $O/ov_c5.log
tools/nbuf_sweep.sh r03 c2 "20 40 60 80 120" > $O/nbuf.log 2>&1; cat $O/nbuf.log | cut -c1-250
ls gpurun_out/profiles_r03
