#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so
O=gpurun_out/c43; mkdir -p $O
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR > $O/wide1.log 2>&1; echo rc=$?
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 --streams 2 --threads 1 $CUR > $O/wide2.log 2>&1; echo rc=$?
grep -E "==|us/launch|fused" $O/wide1.log $O/wide2.log | cut -c1-260
