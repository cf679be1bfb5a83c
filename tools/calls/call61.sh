#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; V=tools/variants
O=gpurun_out/c61; mkdir -p $O
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $V/libjsg_cur.so $V/libjsg_ilpall.so > $O/wide.log 2>&1
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $V/libjsg_cur.so $V/libjsg_ilpall.so > $O/c5.log 2>&1
timeout -k 10 300 $AB --cfg x4096 --reps 50 --rounds 3 $V/libjsg_cur.so $V/libjsg_ilpall.so > $O/x4096.log 2>&1
grep -E "==|us/launch" $O/wide.log $O/c5.log $O/x4096.log | cut -c1-200
