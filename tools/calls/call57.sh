#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; V=tools/variants
O=gpurun_out/c57; mkdir -p $O
AB_IMGSTAMPS=1 timeout -k 10 300 $AB --cfg c5wide --reps 5 --rounds 1 $V/libjsg_imgstamp.so > $O/wide.log 2>&1; echo rc=$?
AB_IMGSTAMPS=1 timeout -k 10 300 $AB --cfg c5 --reps 20 --rounds 1 $V/libjsg_imgstamp.so > $O/c5.log 2>&1; echo rc=$?
grep -E "==|image stamps|median|fused" $O/wide.log $O/c5.log | cut -c1-200
