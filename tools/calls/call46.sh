#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
O=gpurun_out/c46; mkdir -p $O
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR $V/libjsg_imgabl2.so $V/libjsg_imgabl4.so $V/libjsg_imgabl5.so > $O/wide.log 2>&1 && echo ok
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $CUR $V/libjsg_imgabl2.so $V/libjsg_imgabl4.so $V/libjsg_imgabl5.so > $O/c5.log 2>&1 && echo ok
grep -E "==|us/launch" $O/wide.log $O/c5.log | cut -c1-260
