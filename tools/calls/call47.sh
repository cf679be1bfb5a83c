#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
O=gpurun_out/c47; mkdir -p $O
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR $V/libjsg_imgabl6.so $V/libjsg_imgabl7.so $V/libjsg_imgabl8.so > $O/wide.log 2>&1 && echo ok
grep -E "==|us/launch|vs first" $O/wide.log | cut -c1-260
