#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
run() { echo "--- $1 ($2 streams $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 2 --reps 200 --rounds 3 $V/libjsg_r02.so $CUR $V/libjsg_wps6.so $V/libjsg_wpb9.so $V/libjsg_wpb9b.so 2>&1 | grep -E "==|us/launch" | cut -c1-200; }
run "X=1" s1024 4
run "X=1" c2 4
run "X=1" big 1
