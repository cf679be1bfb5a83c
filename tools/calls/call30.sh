#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
run() { echo "--- $1 ($2 streams $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 2 --reps 200 --rounds 3 $CUR $V/libjsg_stg1.so $V/libjsg_stg2.so $V/libjsg_stg4.so 2>&1 | grep -E "==|us/launch" | cut -c1-200; }
run "X=1" c3 2
run "X=1" c5 1
