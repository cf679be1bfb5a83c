#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
run() { echo "--- $1 ($2)"; env $1 timeout -k 10 300 $AB --cfg $2 --reps 100 --rounds 3 $CUR 2>&1 | grep -E "==|us/launch" | cut -c1-150; }
run "JSG_2048_PLAN=2" x2048
run "JSG_2048_PLAN=3" x2048
