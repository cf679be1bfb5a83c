#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
run() { echo "--- $1 ($2 $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 1 --reps 200 --rounds 3 $CUR $V/libjsg_plain.so 2>&1 | grep -E "us/launch|vs first|fused" | cut -c1-220; }
run "X=1" c5 1
