#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c12; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
run() { echo "--- $1 ($2 streams $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 2 --reps 200 --rounds 3 $CUR $V/libjsg_ilp.so $V/libjsg_iilp.so 2>&1 | grep -E "us/launch|vs first" | cut -c1-200; }
run "X=1" c2 4
run "X=1" c3 2
run "X=1" c5 3
run "X=1" big 1
run "X=1" c4 1
run "X=1" n8192 1
run "X=1" n512 1
