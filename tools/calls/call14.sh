#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
run() { echo "--- $1 ($2 streams $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 2 --reps 200 --rounds 3 $CUR $V/libjsg_nomis.so $V/libjsg_nopost.so $V/libjsg_memcl.so $V/libjsg_minreg.so $V/libjsg_relax.so 2>&1 | grep -E "us/launch|vs first" | grep -v "= 0, elements" | cut -c1-200; }
run "X=1" c2 1
run "X=1" c3 1
run "X=1" c5 1
run "JSG_4096_PLAN=3" c5 1
