#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c09; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
run() { echo "--- $1 ($2 $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 1 --reps 200 --rounds 3 $V/libjsg_r02.so $CUR 2>&1 | grep -E "us/launch|vs first|fused" | cut -c1-220; }
run "JSG_4096_PLAN=2" x4096 1
run "X=1" c5 3
run "X=1" n512 1
