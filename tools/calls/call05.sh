#!/bin/bash
# round 3, call 5: C5 with the one-wavefront-per-frame 4096-point kernel ("B") forced; tests of the new auto-selection tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/c05; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
run() { echo "--- $1 ($2 $3)"; env $1 timeout -k 10 300 $AB --cfg $2 --streams $3 --threads 1 --reps 200 --rounds 3 $CUR 2>&1 | grep -E "us/launch|vs first|fused" | cut -c1-220; }
run "X=1" c5 3
run "JSG_4096_PLAN=2" c5 3
run "JSG_4096_PLAN=2 JSG_STFT_BLOCKS_PER_CU=1" c5 3
run "X=1" x4096 1
run "JSG_4096_PLAN=2" x4096 1
run "JSG_4096_PLAN=3" x4096 1
run "X=1" mid 1
run "JSG_4096_PLAN=2" mid 1
