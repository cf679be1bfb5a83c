#!/bin/bash
# round 3, call 4: 16-byte stores + frame reuse: tests, then A/B (1 GB rotation)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c04; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
run() { echo "--- $1"; env $1 timeout -k 10 300 $AB --cfg $2 --streams 4 --threads 2 --reps 400 --rounds 3 $V/libjsg_r02.so $CUR 2>&1 | grep -E "us/launch|vs first" | cut -c1-200; }
run "X=1" c2
run "JSG_NO_ST16=1" c2
run "JSG_NO_REUSE=1" c2
run "JSG_NO_REUSE=1 JSG_NO_ST16=1" c2
run "JSG_STFT_MAX_BLOCKS=128" c2
run "AB_BPC=2" c2
echo "--- in-order bpc sweeps"
for b in 1 2 8; do echo "bpc $b"; JSG_STFT_BLOCKS_PER_CU=$b timeout -k 10 300 $AB --cfg c2 --reps 400 --rounds 3 $V/libjsg_r02.so $CUR 2>&1 | grep -E "us/launch" | cut -c1-200; done
run "X=1" c4
run "X=1" big
