#!/bin/bash
# round 3: fuzz campaign + soak with the final kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/c25; mkdir -p $O
JSG_FUZZ_CASES=6000 JSG_FUZZ_SCENARIOS=200 JSG_FUZZ_SEED=31 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_colormap.py -q -k "seeded_random" > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz.log
timeout -k 10 600 python tools/soak.py > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -8 $O/soak.txt
