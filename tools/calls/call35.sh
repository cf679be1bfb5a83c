#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c35; mkdir -p $O
JSG_FUZZ_CASES=20000 JSG_FUZZ_SCENARIOS=400 JSG_FUZZ_SEED=77 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_colormap.py -q -k "seeded_random" -x > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log
