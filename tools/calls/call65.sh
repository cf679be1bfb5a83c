#!/bin/bash
# seeded sweeps (incl. the strided image batches) + soak with the final kernels of round 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/c65; mkdir -p $O
JSG_FUZZ_CASES=40000 JSG_FUZZ_SCENARIOS=500 JSG_FUZZ_SEED=909 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_colormap.py -q -k "seeded_random" -x > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
timeout -k 10 120 python tools/soak.py > $O/soak.txt 2>&1; echo "soak rc=$? $(grep -c '"mismatching_buffers": 0' $O/soak.txt) plans clean"
