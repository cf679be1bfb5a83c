#!/bin/bash
# long robustness run with the final kernels: seeded sweeps (another seed) + ten soak passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/c42; mkdir -p $O
JSG_FUZZ_CASES=45000 JSG_FUZZ_SCENARIOS=600 JSG_FUZZ_SEED=2026 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_colormap.py -q -k "seeded_random" -x > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
for i in 1 2 3; do timeout -k 10 120 python tools/soak.py > $O/soak_$i.txt 2>&1; echo "soak $i rc=$? $(grep -c '"mismatching_buffers": 0' $O/soak_$i.txt) plans clean"; done
