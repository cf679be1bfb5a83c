#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c62; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "strided" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-250
