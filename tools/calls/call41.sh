#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c41; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "every_bin" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-300
