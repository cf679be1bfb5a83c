#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c59; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "strided" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
JSG_IMAGE_HALVES=0 timeout -k 10 200 python tools/image_batch_probe.py 43 43 > $O/probe0.log 2>&1; echo "probe rc=$?"; tail -2 $O/probe0.log
timeout -k 10 200 python tools/image_batch_probe.py 43 43 > $O/probe1.log 2>&1; echo "probe rc=$?"; tail -2 $O/probe1.log
