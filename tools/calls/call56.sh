#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c56; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_colormap.py tests/test_gpu_parity.py -m gpu -x -q -k "image or fused or colo or strided" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
JSG_IMAGE_CPW=1 timeout -k 10 300 python tools/image_batch_probe.py 43 43 > $O/probe1.log 2>&1; echo "probe rc=$?"; tail -2 $O/probe1.log
timeout -k 10 300 python tools/image_batch_probe.py 43 43 > $O/probe2.log 2>&1; echo "probe rc=$?"; tail -2 $O/probe2.log
JSG_IMAGE_CPW=1 timeout -k 10 300 python tools/image_batch_probe.py 43 > $O/probe1.log 2>&1; tail -1 $O/probe1.log
timeout -k 10 300 python tools/image_batch_probe.py 43 > $O/probe2.log 2>&1; tail -1 $O/probe2.log
