#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c51; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_colormap.py tests/test_gpu_parity.py -m gpu -x -q -k "image or fused or colo" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-250
timeout -k 10 300 python tools/image_batch_probe.py 4 16 43 > $O/probe.log 2>&1; echo "probe rc=$?"; tail -5 $O/probe.log
