#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
O=gpurun_out/c53; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_colormap.py tests/test_gpu_parity.py -m gpu -x -q -k "image or fused or colo" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR $V/libjsg_spB.so $V/libjsg_imgabl2.so > $O/wide.log 2>&1 && echo ok
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $CUR $V/libjsg_spB.so > $O/c5.log 2>&1 && echo ok
grep -E "==|us/launch|differing" $O/wide.log $O/c5.log | grep -v imgabl2.so.*differing | cut -c1-260
timeout -k 10 300 python tools/image_batch_probe.py 43 43 > $O/probe.log 2>&1; echo "probe rc=$?"; tail -2 $O/probe.log
