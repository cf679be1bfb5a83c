#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
O=gpurun_out/c50; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_colormap.py tests/test_gpu_parity.py -m gpu -x -q -k "image or fused or colo" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-250
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR $V/libjsg_sp0.so $V/libjsg_sp1.so $V/libjsg_spA.so $V/libjsg_imgabl2.so > $O/wide.log 2>&1 && echo ok
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $CUR $V/libjsg_sp0.so $V/libjsg_sp1.so $V/libjsg_spA.so $V/libjsg_imgabl2.so > $O/c5.log 2>&1 && echo ok
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 --streams 3 --threads 1 $CUR $V/libjsg_sp0.so $V/libjsg_spA.so > $O/c5s3.log 2>&1 && echo ok
grep -E "==|us/launch|differing" $O/wide.log $O/c5.log $O/c5s3.log | grep -v imgabl2.so.*differing | cut -c1-260
