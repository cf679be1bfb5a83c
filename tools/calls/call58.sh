#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; V=tools/variants; CUR=jadespectrogram_amd/libjsg.so
O=gpurun_out/c58; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_colormap.py tests/test_gpu_parity.py -m gpu -x -q -k "image or fused or colo or strided" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
AB_IMGSTAMPS=1 timeout -k 10 300 $AB --cfg c5 --reps 20 --rounds 1 $V/libjsg_imgstamp2.so > $O/c5st.log 2>&1; echo rc=$?
AB_IMGSTAMPS=1 JSG_IMAGE_CPW=1 timeout -k 10 300 $AB --cfg c5wide --reps 5 --rounds 1 $V/libjsg_imgstamp2.so > $O/widest.log 2>&1; echo rc=$?
grep -E "==|image stamps|median" $O/c5st.log $O/widest.log | cut -c1-200
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $CUR $V/libjsg_spB.so > $O/c5.log 2>&1
JSG_IMAGE_CPW=1 timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR $V/libjsg_spB.so > $O/wide1.log 2>&1
timeout -k 10 300 $AB --cfg c5wide --reps 20 --rounds 3 $CUR $V/libjsg_spB.so > $O/wide2.log 2>&1
grep -E "==|us/launch" $O/c5.log $O/wide1.log $O/wide2.log | cut -c1-200
JSG_IMAGE_CPW=1 timeout -k 10 300 python tools/image_batch_probe.py 43 > $O/probe1.log 2>&1; tail -1 $O/probe1.log
timeout -k 10 300 python tools/image_batch_probe.py 43 > $O/probe2.log 2>&1; tail -1 $O/probe2.log
