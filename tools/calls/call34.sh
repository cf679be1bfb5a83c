#!/bin/bash
# C3 occupancy / LDS-pressure sweep (BASELINE configs[2]): the three-stage 2048-point kernel at different residencies and the two-stage kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/c34; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
echo "--- three-stage kernel (JSG_2048_PLAN=3): default | WPS 2 | WPS 4 (128 VGPR) | full tables (2 WGs/CU) | 8-wave WG (1 WG/CU)" | tee $O/sweep.txt
JSG_2048_PLAN=3 timeout -k 10 300 $AB --cfg c3 --streams 2 --threads 2 --reps 200 --rounds 3 $CUR $V/libjsg_o_w2.so $V/libjsg_o_w4.so $V/libjsg_o_twf0.so $V/libjsg_o_wpb8.so 2>&1 | grep -E "==|us/launch" | cut -c1-200 | tee -a $O/sweep.txt
echo "--- two-stage kernel (automatic choice for C3)" | tee -a $O/sweep.txt
timeout -k 10 300 $AB --cfg c3 --streams 2 --threads 2 --reps 200 --rounds 3 $CUR 2>&1 | grep -E "us/launch" | cut -c1-200 | tee -a $O/sweep.txt
