#!/bin/bash
# round 3, call 3: C2 memory-pattern ablations (1 GB rotation)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c03; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
AB_FLOOR=1 AB_FLOOR_WG=256 timeout -k 10 300 $AB --cfg c2 --streams 4 --threads 2 --reps 400 --rounds 3 $CUR $V/libjsg_abl1.so $V/libjsg_m1.so $V/libjsg_m2.so $V/libjsg_m4.so $V/libjsg_m8.so $V/libjsg_m7.so $V/libjsg_f2.so $V/libjsg_f8.so > $O/ab_c2.log 2>&1
grep -E "==|us/launch|floor" $O/ab_c2.log | cut -c1-230
AB_FLOOR=1 timeout -k 10 300 $AB --cfg big --reps 60 --rounds 3 $CUR $V/libjsg_abl1.so $V/libjsg_m1.so $V/libjsg_m2.so $V/libjsg_m7.so $V/libjsg_f2.so > $O/ab_big.log 2>&1
grep -E "==|us/launch|floor" $O/ab_big.log | cut -c1-230
