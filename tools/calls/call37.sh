#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
echo "--- c2, workgroups of 8 (product) / 4 / 6 / 12 / 16 wavefronts, 1 GB rotation"
timeout -k 10 300 $AB --cfg c2 --streams 4 --threads 2 --reps 400 --rounds 3 $CUR $V/libjsg_wpb4.so $V/libjsg_wpb6.so $V/libjsg_wpb12.so $V/libjsg_wpb16.so 2>&1 | grep -E "us/launch" | cut -c1-200
timeout -k 10 300 $AB --cfg big --reps 60 --rounds 3 $CUR $V/libjsg_wpb4.so $V/libjsg_wpb6.so $V/libjsg_wpb12.so $V/libjsg_wpb16.so 2>&1 | grep -E "us/launch" | cut -c1-200
