#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
timeout -k 10 300 $AB --cfg c5 --reps 200 --rounds 3 $CUR $V/libjsg_nostore.so $V/libjsg_nolut.so 2>&1 | grep -E "us/launch|fused" | cut -c1-220
