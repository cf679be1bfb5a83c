#!/bin/bash
cd $GRAFT_REPO_ROOT
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
for i in 1 2 3; do timeout -k 10 300 $AB --cfg c5 --reps 100 --rounds 2 $CUR 2>&1 | grep -E "fused|us/launch" | cut -c1-220; done
