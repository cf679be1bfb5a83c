#!/bin/bash
# round 3, call 2: C2 with a rotation well past the Infinity Cache: floors, ablations, pad-line A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/c02; mkdir -p $O
AB=tools/variants/abbench; CUR=jadespectrogram_amd/libjsg.so; V=tools/variants
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
AB_FLOOR=1 AB_PAD=0 timeout -k 10 200 $AB --cfg c2 --streams 4 --threads 2 --reps 400 --rounds 3 $V/libjsg_r02.so $CUR $V/libjsg_abl1.so $V/libjsg_abl2.so > $O/ab_c2_pad0.log 2>&1
AB_PAD=1 timeout -k 10 200 $AB --cfg c2 --streams 4 --threads 2 --reps 400 --rounds 3 $V/libjsg_r02.so $CUR > $O/ab_c2_pad1.log 2>&1
AB_ROT_MB=300 AB_PAD=0 timeout -k 10 200 $AB --cfg c2 --streams 4 --threads 2 --reps 400 --rounds 3 $CUR > $O/ab_c2_rot300.log 2>&1
grep -E "==|us/launch|floor" $O/ab_c2_pad0.log $O/ab_c2_pad1.log $O/ab_c2_rot300.log | cut -c1-230
AB_PAD=0 timeout -k 10 200 $AB --cfg big --reps 60 --rounds 3 $V/libjsg_r02.so $CUR > $O/ab_big0.log 2>&1
AB_PAD=1 timeout -k 10 200 $AB --cfg big --reps 60 --rounds 3 $CUR >> $O/ab_big0.log 2>&1
grep -E "==|us/launch" $O/ab_big0.log | cut -c1-200
for fl in "" "--no-pad-line"; do python bench.py --nbuf 60 --steps 20 --warmup 5 --no-cpu-baseline $fl > $O/bench_c2.json 2> $O/bench_c2.err; python - <<PY
import json; l=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1]); print("nbuf 60 '$fl' value %.4g region_frac %.4f inorder_us %.3f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"], l["roofline"]["avg_launch_us"]))
PY
done
for bpc in 2 3; do python bench.py --nbuf 60 --steps 20 --warmup 5 --no-cpu-baseline --blocks-per-cu $bpc > $O/bench_c2.json 2> $O/bench_c2.err; python - <<PY
import json; l=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1]); print("nbuf 60 bpc $bpc value %.4g region_frac %.4f" % (l["value"], l["roofline"]["timed_region_frac_of_8p0"]))
PY
done
