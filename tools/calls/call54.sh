#!/bin/bash
# round 3: re-record the committed profile set after the display-path changes (strided image batches, store phase)
cd $GRAFT_REPO_ROOT
O=gpurun_out/c54; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for c in c2 c3 c5; do tools/profile_bench.sh r03 $c > $O/prof_$c.log 2>&1; grep -E "failed" $O/prof_$c.log; done
tools/profile_overlap.sh r03 c2 4 > $O/ov_c2.log 2>&1; tools/profile_overlap.sh r03 c3 2 > $O/ov_c3.log 2>&1; tools/profile_overlap.sh r03 c5 3 > $O/ov_c5.log 2>&1
mkdir -p profiles; cp gpurun_out/profiles_r03/r03_* profiles/
python bench.py --steps 20 --warmup 5 > $O/bench_c2_steps20.json 2> $O/bench_c2.err
python bench.py > $O/bench_c2_default.json 2>> $O/bench_c2.err
python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c5 --images-per-launch 1 --no-cpu-baseline --no-boundary > $O/bench_c5_single_3streams.json 2>> $O/bench_c5.err
python bench.py --config c5 --images-per-launch 1 --streams 1 --no-cpu-baseline --no-boundary > $O/bench_c5_single_inorder.json 2>> $O/bench_c5.err
python bench.py --config c3 --streams 1 --no-cpu-baseline --no-boundary > $O/bench_c3_streams1.json 2>> $O/bench_c3.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    l=json.loads(open(f).read().strip().splitlines()[-1]); r=l["roofline"]
    print(f.split("/")[-1], "value %.4g ms/step %.3f region %.4f inorder_us %.3f frac %.4f (%s) events %.4f traffic %s" % (l["value"], l["ms_per_step"], r["timed_region_frac_of_8p0"], r["avg_launch_us"], r["frac"], "rocprof" if r["frac_rocprof"] else "events", r["frac_event_timed"], r["traffic"]), round(r["second_roof"]["frac"],3), (l["config"].get("same_region_default_environment") or {}).get("value"), (l.get("boundary") or {}).get("process_block_latency", {}).get("p99_us"))
PY
python3 -c "
import json
for c in ('c2','c3','c5'):
    d=json.load(open('profiles/r03_%s_overlap.json'%c)); print(c, {k:(round(v['us_per_launch_wall'],2), round(v['frac_of_8p0'],3), round(v['avg_dispatch_us'],2), round(v['mean_kernels_in_flight'],2)) for k,v in d.items() if isinstance(v,dict) and 'frac_of_8p0' in v})
    h=json.load(open('profiles/r03_%s_hbm_traffic.json'%c)); print('  ', round(h['avg_us'],3), round(h['traffic_over_algorithmic'],4), round(h['frac_of_8p0_from_trace_avg'],4), h['kernel_source_sha'], {k:(round(v['valu_instructions_per_fft'],1), round(v['share_of_wave_cycles']['issuing_valu'],3), round(v['share_of_wave_cycles']['parked_on_waitcnt_or_barrier'],3)) for k,v in h.get('derived',{}).items()})"

timeout -k 10 200 tools/variants/abbench --cfg c3big --reps 50 --rounds 3 jadespectrogram_amd/libjsg.so 2>&1 | grep -E "==|us/launch" | cut -c1-200
