#!/bin/bash
# Run ON THE GPU BOX: per-dispatch durations of the bench's C2 dispatches from a rocprofv3 kernel trace (no counters): distribution and pattern in time.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/dispatch_spread; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config ${1:-c2} --steps 40 --warmup 5 --no-cpu-baseline --no-boundary --no-parity --no-extra --no-calibration --no-single --no-power > $OUT/bench.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, statistics, json
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "stft_db_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
s = sorted(d)
print(json.dumps({"dispatches": len(d), "min": s[0], "p10": s[len(s) // 10], "median": statistics.median(d), "mean": statistics.mean(d), "p90": s[9 * len(s) // 10], "max": s[-1],
                  "gap_median_us": statistics.median(gap), "gap_p90_us": sorted(gap)[9 * len(gap) // 10]}))
for k in range(0, len(d), 64):
    print(k, [round(x) for x in d[k:k + 16]])
PY
find $OUT -name '*.csv' -delete
