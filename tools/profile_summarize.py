#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into small, committable summaries (profiles/)."""
import csv, glob, json, os, sys, collections
out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)
res = {"tag": tag, "command": "python3 bench.py --streams 1 --steps 5000 --warmup 500 --no-cpu-baseline"}
# 1. kernel stats
for f in glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:120]
            w.writerow(r)
    for r in rows:
        if "stft_db_kernel" in r["Name"]:
            res["kernel"] = r["Name"][:120]
            res["calls"] = int(r["Calls"]); res["avg_ns"] = float(r["AverageNs"])
            res["min_ns"] = float(r["MinNs"]); res["max_ns"] = float(r["MaxNs"])
# 2. counters (per dispatch means over the stft kernel)
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out_dir, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft_db_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res["counters_mean_per_dispatch"] = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
c = res["counters_mean_per_dispatch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
    # bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for streaming stores.
    res["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
    res["fetch_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
    res["write_bytes"] = c["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    res["algorithmic_bytes_per_launch"] = 4100 * 4096
json.dump(res, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
