#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into small, committable summaries (profiles/)."""
import csv, glob, json, os, sys, collections
out_dir, tag = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
dst = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)
res = {"tag": tag, "config": cfg, "command": open(os.path.join(out_dir, "command.txt")).read().strip().replace(os.environ.get("GRAFT_REPO_ROOT", root) + "/", "")}
try:
    res["commit"] = json.load(open(os.path.join(root, "jadespectrogram_amd", "_build_info.json"))).get("commit")
except Exception:
    res["commit"] = None
res["kernel_source_sha"] = bench.kernel_source_sha()   # bench.py trusts these figures only for a build with the same hash
# the bench line printed under the tracer (its in-order number must agree with the tracer's average)
try:
    line = json.loads(open(os.path.join(out_dir, "bench_lines.jsonl")).readline())
    res["bench_line_under_trace"] = {"value": line["value"], "unit": line["unit"], "avg_dispatch_us": line["roofline"]["avg_dispatch_us"],
                                     "frac": line["roofline"]["frac"], "timed_region_frac": line["roofline"]["timed_region_frac"],
                                     "dispatches_timed": line["steps"] * line["config"]["dispatches_per_step"]}
    ipl = int(line["config"].get("batches_per_dispatch") or 1)       # batches (c5: images) of one strided dispatch
except Exception:
    ipl = 1
res["batches_per_dispatch"] = ipl
# 1. kernel stats
main_kernel = "stft_db_kernel"
for f in glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"), "w") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:120]
            w.writerow(r)
    res["kernels"] = {}
    for r in rows:
        for key in ("stft_db_kernel", "colormap_kernel"):
            if key in r["Name"] and "spin_kernel" not in r["Name"] and (key not in res["kernels"] or int(r["Calls"]) > res["kernels"][key]["calls"]):   # the bench's own instantiation
                res["kernels"][key] = {"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                       "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
    if main_kernel in res["kernels"]:
        top = max(k["calls"] for k in res["kernels"].values())
        res["avg_us"] = sum(k["avg_us"] for k in res["kernels"].values() if k["calls"] * 2 > top)   # c5: both kernels of a launch
# 2. counters (per dispatch means)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("stft_db_kernel", "colormap_kernel"):
            if key in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# keep the instantiations the bench loop itself launched (most dispatches), not the one-off ones of the parity report
most = max((len(next(iter(d.values()))) for d in agg.values()), default=0)
agg = {k: d for k, d in agg.items() if len(next(iter(d.values()))) * 2 > most}
res["counters_mean_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in agg.items()}
res["counter_dispatches"] = {k: len(next(iter(d.values()))) for k, d in agg.items()}
fetch = sum(d.get("FETCH_SIZE", 0.0) for d in res["counters_mean_per_dispatch"].values())
write = sum(d.get("WRITE_SIZE", 0.0) for d in res["counters_mean_per_dispatch"].values())
if fetch and write:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
    # bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for streaming stores.
    res["fetch_bytes_raw"] = fetch * 1024
    res["fetch_bytes_corrected"] = fetch * 1024 * 2
    res["write_bytes"] = write * 1024
    res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    res["algorithmic_bytes_per_launch"] = bench.algorithmic_bytes_per_batch(bench.CONFIGS[cfg]) * ipl
    res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    if "avg_us" in res:
        res["frac_of_8p0_from_trace_avg"] = res["algorithmic_bytes_per_launch"] / (res["avg_us"] * 1e-6) / 8e12
# derived: instructions per FFT and how busy the vector and LDS pipes were (SQ counters are per dispatch means; SQ_WAVE_CYCLES and
# SQ_ACTIVE_INST_* count quad-cycles summed over the waves, SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES LDS-array / CU cycles)
c = bench.CONFIGS[cfg]
ffts = c["frames"] * c["channels"] * ipl
for kname, d in res["counters_mean_per_dispatch"].items():
    if "stft_db_kernel" not in kname or "SQ_WAVE_CYCLES" not in d:
        continue
    w = d.get("SQ_WAVES", 0.0)
    der = {"ffts_per_dispatch": ffts, "waves": w,
           "valu_instructions_per_fft": d.get("SQ_INSTS_VALU", 0.0) / ffts, "lds_instructions_per_fft": d.get("SQ_INSTS_LDS", 0.0) / ffts,
           "vmem_instructions_per_fft": d.get("SQ_INSTS_VMEM", 0.0) / ffts,
           "wave_lifetime_cycles": 4.0 * d["SQ_WAVE_CYCLES"] / max(w, 1.0),
           "share_of_wave_cycles": {k: d[v] / d["SQ_WAVE_CYCLES"] for k, v in (("issuing_valu", "SQ_ACTIVE_INST_VALU"), ("issuing_any", "SQ_ACTIVE_INST_ANY"),
                                                                              ("waiting_to_issue", "SQ_WAIT_INST_ANY"), ("parked_on_waitcnt_or_barrier", "SQ_WAIT_ANY"),
                                                                              ("waiting_to_issue_lds", "SQ_WAIT_INST_LDS")) if v in d},
           "lds_bank_conflict_share_of_lds_cycles": (d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]) if d.get("SQ_LDS_IDX_ACTIVE") else None,
           "lds_array_cycles_per_fft": d.get("SQ_LDS_IDX_ACTIVE", 0.0) / ffts,
           "note": "issuing_valu x (waves per SIMD: 5 at C2, 2 at C3 / C5) = share of the time a SIMD's vector pipe is issuing"}
    res.setdefault("derived", {})[kname] = der
json.dump(res, open(os.path.join(dst, f"{tag}_{cfg}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
