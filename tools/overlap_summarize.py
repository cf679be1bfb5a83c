#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace of `bench.py --streams S --no-graph --gate` (tools/profile_overlap.sh) into
profiles/<tag>_<cfg>_overlap.json: the achieved algorithmic bytes per second of the OVERLAPPED launch mode, from the tracer's own
begin / end timestamps -- sum of the algorithmic bytes of a step's dispatches / (last end - first begin) -- and how many STFT
kernels were in flight over that time.  The in-order leg of the same run (one stream) is summarised beside it.

usage: overlap_summarize.py <rocprof output dir> <tag> <cfg>
"""
import csv, glob, json, os, sys

out_dir, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench

algo = bench.algorithmic_bytes_per_launch(bench.CONFIGS[cfg])
rows = []
for f in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft_db_kernel" in r["Kernel_Name"] or "stft_image_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"][:110]))
rows.sort()
# groups: a new one starts where no STFT kernel has been running for 30 us (the gates / the host's pause between steps)
groups, cur, cur_end = [], [], None
for b, e, q, name in rows:
    if cur and b - cur_end > 30_000:
        groups.append(cur); cur = []; cur_end = None
    cur.append((b, e, q, name))
    cur_end = e if cur_end is None else max(cur_end, e)
if cur:
    groups.append(cur)


def summarise(gs):
    n = sum(len(g) for g in gs)
    span = sum(max(e for _, e, _, _ in g) - min(b for b, _, _, _ in g) for g in gs)   # ns
    busy = sum(e - b for g in gs for b, e, _, _ in g)
    # time-weighted number of kernels in flight, and the share of the span with k kernels in flight
    hist = {}
    for g in gs:
        ev = sorted([(b, 1) for b, _, _, _ in g] + [(e, -1) for _, e, _, _ in g])
        k, t_prev = 0, ev[0][0]
        for t, d in ev:
            hist[k] = hist.get(k, 0) + (t - t_prev)
            k += d; t_prev = t
    tot = sum(hist.values()) or 1
    return {"groups": len(gs), "dispatches": n, "sum_of_spans_us": span / 1e3,
            "us_per_launch_wall": span / 1e3 / max(n, 1),
            "achieved_GBps": n * algo / max(span, 1), "frac_of_8p0": n * algo / max(span, 1) / 8000.0,
            "avg_dispatch_us": busy / 1e3 / max(n, 1), "mean_kernels_in_flight": busy / max(span, 1),
            "share_of_span_with_k_in_flight": {str(k): round(v / tot, 4) for k, v in sorted(hist.items())}}


big = [g for g in groups if len(g) >= 16]
multi = [g for g in big if len({q for _, _, q, _ in g}) > 1]
single = [g for g in big if len({q for _, _, q, _ in g}) == 1]
res = {"tag": tag, "config": cfg, "algorithmic_bytes_per_launch": algo,
       "how": "rocprofv3 --kernel-trace timestamps; a group = the dispatches of one gated step (no STFT kernel running for 30 us "
              "before it); achieved = dispatches x algorithmic bytes / sum over groups of (last end - first begin)",
       "kernel": rows[len(rows) // 2][3] if rows else None}
try:
    res["command"] = open(os.path.join(out_dir, "command.txt")).read().strip().replace(os.environ.get("GRAFT_REPO_ROOT", root) + "/", "")
    res["commit"] = json.load(open(os.path.join(root, "jadespectrogram_amd", "_build_info.json"))).get("commit")
except Exception:
    pass
if multi:
    res["overlapped_streams"] = summarise(multi)
    res["overlapped_streams"]["queues"] = sorted({q for g in multi for _, _, q, _ in g})
if single:
    res["in_order_one_stream"] = summarise(single)
try:
    line = json.loads(open(os.path.join(out_dir, "bench_lines.jsonl")).readline())
    res["bench_line_under_trace"] = {"value": line["value"], "ms_per_step": line["ms_per_step"],
                                     "hip_streams": line["config"]["hip_streams_per_gpu"], "GPU_MAX_HW_QUEUES": line["config"]["GPU_MAX_HW_QUEUES"]}
except Exception:
    pass
dst = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)
json.dump(res, open(os.path.join(dst, f"{tag}_{cfg}_overlap.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
