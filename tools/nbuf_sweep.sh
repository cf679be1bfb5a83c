#!/bin/bash
# Run ON THE GPU BOX: whole-job rate of bench.py's timed region against the size of the rotation (distinct batches), to show how
# much of a figure leans on the 256 MiB Infinity Cache.   usage: tools/nbuf_sweep.sh <tag> <config> "<nbuf values>"
TAG=${1:-r04}; CFG=${2:-c2}; NB=${3:-"16 32 64 96 128"}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
python3 - "$TAG" "$CFG" $NB <<'PY'
import json, subprocess, sys, os
tag, cfg, nbs = sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3:]]
rows = []
for nb in nbs:
    out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--nbuf", str(nb), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-boundary",
                          "--no-extra", "--no-calibration", "--no-parity", "--dispatches-per-step", str(max(1, 1024 // nb))],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    l = json.loads(out)
    import bench
    c = bench.CONFIGS[cfg]
    H = c["n"] // 2 + 1
    per_batch = c["channels"] * (c["frames"] * c["hop"] + c["n"] - c["hop"]) * 4 + c["frames"] * ((H + 31) // 32 * 32) * 4
    rows.append({"batches_per_dispatch": l["config"]["batches_per_dispatch"], "rotation_MB": round(l["config"]["distinct_batches"] * per_batch / 1e6),
                 "value": l["value"], "unit": l["unit"], "frac": l["roofline"]["frac"], "timed_region_frac": l["roofline"]["timed_region_frac"],
                 "avg_dispatch_us": l["roofline"]["avg_dispatch_us"], "us_per_batch": l["roofline"]["avg_dispatch_us"] / l["config"]["batches_per_dispatch"],
                 "one_batch_per_dispatch_frac": l["roofline"].get("one_batch_per_dispatch", {}).get("frac")})
    print(rows[-1], flush=True)
json.dump({"tag": tag, "config": cfg, "infinity_cache_MB": 268, "command": "bench.py --config %s --nbuf N --dispatches-per-step 1024/N --steps 20 --warmup 5" % cfg,
           "kernel_source_sha": bench.kernel_source_sha(), "rows": rows}, open(os.path.join("gpurun_out", f"profiles_{tag}", f"{tag}_{cfg}_nbuf_sweep.json"), "w"), indent=1)
PY
