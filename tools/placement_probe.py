#!/usr/bin/env python3
"""DEV TOOL: does the PLACEMENT of the input batches and the rings in HBM matter for the strided C2 dispatch?  (The same kernel measured
0.586 and 0.628 of 8 TB/s in two processes on one box: only the addresses differed.)  One 4 GiB pool; inputs at offset 0, rings at
in_bytes + delta for a sweep of deltas; interleaved rounds, median."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg

n, hop, F, K = 1024, 512, 4096, 64
H, pitch = 513, 544
ns = (F * hop + n - hop + 3) // 4 * 4
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
in_floats, out_floats = K * ns, K * F * pitch
pool = torch.empty((4 << 30) // 4, dtype=torch.float32, device="cuda")
pool[:in_floats].uniform_(-0.5, 0.5)
d_in = pool[:in_floats].view(K, 1, ns)
st = torch.cuda.Stream()
algo = 4100 * F
deltas_kb = [int(v) for v in os.environ.get("PP_DELTAS_KB", "0,4,64,256,1024,2048,3072,4096,8192,65536,262144,1048576").split(",")]
outs = []
for dk in deltas_kb:
    off = (in_floats + dk * 256 + 31) // 32 * 32
    outs.append((dk, pool[off:off + out_floats].view(K, F, pitch)))
# and the usual way: separate torch allocations
sep_in = torch.empty((K, 1, ns), device="cuda").copy_(d_in)
sep_out = torch.empty((K, F, pitch), device="cuda")
cfgs = [(f"pool, rings {dk} KB behind the inputs", d_in, o) for dk, o in outs] + [("separate allocations", sep_in, sep_out)]
times = {c[0]: [] for c in cfgs}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(int(os.environ.get("PP_ROUNDS", "8"))):
    for label, i_, o_ in cfgs:
        with torch.cuda.stream(st):
            jsg.stft_db_strided(plan, i_, hop, F, o_, stream=st.cuda_stream)
            e0.record(st)
            for _ in range(10):
                jsg.stft_db_strided(plan, i_, hop, F, o_, stream=st.cuda_stream)
            e1.record(st)
        torch.cuda.synchronize()
        times[label].append(e0.elapsed_time(e1) * 1e3 / (10 * K))
for label, _, o_ in cfgs:
    t = sorted(times[label][2:])
    med = t[len(t) // 2]
    print(json.dumps({"placement": label, "us_per_batch": round(med, 3), "frac_of_8": round(algo / med / 8e6, 4), "in_ptr_mod_2MB": d_in.data_ptr() % (2 << 20), "out_ptr_mod_2MB": o_.data_ptr() % (2 << 20)}))
