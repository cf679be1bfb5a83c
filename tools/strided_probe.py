#!/usr/bin/env python3
"""DEV TOOL: C2 batches (mono 1024 points, hop 512, 4096 frames per batch) through
  a  one launch per batch in order (hipGraph replay)
  b  jsg_stft_db_launch_batches (the multi-stream pool)
  cN ONE strided launch with the plan's usual kernel (plan_select = 1), N workgroups per CU (SP_BPC list; 0 = the library's default)
  d  ONE strided launch in the staged form (Cfg1024S)
interleaved over SP_ROUNDS rounds (the clocks settle during the first ones): median / best us per batch, fraction of 8 TB/s in
algorithmic bytes, bit identity of every output against (a).  SP_LIB=path runs a variant build instead of the product library."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):          # a variant build (python -m jadespectrogram_amd._build --variant NAME ...)
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])

n, hop = 1024, int(os.environ.get("SP_HOP", "512"))
F = int(os.environ.get("SP_FRAMES", "4096"))
C = int(os.environ.get("SP_CHANNELS", "1"))
mix = jsg.capi.MIX_PER_CHANNEL if os.environ.get("SP_PER_CHANNEL") else jsg.capi.MIX_ABSMEAN
K = int(os.environ.get("SP_BATCHES", "60"))
reps = int(os.environ.get("SP_REPS", "10"))
rounds = int(os.environ.get("SP_ROUNDS", "5"))
modes = os.environ.get("SP_MODES", "a,b,c,d").split(",")
H, pitch = n // 2 + 1, 544
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
ns = F * hop + n - hop
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = (torch.rand((K, C, ns), device="cuda", generator=g) - 0.5)
shape = (K, C, F, pitch) if mix == jsg.capi.MIX_PER_CHANNEL else (K, F, pitch)
st = torch.cuda.Stream()
rows = C if mix == jsg.capi.MIX_PER_CHANNEL else 1
algo = (4 * hop * C + 4 * H * rows) * F
kw = dict(feedblocks=n // hop, mix_mode=mix)
ref = torch.full(shape, -7.0, device="cuda")
out = torch.full(shape, -7.0, device="cuda")

cfgs = []
if "a" in modes:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        for b in range(K):
            jsg.stft_db(plan, d_in[b], hop, F, out[b], stream=st.cuda_stream, **kw)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=st):
            for b in range(K):
                jsg.stft_db(plan, d_in[b], hop, F, out[b], stream=st.cuda_stream, **kw)
    cfgs.append(("a: one launch per batch, in order (hipGraph)", lambda: graph.replay()))
if "b" in modes:
    cfgs.append(("b: jsg_stft_db_launch_batches (4 streams, 2 threads; GPU_MAX_HW_QUEUES=" + str(os.environ.get("GPU_MAX_HW_QUEUES")) + ")",
                 lambda: jsg.stft_db_batches(plan, [(d_in[b], out[b]) for b in range(K)], hop, F, stream=st.cuda_stream, **kw)))
if "c" in modes:
    name = jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, out, plan_select=1, **kw)
    for bpc in [int(v) for v in os.environ.get("SP_BPC", "0").split(",")]:
        cfgs.append((f"c{bpc}: ONE strided launch, kernel {name}, {bpc or 'default'} workgroups per CU",
                     (lambda bpc: lambda: jsg.stft_db_strided(plan, d_in, hop, F, out, plan_select=1, blocks_per_cu=bpc, stream=st.cuda_stream, **kw))(bpc)))
if "d" in modes:
    name = jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, out, plan_select=2, **kw)
    cfgs.append((f"d: ONE strided launch, kernel {name}", lambda: jsg.stft_db_strided(plan, d_in, hop, F, out, plan_select=2, stream=st.cuda_stream, **kw)))

for b in range(K):
    jsg.stft_db(plan, d_in[b], hop, F, ref[b], **kw)
torch.cuda.synchronize()
times = {label: [] for label, _ in cfgs}
same = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    for label, fn in cfgs:
        with torch.cuda.stream(st):
            if r == 0:
                out.fill_(-7.0)
                fn()
                torch.cuda.synchronize()
                same[label] = bool(torch.equal(out, ref))
            fn()
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
        torch.cuda.synchronize()
        times[label].append(e0.elapsed_time(e1) * 1e3 / (reps * K))
for label, _ in cfgs:
    t = sorted(times[label])
    med, best = t[len(t) // 2], t[0]
    print(json.dumps({"mode": label, "us_per_batch_median": round(med, 3), "best": round(best, 3), "frac_of_8_median": round(algo / med / 8e6, 4),
                      "frac_of_8_best": round(algo / best / 8e6, 4), "frames_per_s": round(F * C / med * 1e6), "identical": same[label],
                      "rounds": [round(x, 3) for x in times[label]]}))
print(json.dumps({"lib": os.environ.get("SP_LIB", "product"), "batches": K, "frames": F, "hop": hop, "channels": C, "rotation_MB": round(K * algo / 1e6)}))
