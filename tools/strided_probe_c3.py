#!/usr/bin/env python3
"""DEV TOOL: C3 batches (8 channels, 2048 points, hop 512, AbsMean, 4096 columns per batch) as ONE strided dispatch: the three-stage
kernel (plan_select = 1) against the two-stage "B" kernel (plan_select = 2), workgroups per CU swept -- interleaved rounds, median."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])
n, hop, F, C = int(os.environ.get("SP_N", "2048")), 512, int(os.environ.get("SP_FRAMES", "4096")), int(os.environ.get("SP_CHANNELS", "8"))
K = int(os.environ.get("SP_BATCHES", "12"))
H = n // 2 + 1
pitch = (H + 31) // 32 * 32
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
ns = (F * hop + n - hop + 3) // 4 * 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = torch.rand((K, C, ns), device="cuda", generator=g) - 0.5
out = torch.empty((K, F, pitch), device="cuda")
st = torch.cuda.Stream()
algo = (4 * hop * C + 4 * H) * F
kw = dict(feedblocks=n // hop, mix_mode=jsg.capi.MIX_ABSMEAN)
cfgs = []
for sel in (1, 2):
    name = jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, out, plan_select=sel, **kw)
    for bpc in [int(v) for v in os.environ.get("SP_BPC_" + str(sel), "0").split(",")]:
        cfgs.append((f"{name}, {bpc or 'default'} workgroups per CU", sel, bpc))
times = {c[0]: [] for c in cfgs}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(int(os.environ.get("SP_ROUNDS", "8"))):
    for label, sel, bpc in cfgs:
        with torch.cuda.stream(st):
            jsg.stft_db_strided(plan, d_in, hop, F, out, plan_select=sel, blocks_per_cu=bpc, stream=st.cuda_stream, **kw)
            e0.record(st)
            for _ in range(5):
                jsg.stft_db_strided(plan, d_in, hop, F, out, plan_select=sel, blocks_per_cu=bpc, stream=st.cuda_stream, **kw)
            e1.record(st)
        torch.cuda.synchronize()
        times[label].append(e0.elapsed_time(e1) * 1e3 / (5 * K))
for label, _, _ in cfgs:
    t = sorted(times[label][2:]); med = t[len(t) // 2]
    print(json.dumps({"kernel": label, "us_per_batch": round(med, 2), "ffts_per_s": round(F * C / med * 1e6), "frac_of_8": round(algo / med / 8e6, 4)}))
