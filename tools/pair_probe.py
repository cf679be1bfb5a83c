#!/usr/bin/env python3
"""DEV TOOL (round 5): the C3 strided dispatch (8 channels, 2048 points, hop 512, AbsMean, 12 batches of 4096 columns) with each of the
three 2048-point kernels -- Cfg2048 (plan_select 1), Cfg2048B (2), Cfg2048P (3: the pair plan) -- interleaved rounds in one process, HIP
events on the launch stream.  PP_CHANNELS / PP_HOP / PP_FRAMES / PP_BATCHES / PP_PLANS / PP_TAIL=1 (tail-plane layout)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])
n = 2048; hop = int(os.environ.get("PP_HOP", "512")); C = int(os.environ.get("PP_CHANNELS", "8"))
F = int(os.environ.get("PP_FRAMES", "4096")); K = int(os.environ.get("PP_BATCHES", "12"))
reps = int(os.environ.get("PP_REPS", "8")); rounds = int(os.environ.get("PP_ROUNDS", "9"))
plans = [int(v) for v in os.environ.get("PP_PLANS", "2,3").split(",")]
bpcs = [int(v) for v in os.environ.get("PP_BPC", "0").split(",")]          # workgroups per CU of the grid (0: the library's default)
use_tail = bool(os.environ.get("PP_TAIL"))
M, H = n // 2, n // 2 + 1
pitch = M if use_tail else 1056
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
ns = (F * hop + n - hop + 3) // 4 * 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = torch.rand((K, C, ns), device="cuda", generator=g) - 0.5
outs = {p: torch.full((K, F, pitch), -7.0, device="cuda") for p in plans}
tails = {p: (torch.full((K, 1, F), -7.0, device="cuda") if use_tail else None) for p in plans}
st = torch.cuda.Stream()
kw = dict(feedblocks=n // hop, mix_mode=jsg.capi.MIX_ABSMEAN)
algo = (4 * hop * C + 4 * H) * F
names = {p: jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, outs[p], plan_select=p, d_tail=tails[p], **kw) for p in plans}
plans = [(p, b) for p in plans for b in bpcs]
outs = {pb: outs[pb[0]] for pb in plans}; tails = {pb: tails[pb[0]] for pb in plans}; names = {pb: names[pb[0]] for pb in plans}
fns = {pb: (lambda pb=pb: jsg.stft_db_strided(plan, d_in, hop, F, outs[pb], plan_select=pb[0], blocks_per_cu=pb[1], d_tail=tails[pb], stream=st.cuda_stream, **kw)) for pb in plans}
with torch.cuda.stream(st):
    for p in plans:
        fns[p]()
torch.cuda.synchronize()
times = {p: [] for p in plans}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    for p in plans:
        with torch.cuda.stream(st):
            fns[p]()
            e0.record(st)
            for _ in range(reps):
                fns[p]()
            e1.record(st)
        torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) * 1e3 / reps)
base = outs[plans[0]][..., :M].double()
for p in plans:
    t = sorted(times[p]); med = t[len(t) // 2]
    dev = float((outs[p][..., :M].double() - base).abs().max())
    print(json.dumps({"plan_select": p[0], "blocks_per_cu": p[1], "kernel": names[p], "us_per_dispatch_median": round(med, 1), "best": round(t[0], 1), "fft_per_s_median": round(K * F * C / med * 1e6),
                      "frac_of_8_median": round(K * algo / med / 8e6, 4), "max_abs_dB_difference_to_first_plan": dev, "rounds": [round(x, 1) for x in times[p]]}))
print(json.dumps({"channels": C, "hop": hop, "frames": F, "batches": K, "tail_plane": use_tail}))
