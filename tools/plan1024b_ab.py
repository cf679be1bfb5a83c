#!/usr/bin/env python3
"""DEV TOOL (round 6): interleaved A/B of the two 1024-point plans on the C2 dispatch -- the three-stage Cfg1024
(plan_select 1) against the two-stage Cfg1024B (plan_select 2: split-radix 16 x 32, 16 lanes per frame, four frames per wavefront) --
64 batches x 4096 frames per strided dispatch, reference column layout, rotating over ~1.1 GB, HIP events on the launch stream, socket
power and core clock from the card's hwmon files while each variant runs (tools/hwmon.py, read-only).
AB_CHANNELS / AB_MIX (absmean | per_channel) / AB_HOP / AB_BATCHES / AB_ROUNDS / AB_REPS / AB_TAIL=1 (tail-plane layout) / AB_BPC (grid of the
three-stage plan, workgroups per CU)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import jadespectrogram_amd as jsg
from hwmon import Hwmon, Watch
if os.environ.get("SP_LIB"):
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])

n, F = 1024, 4096
hop = int(os.environ.get("AB_HOP", "512")); C = int(os.environ.get("AB_CHANNELS", "1")); K = int(os.environ.get("AB_BATCHES", "64" if C == 1 else "8"))
mixname = os.environ.get("AB_MIX", "absmean")
per_ch = mixname == "per_channel"
mix = jsg.capi.MIX_PER_CHANNEL if per_ch else jsg.capi.MIX_ABSMEAN
rounds = int(os.environ.get("AB_ROUNDS", "7")); reps = int(os.environ.get("AB_REPS", "2400"))   # ~0.5 s per leg: long enough for the 50 ms telemetry samples
use_tail = os.environ.get("AB_TAIL") == "1"
M, H = n // 2, n // 2 + 1
pitch = M if use_tail else (H + 31) // 32 * 32
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
ns = (F * hop + n - hop + 3) // 4 * 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = torch.rand((K, C, ns), device="cuda", generator=g) - 0.5
rows = C if per_ch else 1
outs = {sel: torch.full(((K, C, F, pitch) if per_ch else (K, F, pitch)), -7.0, device="cuda") for sel in (1, 2)}
tails = {sel: (torch.full((K, rows, F), -7.0, device="cuda") if use_tail else None) for sel in (1, 2)}
st = torch.cuda.Stream()
kw = dict(feedblocks=n // hop, mix_mode=mix)
algo = ((4 * hop + 4 * H) * C if per_ch else (4 * hop * C + 4 * H)) * F * K
ffts = C * F * K
hw = Hwmon(0)


def run(sel):
    jsg.stft_db_strided(plan, d_in, hop, F, outs[sel], d_tail=tails[sel], plan_select=sel, blocks_per_cu=int(os.environ.get("AB_BPC", "0")) if sel == 1 else 0,
                        stream=st.cuda_stream, **kw)


names = {sel: jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, outs[sel], d_tail=tails[sel], plan_select=sel, **kw) for sel in (1, 2)}
with torch.cuda.stream(st):
    for sel in (1, 2):
        run(sel)
torch.cuda.synchronize()
# the two plans round differently in the last bits: compare against each other inside the float32 bound, not bit for bit
diff = (outs[1][..., :M] - outs[2][..., :M]).abs().max().item()
t0 = time.perf_counter()
with torch.cuda.stream(st):
    while time.perf_counter() - t0 < 1.0:      # settle the clocks
        run(1); run(2)
        torch.cuda.synchronize()
times = {1: [], 2: []}
tele = {1: [], 2: []}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    for sel in ((1, 2) if r % 2 == 0 else (2, 1)):
        with torch.cuda.stream(st):
            for _ in range(4):
                run(sel)
            if hw.ok:
                w = Watch(hw, settle_s=0.1, period_s=0.05)
                w.__enter__()
            e0.record(st)
            for _ in range(reps):
                run(sel)
            e1.record(st)
            torch.cuda.synchronize()
            if hw.ok:
                w.__exit__(None, None, None)
                tele[sel].append(w.summary())
        times[sel].append(e0.elapsed_time(e1) * 1e3 / reps)
res = {}
for sel in (1, 2):
    t = sorted(times[sel]); med = t[len(t) // 2]
    tl = [x for x in tele[sel] if x]
    def medk(k):
        v = sorted(x[k] for x in tl if x.get(k) is not None)
        return v[len(v) // 2] if v else None
    res[names[sel]] = {"plan_select": sel, "us_per_dispatch_median": round(med, 2), "best": round(t[0], 2), "frac_of_8_median": round(algo / med / 8e6, 4),
                       "ffts_per_s_median": round(ffts / med * 1e6), "socket_W": medk("socket_W_median"), "sclk_MHz": medk("sclk_MHz_median"),
                       "rounds_us": [round(x, 1) for x in times[sel]]}
a, b = res[names[1]]["us_per_dispatch_median"], res[names[2]]["us_per_dispatch_median"]
print(json.dumps({"workload": f"{K} batches x {C} channel(s) x {F} frames, hop {hop}, {mixname}, {'tail plane' if use_tail else 'reference layout'}",
                  "plans": res, "two_stage_vs_three_stage_rate": round(a / b, 4), "max_abs_dB_difference_between_the_plans": diff,
                  "rotation_MB": round((d_in.numel() + outs[1].numel()) * 4 / 1e6)}))
