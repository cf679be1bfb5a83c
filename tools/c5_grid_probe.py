#!/usr/bin/env python3
"""DEV TOOL (round 5): the C5 strided image dispatch (44 stereo 4096-point images of 1875 columns -> ARGB, one kernel) with 1 / 2 / 3 / 4 ...
workgroups per CU in the grid (the one-wavefront-per-frame kernel holds ONE workgroup per CU at a time: more than one queue behind it)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])
n, hop, C, F, K = 4096, 512, 2, 1875, int(os.environ.get("C5_IMAGES", "44"))
H = n // 2 + 1
pitch = (F + 31) // 32 * 32
ns = (F * hop + n - hop + 3) // 4 * 4
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = torch.rand((K, C, ns), device="cuda", generator=g) - 0.5
lut = torch.from_numpy(jsg.colormap_lut(256, jsg.capi.CM_JADE)).cuda()
bpcs = [int(v) for v in os.environ.get("C5_BPC", "1,2,3,4,8").split(",")]
imgs = {b: torch.zeros((K, H, pitch), dtype=torch.int32, device="cuda") for b in bpcs}
st = torch.cuda.Stream()
algo = K * F * (C * hop * 4 + H * 4)
fns = {b: (lambda b=b: jsg.stft_image_strided(plan, d_in, hop, F, lut, -50.0, 50.0, imgs[b][:, :, :F], None, stream=st.cuda_stream, feedblocks=n // hop,
                                               mix_mode=jsg.capi.MIX_ABSMEAN, blocks_per_cu=b)) for b in bpcs}
with torch.cuda.stream(st):
    for b in bpcs: fns[b]()
torch.cuda.synchronize()
times = {b: [] for b in bpcs}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(int(os.environ.get("C5_ROUNDS", "7"))):
    for b in bpcs:
        with torch.cuda.stream(st):
            fns[b](); e0.record(st)
            for _ in range(6): fns[b]()
            e1.record(st)
        torch.cuda.synchronize()
        times[b].append(e0.elapsed_time(e1) * 1e3 / 6)
for b in bpcs:
    t = sorted(times[b]); med = t[len(t) // 2]
    print(json.dumps({"blocks_per_cu": b, "us_per_dispatch_median": round(med, 1), "best": round(t[0], 1), "columns_per_s": round(K * F / med * 1e6), "frac_of_8": round(algo / med / 8e6, 4),
                      "identical_to_first": bool(torch.equal(imgs[b], imgs[bpcs[0]]))}))
