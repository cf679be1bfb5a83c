#!/usr/bin/env python3
"""DEV TOOL: colour-loop kernel timing (C5 geometry: W=1875 columns x H=2049 bins) and C5 end to end."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
def t(fn, steps=48):
    if os.environ.get("JSG_KBENCH_GRAPH"):
        # GPU-side time per call with the host taken out: one hipGraph holding `steps` calls, replayed
        import time
        s2 = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s2):
            for _ in range(8): fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s2):
                for _ in range(steps): fn()
        torch.cuda.synchronize()
        for _ in range(2): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (10 * steps) * 1e6
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
for (W, H, pitch) in ((1875, 2049, 2080), (938, 1025, 1056), (4096, 513, 544)):
    nb = 8
    db = [torch.rand((W, pitch), device="cuda") * 120 - 70 for _ in range(nb)]
    lut = torch.from_numpy(jsg.colormap_lut(256, 6)).cuda()
    img = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(nb)]
    idx = [torch.zeros((H, W), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    k = [0]
    def f():
        i = k[0] % nb; k[0] += 1
        jsg.colormap(db[i], lut, -50.0, 50.0, d_argb=img[i], x_first=17, height=H)
    us = t(f)
    print(json.dumps(dict(kernel="colormap argb", W=W, H=H, us=round(us, 2), GBs=round(W * H * 8 / us / 1e3, 1))), flush=True)
    def g():
        i = k[0] % nb; k[0] += 1
        jsg.colormap(db[i], lut, -50.0, 50.0, d_argb=img[i], d_index=idx[i], x_first=17, height=H)
    us = t(g)
    print(json.dumps(dict(kernel="colormap argb+index", W=W, H=H, us=round(us, 2), GBs=round(W * H * 9 / us / 1e3, 1))), flush=True)
# C5 end to end: stereo 96 kHz, N=4096, hop 512, 1875 columns -> dB ring -> ARGB
n, hop, F = 4096, 512, 1875
plan = jsg.Plan(n, jsg.window(1, n))
x = [torch.rand((2, F * hop + n - hop), device="cuda") * 2 - 1 for _ in range(8)]
ring = [torch.empty((F, 2080), device="cuda") for _ in range(8)]
img = [torch.zeros((2049, F), dtype=torch.int32, device="cuda") for _ in range(8)]
lut = torch.from_numpy(jsg.colormap_lut(256, 6)).cuda()
k = [0]
def c5():
    i = k[0] % 8; k[0] += 1
    jsg.stft_db(plan, x[i], hop, F, ring[i], feedblocks=8)
    jsg.colormap(ring[i], lut, -50.0, 50.0, d_argb=img[i], height=2049)
us = t(c5)
print(json.dumps(dict(kernel="C5 stft+colormap", columns=F, us=round(us, 2), columns_per_s=round(F / us * 1e6), GBs_rgba_only=round(F * 12292 / us / 1e3, 1), GBs_with_db=round(F * 20488 / us / 1e3, 1))), flush=True)
