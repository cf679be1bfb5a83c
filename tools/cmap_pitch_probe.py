#!/usr/bin/env python3
"""DEV TOOL: colour loop with the image pitch as given (W) vs rounded up to 32 pixels (128-byte rows), GPU-side (hipGraph)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
def t(fn, steps=48):
    s2 = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
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
for (W, H, pitch) in ((1875, 2049, 2080), (938, 1025, 1056)):
    lut = torch.from_numpy(jsg.colormap_lut(256, 6)).cuda()
    for wp in (W, (W + 31) // 32 * 32):
        for x0 in (0, 17):
            nb = 8
            db = [torch.rand((W, pitch), device="cuda") * 120 - 70 for _ in range(nb)]
            img = [torch.zeros((H, wp), dtype=torch.int32, device="cuda")[:, :W] for _ in range(nb)]
            k = [0]
            def f():
                i = k[0] % nb; k[0] += 1
                jsg.colormap(db[i], lut, -50.0, 50.0, d_argb=img[i], x_first=x0, height=H)
            print(json.dumps(dict(W=W, H=H, image_pitch=wp, x_first=x0, us=round(t(f), 2))), flush=True)
