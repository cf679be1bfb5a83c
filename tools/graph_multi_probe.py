#!/usr/bin/env python3
"""DEV TOOL: one hipGraph holding 24 independent C2 launches forked over S streams; replay -> us per launch."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
n, hop, frames = 1024, 512, 4096
plan = jsg.Plan(n, jsg.window(1, n))
nbuf = 24
d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
d_out = [torch.empty((frames, 544), device="cuda") for _ in range(nbuf)]
L = [jsg.StftLaunch(plan, d_in[b], hop, frames, d_out[b], feedblocks=2) for b in range(nbuf)]
for b in range(nbuf): L[b].launch(C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
for S in (1, 2, 4, 6):
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(S - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            for st in side: st.wait_stream(main)                       # fork
            allst = [main] + side
            for b in range(nbuf):
                L[b].launch(C.c_void_p(allst[b % S].cuda_stream))
            for st in side: main.wait_stream(st)                       # join
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    R = 100
    t0 = time.perf_counter()
    for _ in range(R): g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(streams_in_graph=S, us_per_launch=round(dt / (R * nbuf) * 1e6, 2), Mframes_s=round(R * nbuf * frames / dt / 1e6, 1),
                          frac=round(R * nbuf * frames * 4100 / dt / 8e12, 4))), flush=True)
