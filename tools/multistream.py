#!/usr/bin/env python3
"""DEV TOOL: C2 launches (4096 frames each) issued round-robin on S HIP streams -- aggregate frames/s."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
n, hop, frames = 1024, 512, 4096
plan = jsg.Plan(n, jsg.window(1, n))
nbuf = 24
d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
d_out = [torch.empty((frames, 544), device="cuda") for _ in range(nbuf)]
for S in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    K = 3000
    for i in range(200):
        jsg.stft_db(plan, d_in[i % nbuf], hop, frames, d_out[i % nbuf], feedblocks=2, stream=streams[i % S].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        jsg.stft_db(plan, d_in[i % nbuf], hop, frames, d_out[i % nbuf], feedblocks=2, stream=streams[i % S].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(streams=S, us_per_launch=round(dt / K * 1e6, 2), Mframes_s=round(K * frames / dt / 1e6, 1),
                          GBs=round(K * frames * 4100 / dt / 1e9, 1))), flush=True)
