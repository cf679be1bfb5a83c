#!/usr/bin/env python3
"""DEV TOOL: host time to issue one prepared launch vs GPU time per launch (is the C2 loop host-bound?)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
n, hop, frames = 1024, 512, 4096
plan = jsg.Plan(n, jsg.window(1, n))
nbuf = 24
d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
d_out = [torch.empty((frames, 544), device="cuda") for _ in range(nbuf)]
L = [jsg.StftLaunch(plan, d_in[b], hop, frames, d_out[b], feedblocks=2) for b in range(nbuf)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(200): L[i % nbuf].launch(st)
torch.cuda.synchronize()
K = 3000
t0 = time.perf_counter()
for i in range(K): L[i % nbuf].launch(st)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps(dict(host_issue_us=round((t1 - t0) / K * 1e6, 2), total_us_per_launch=round((t2 - t0) / K * 1e6, 2))))
