#!/usr/bin/env python3
"""DEV TOOL: is the C2 loop host-launch-bound?  eager ctypes launches vs one hipGraph holding 24 launches."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
n, hop, frames = 1024, 512, 4096
plan = jsg.Plan(n, jsg.window(1, n))
nbuf = 24
d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
d_out = [torch.empty((frames, 544), device="cuda") for _ in range(nbuf)]
st = torch.cuda.current_stream().cuda_stream
def launch(i, stream):
    jsg.stft_db(plan, d_in[i % nbuf], hop, frames, d_out[i % nbuf], feedblocks=2, stream=stream)
K = 2400
for i in range(100): launch(i, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K): launch(i, st)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps(dict(mode="eager", host_issue_us=round(t_issue / K * 1e6, 2), total_us_per_launch=round(t_all / K * 1e6, 2))), flush=True)
g = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    for i in range(10): launch(i, s2.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s2):
        for i in range(nbuf): launch(i, s2.cuda_stream)
torch.cuda.synchronize()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
R = K // nbuf
for _ in range(R): g.replay()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(json.dumps(dict(mode="graph(24 launches)", total_us_per_launch=round(t_all / (R * nbuf) * 1e6, 2),
                      Mframes_s=round(R * nbuf * frames / t_all / 1e6, 1))), flush=True)
