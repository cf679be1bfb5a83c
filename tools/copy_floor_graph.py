#!/usr/bin/env python3
"""DEV TOOL: GPU-side time per launch with the host taken out (hipGraph of 24 launches): tuned copy vs STFT kernel."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
lib = jsg.capi.lib()
lib.jsg_dev_copy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
nbuf = 24
def graph_time(issue, reps=100):
    s2 = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        for i in range(nbuf): issue(i, s2.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s2):
            for i in range(nbuf): issue(i, s2.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * nbuf) * 1e6
nbytes = 8_396_800
src = [torch.rand(nbytes // 4, device="cuda") for _ in range(nbuf)]; dst = [torch.empty(nbytes // 4, device="cuda") for _ in range(nbuf)]
for blocks in (1024, 2048, 4096):
    us = graph_time(lambda i, st: lib.jsg_dev_copy_launch(src[i].data_ptr(), dst[i].data_ptr(), nbytes, blocks, 1, ctypes.c_void_p(st)))
    print(json.dumps(dict(kernel="tuned nt copy 8.4MB+8.4MB", blocks=blocks, us_per_launch=round(us, 2), TBs=round(2 * nbytes / us / 1e6, 2))), flush=True)
n, hop, frames = 1024, 512, 4096
plan = jsg.Plan(n, jsg.window(1, n))
d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
d_out = [torch.empty((frames, 544), device="cuda") for _ in range(nbuf)]
L = [jsg.StftLaunch(plan, d_in[b], hop, frames, d_out[b], feedblocks=2) for b in range(nbuf)]
us = graph_time(lambda i, st: L[i].launch(ctypes.c_void_p(st)))
print(json.dumps(dict(kernel="stft_db C2", us_per_launch=round(us, 2), TBs_algorithmic=round(4100 * frames / us / 1e6, 2))), flush=True)
