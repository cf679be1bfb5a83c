#!/usr/bin/env python3
"""DEV TOOL: C2 launches issued from C (jsg_stft_db_launch_many) on 1..8 streams: interval per launch."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
from jadespectrogram_amd import capi
from jadespectrogram_amd.spectrogram import _stft_args
n, hop, frames = 1024, 512, int(os.environ.get("FRAMES", "4096"))   # FRAMES=8: pure host + command-processor issue rate
plan = jsg.Plan(n, jsg.window(1, n))
nbuf = 24
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
if mode == "rand":
    d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
    d_out = [torch.empty((frames, 544), device="cuda") for _ in range(nbuf)]
else:   # like bench.py: the SURVEY 8d signal, input and output buffers allocated alternately
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synth_audio
    ns = frames * hop + n - hop
    base = synth_audio(1, ns + nbuf * 64)
    d_in, d_out = [], []
    for b in range(nbuf):
        if mode == "bench_rand":
            d_in.append(torch.rand((1, ns), device="cuda") * 2 - 1)
        else:
            d_in.append(torch.from_numpy(np.ascontiguousarray(base[:, b * 64:b * 64 + ns])).cuda())
        d_out.append(torch.empty((frames, 544), device="cuda"))
K = 2400
arr = (capi.StftArgs * K)()
for i in range(K):
    a = _stft_args(plan, d_in[i % nbuf], hop, frames, d_out[i % nbuf], feedblocks=2)
    C.memmove(C.byref(arr, i * C.sizeof(capi.StftArgs)), C.byref(a), C.sizeof(capi.StftArgs))
lib = capi.lib()
for S in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream() for _ in range(S)]
    sarr = (C.c_void_p * S)(*[s.cuda_stream for s in streams])
    lib.jsg_stft_db_launch_many(plan._p, arr, 200, sarr, S); torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(lib.jsg_stft_db_launch_many(plan._p, arr, K, sarr, S))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps(dict(streams=S, host_issue_us=round((t1 - t0) / K * 1e6, 2), us_per_launch=round((t2 - t0) / K * 1e6, 2),
                          Mframes_s=round(K * frames / (t2 - t0) / 1e6, 1), frac=round(K * frames * 4100 / (t2 - t0) / 8e12, 4))), flush=True)
