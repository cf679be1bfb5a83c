#!/usr/bin/env python3
"""DEV TOOL: what does it take just to MOVE the bytes of one launch?  torch's copy kernel and a tuned float4
streaming copy (with / without non-temporal stores) for the C2 byte count and for larger batches."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
lib = jsg.capi.lib()
lib.jsg_dev_copy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
def timeit(fn, steps):
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
for nbytes in (8_396_800, 33_587_200, 134_348_800):
    n = nbytes // 4
    nbuf = max(2, int(600e6 // (2 * nbytes)) + 1)
    src = [torch.rand(n, device="cuda") for _ in range(nbuf)]; dst = [torch.empty(n, device="cuda") for _ in range(nbuf)]
    st = torch.cuda.current_stream().cuda_stream
    res = dict(bytes_each_way=nbytes)
    res["torch_copy_us"] = round(timeit(lambda i: dst[i % nbuf].copy_(src[i % nbuf]), 300), 2)
    best = None
    for blocks in (256, 512, 1024, 2048, 4096):
        for nt in (0, 1):
            us = timeit(lambda i: lib.jsg_dev_copy_launch(src[i % nbuf].data_ptr(), dst[i % nbuf].data_ptr(), nbytes, blocks, nt, st), 300)
            if best is None or us < best[0]: best = (us, blocks, nt)
    res["tuned_copy_us"], res["tuned_blocks"], res["tuned_nt"] = round(best[0], 2), best[1], best[2]
    res["tuned_TBs"] = round(2 * nbytes / best[0] / 1e6, 2)
    print(json.dumps(res), flush=True)
