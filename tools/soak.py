#!/usr/bin/env python3
"""DEV TOOL: soak / determinism check.  Every plan is launched thousands of times on several concurrent HIP streams
(different timing, co-resident workgroups of different launches); each output must stay bit-identical to the first."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
from jadespectrogram_amd import capi
from jadespectrogram_amd.spectrogram import _stft_args
lib = capi.lib()
for n, frames, ch, sel in ((512, 3001, 1, 0), (1024, 4096, 1, 0), (1024, 1000, 3, 0), (2048, 2049, 2, 0), (2048, 2049, 5, 2), (4096, 1025, 1, 0),
                           (4096, 1025, 3, 2), (8192, 513, 2, 0)):   # sel 2: the "B" kernels of 2048 / 4096 points
    hop = n // 4
    H = n // 2 + 1; pitch = (H + 31) // 32 * 32
    plan = jsg.Plan(n, jsg.window(2, n))
    S, B = 6, 12
    xs = [torch.rand((ch, (frames - 1) * hop + n), device="cuda") * 2 - 1 for _ in range(B)]
    ref = [torch.empty((frames, pitch), device="cuda") for _ in range(B)]
    out = [torch.empty((frames, pitch), device="cuda") for _ in range(B)]
    for b in range(B):
        jsg.stft_db(plan, xs[b], hop, frames, ref[b], feedblocks=4, plan_select=sel)
    torch.cuda.synchronize()
    reps = 150
    arr = (capi.StftArgs * (B * reps))()
    for i in range(B * reps):
        a = _stft_args(plan, xs[i % B], hop, frames, out[i % B], feedblocks=4, blocks_per_cu=(i // B) % 3, plan_select=sel)
        C.memmove(C.byref(arr, i * C.sizeof(capi.StftArgs)), C.byref(a), C.sizeof(capi.StftArgs))
    streams = [torch.cuda.Stream() for _ in range(S)]
    sarr = (C.c_void_p * S)(*[s.cuda_stream for s in streams])
    t0 = time.perf_counter()
    bad = 0
    for r in range(6):
        capi.check(lib.jsg_stft_db_launch_many(plan._p, arr, B * reps, sarr, S))
        torch.cuda.synchronize()
        for b in range(B):
            if not torch.equal(out[b][:, :H], ref[b][:, :H]):
                bad += 1
            out[b].zero_()
    print(json.dumps(dict(n=n, frames=frames, channels=ch, kernel=sel, launches=6 * B * reps, mismatching_buffers=bad,
                          seconds=round(time.perf_counter() - t0, 2))), flush=True)
    assert bad == 0
# round 6: the strided dispatches -- the "runs" kernel (1024 points, one channel per column, 50 % overlap: C2 and the per-channel C4 shard) and the
# two-stage plan Cfg1024B (8 channels AbsMean) -- repeated on two streams at once; every repetition bit-identical to the first
for label, ch, mix, K, hop_div in (("c2 runs", 1, capi.MIX_ABSMEAN, 24, 2), ("c4 shard runs", 8, capi.MIX_PER_CHANNEL, 3, 2), ("Cfg1024B 8 ch AbsMean", 8, capi.MIX_ABSMEAN, 6, 2)):
    n, frames = 1024, 4096
    hop = n // hop_div
    H = n // 2 + 1; pitch = (H + 31) // 32 * 32
    plan = jsg.Plan(n, jsg.window(1, n))
    x = torch.rand((K, ch, (frames - 1) * hop + n), device="cuda") * 2 - 1
    shape = (K, ch, frames, pitch) if mix == capi.MIX_PER_CHANNEL else (K, frames, pitch)
    ref = torch.empty(shape, device="cuda")
    kw = dict(feedblocks=n // hop, mix_mode=mix)
    name = jsg.stft_db_strided_kernel_name(plan, x, hop, frames, ref, **kw)
    jsg.stft_db_strided(plan, x, hop, frames, ref, **kw)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [torch.empty(shape, device="cuda") for _ in range(2)]
    t0 = time.perf_counter()
    bad, launches = 0, 0
    for r in range(150):
        for si, st in enumerate(streams):
            for _ in range(4):
                jsg.stft_db_strided(plan, x, hop, frames, outs[si], stream=st.cuda_stream, blocks_per_cu=(r % 3) * 8, **kw)
                launches += 1
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o[..., :H], ref[..., :H]):
                bad += 1
            o.zero_()
    print(json.dumps(dict(strided=label, kernel=name, batches=K, channels=ch, dispatches=launches, mismatching_buffers=bad, seconds=round(time.perf_counter() - t0, 2))), flush=True)
    assert bad == 0
print("soak ok")
