#!/usr/bin/env python3
"""C5 images (stereo, 4096 points, hop 512, 1875 columns -> ARGB): K separate jsg_stft_image_launch calls in order against ONE
jsg_stft_image_launch_strided call over the same K images (K image sets = ~1 GB, so every launch streams from / to HBM).
Usage: python tools/image_batch_probe.py [K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jadespectrogram_amd as jsg
from bench import synth_audio

n, hop, C, F = 4096, 512, 2, 1875
H = n // 2 + 1
pitch = (F + 31) // 32 * 32
ns = F * hop + n - hop
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
lut = torch.from_numpy(jsg.colormap_lut(256, jsg.capi.CM_JADE)).cuda()
Ks = [int(a) for a in sys.argv[1:]] or [43]
for K in Ks:
    base = synth_audio(C, ns + K * 64, fs=96000.0, seed=1234)
    d_in = torch.stack([torch.from_numpy(np.ascontiguousarray(base[:, b * 64:b * 64 + ns])) for b in range(K)]).cuda()
    img = torch.zeros((K, H, pitch), dtype=torch.int32, device="cuda")
    ref = torch.zeros((K, H, pitch), dtype=torch.int32, device="cuda")
    kw = dict(feedblocks=n // hop, mix_mode=jsg.capi.MIX_ABSMEAN)
    st = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        def singles(dst):
            for k in range(K):
                jsg.stft_image(plan, d_in[k], hop, F, lut, -50.0, 50.0, dst[k][:, :F], None, stream=st.cuda_stream, **kw)
        def batch(dst):
            jsg.stft_image_strided(plan, d_in, hop, F, lut, -50.0, 50.0, dst[:, :, :F], None, stream=st.cuda_stream, **kw)
        res = {}
        for name, fn, dst in (("singles", singles, ref), ("strided", batch, img)):
            g = torch.cuda.CUDAGraph()
            fn(dst); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                fn(dst)
            for _ in range(3): g.replay()
            torch.cuda.synchronize()
            reps = 20
            e0.record(st)
            for _ in range(reps): g.replay()
            e1.record(st)
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / (reps * K)
    same = bool(torch.equal(img, ref))
    algo = F * (C * hop * 4 + H * 4)
    print(f"K={K}: singles in order {res['singles']:.2f} us/image ({algo / res['singles'] / 8e6:.3f} of 8 TB/s), one strided launch "
          f"{res['strided']:.2f} us/image ({algo / res['strided'] / 8e6:.3f}), identical pixels: {same}", flush=True)
    del d_in, img, ref
