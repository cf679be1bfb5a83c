#!/usr/bin/env python3
"""DEV TOOL: accuracy of the GPU power spectrum against the float64 DFT of the float32 windowed frame (numpy only
as the yardstick).  Prints per FFT size: median / max relative error over bins within 20 dB and 60 dB of the frame peak,
and the max absolute error relative to the peak."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jadespectrogram_amd as jsg
for n in (512, 1024, 2048, 4096, 8192):
    hop, F = n // 2, 64
    rng = np.random.default_rng(n)
    t = np.arange(F * hop + n)
    x = (0.5 * np.sin(2 * np.pi * 997.0 * t / 48000.0) + 0.1 * rng.uniform(-1, 1, t.size)).astype(np.float32)[None]
    win = jsg.window(1, n)
    plan = jsg.Plan(n, win)
    out = torch.empty((F, n // 2 + 1 + 31), device="cuda")
    jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, F, out, linear_out=True)
    torch.cuda.synchronize()
    got = out[:, :n // 2 + 1].cpu().numpy().astype(np.float64)
    idx = (np.arange(F) * hop)[:, None] + np.arange(n)[None, :]
    fr = (x[0][idx] * win[None, :]).astype(np.float32).astype(np.float64)
    X = np.fft.rfft(fr, axis=-1); ref = X.real ** 2 + X.imag ** 2
    peak = ref.max(axis=1, keepdims=True)
    rel = np.abs(got - ref) / ref
    m20, m60 = ref > 1e-2 * peak, ref > 1e-6 * peak
    print(json.dumps(dict(n=n, median_rel=float(np.median(rel)), max_rel_within_20dB=float(rel[m20].max()),
                          max_rel_within_60dB=float(rel[m60].max()), max_abs_over_peak=float((np.abs(got - ref) / peak).max()))), flush=True)
