#!/usr/bin/env python3
"""DEV TOOL: accuracy of the GPU power spectrum against the float64 DFT of the float32 windowed frame (numpy only
as the yardstick), per FFT size and kernel (plan_select 1 = small-workgroup kernel, 2 = "B" kernel where one exists): the worst
absolute error relative to the frame peak (what FLOOR in tests/parity_util.py bounds), the worst relative error of bins within
20 dB of the peak, and the share of bins beyond plain 1e-5.  Signals: sine + noise, pure noise, a full-scale sine, a chirp; all
six windows; 1, 2, 3 and 8 channels (AbsMean).  Prints one JSON line per (n, kernel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jadespectrogram_amd as jsg
from oracle import jsg_oracle as oracle   # (dev tool: the float32 channel mix of the reference)

def signals(C, ns, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(ns)
    out = []
    base = np.stack([0.5 * np.sin(2 * np.pi * 220.0 * 2 ** (c / 12) * t / 48000.0) + 0.1 * rng.uniform(-1, 1, ns) for c in range(C)])
    out.append(("sine+noise", base))
    out.append(("noise", rng.uniform(-1, 1, (C, ns))))
    out.append(("full-scale sine", np.stack([np.sin(2 * np.pi * (997.0 + 31 * c) * t / 48000.0) for c in range(C)])))
    out.append(("chirp", np.stack([0.8 * np.sin(2 * np.pi * (50.0 + (8000.0 + 500 * c) * t / ns) * t / 48000.0) for c in range(C)])))
    return [(k, v.astype(np.float32)) for k, v in out]

for n in (512, 1024, 2048, 4096, 8192):
    for sel in (1, 2):
        if sel == 2 and n not in (2048, 4096):
            continue
        worst_abs, worst_rel20, frac_bad, worst_case = 0.0, 0.0, 0.0, None
        for wkind in range(6):
            win = jsg.window(wkind, n)
            plan = jsg.Plan(n, win)
            for C in (1, 2, 3, 8):
                for hop in (n // 2, n // 4):
                    F = 96
                    for name, x in signals(C, (F - 1) * hop + n, n + 7 * C + wkind):
                        H = n // 2 + 1
                        out = torch.empty((F, (H + 31) // 32 * 32), device="cuda")
                        jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, F, out, feedblocks=n // hop, linear_out=True, plan_select=sel)
                        torch.cuda.synchronize()
                        got = out[:, :H].cpu().numpy().astype(np.float64)
                        idx = (np.arange(F) * hop)[:, None] + np.arange(n)[None, :]
                        fr = (x[:, idx] * win[None, None, :]).astype(np.float32)
                        p64 = oracle.power_spectrum_f64(fr)
                        ref = oracle.mix_channels(p64.astype(np.float32), oracle.MIX_ABSMEAN).astype(np.float64)
                        peak = ref.max(axis=1, keepdims=True)
                        ok = peak[:, 0] > 0
                        a = float((np.abs(got - ref)[ok] / peak[ok]).max())
                        rel = np.abs(got - ref) / np.maximum(ref, 1e-300)
                        m20 = ref > 1e-2 * peak
                        r20 = float(rel[m20].max())
                        if a > worst_abs:
                            worst_abs, worst_case = a, dict(window=wkind, channels=C, hop=hop, signal=name)
                        worst_rel20 = max(worst_rel20, r20)
                        frac_bad = max(frac_bad, float((rel > 1e-5).mean()))
        print(json.dumps(dict(n=n, plan_select=sel, max_abs_err_over_frame_peak=worst_abs, worst_case=worst_case,
                              max_rel_err_within_20dB=worst_rel20, max_share_of_bins_beyond_1e5=frac_bad)), flush=True)
