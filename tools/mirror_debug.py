#!/usr/bin/env python3
"""DEV TOOL: where does the GPU's linear power differ from oracle/jsg_mirror.c?  (per plan: count, bins, ulps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jadespectrogram_amd as jsg
from oracle import jsg_oracle as oracle, mirror as mm
m = mm.load()
for n, sel, kernel in ((1024, 0, "Cfg1024"), (512, 0, "Cfg512"), (2048, 1, "Cfg2048"), (2048, 2, "Cfg2048B"), (4096, 1, "Cfg4096"), (4096, 2, "Cfg4096B"), (8192, 0, "Cfg8192")):
    for win_kind in (0, 1):
        hop, F = n // 2, 16
        x = oracle.synth_audio(1, (F - 1) * hop + n, seed=n)
        win = oracle.window(win_kind, n)
        plan = jsg.Plan(n, win)
        d = torch.zeros((F, (n // 2 + 1 + 31) // 32 * 32), device="cuda")
        jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, F, d, linear_out=True, plan_select=sel)
        torch.cuda.synchronize()
        got = d[:, :n // 2 + 1].cpu().numpy()
        ref = m.columns(kernel, x, hop, F, win)
        bad = got.view(np.uint32) != ref.view(np.uint32)
        ulps = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        bins = np.unique(np.argwhere(bad)[:, 1])
        print(kernel, "window", win_kind, "differing", int(bad.sum()), "of", bad.size, "max ulp", int(ulps.max()), "bins", bins[:12], "..." if len(bins) > 12 else "",
              "count per bin-class: k<M/2", int(bad[:, :n // 4].sum()), "k=M/2", int(bad[:, n // 4].sum()), "k>M/2", int(bad[:, n // 4 + 1:].sum()))
