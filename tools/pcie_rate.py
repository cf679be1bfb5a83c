#!/usr/bin/env python3
"""DEV TOOL: PCIe-inclusive rate -- the engine fed from HOST buffers (jsg_process_blocks: H2D copy + kernel), C2 shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jadespectrogram_amd as jsg
n, blocks = 1024, 2048                      # 2048 blocks of 1024 samples = 4096 frames at 50 % hop
s = jsg.Spectrogram(1)
s.setSamplerate(48000.0); s.setmemoryTime_s(60.0); s.setFFTSize(n); s.setfeed_percent(1)
x = (np.random.default_rng(0).uniform(-1, 1, (1, blocks * n))).astype(np.float32)
import torch
xp = torch.from_numpy(x).pin_memory().numpy()
for name, buf in (("pageable", x), ("pinned", xp)):
    for _ in range(5): s.processBlocks(buf)
    s.sync()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): s.processBlocks(buf)
    s.sync()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(host_memory=name, frames_per_s=round(K * blocks * 2 / dt), us_per_batch=round(dt / K * 1e6, 1),
                          h2d_GBs=round(K * x.nbytes / dt / 1e9, 1))), flush=True)
