#!/usr/bin/env python3
"""DEV TOOL: run ONE configuration of the fused kernel a few times (for rocprofv3 --pmc / --kernel-trace)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024); ap.add_argument("--hop", type=int, default=512)
ap.add_argument("--frames", type=int, default=4096); ap.add_argument("--ch", type=int, default=1)
ap.add_argument("--steps", type=int, default=40); ap.add_argument("--nbuf", type=int, default=20)
a = ap.parse_args()
H = a.n // 2 + 1; pitch = (H + 31) // 32 * 32
ns = a.frames * a.hop + a.n - a.hop
plan = jsg.Plan(a.n, jsg.window(1, a.n))
d_in = [torch.rand((a.ch, ns), device="cuda") * 2 - 1 for _ in range(a.nbuf)]
d_out = [torch.empty((a.frames, pitch), device="cuda") for _ in range(a.nbuf)]
for i in range(a.steps):
    jsg.stft_db(plan, d_in[i % a.nbuf], a.hop, a.frames, d_out[i % a.nbuf], feedblocks=max(1, a.n // a.hop))
torch.cuda.synchronize()
print("done")
