#!/usr/bin/env python3
"""DEV TOOL: in-kernel s_memtime stamps of the C2 launch (variant 'S'): where does a wave's lifetime go?"""
import os, sys, ctypes
os.environ["JSG_1024_VARIANT"] = "S"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jadespectrogram_amd as jsg
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, hop = 1024, 512
H = 513; pitch = 544
plan = jsg.Plan(n, jsg.window(1, n))
nbuf = 12
d_in = [torch.rand((1, frames * hop + n - hop), device="cuda") * 2 - 1 for _ in range(nbuf)]
d_out = [torch.empty((frames, pitch), device="cuda") for _ in range(nbuf)]
nw = (frames + 7) // 8 * 8
stamps = torch.zeros((nw * 6,), dtype=torch.int64, device="cuda")
lib = jsg.capi.lib()
lib.jsg_dev_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.jsg_dev_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
for i in range(20):
    jsg.stft_db(plan, d_in[i % nbuf], hop, frames, d_out[i % nbuf], feedblocks=2)
torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(-1, 6).astype(np.float64)
nblk = nw // 8
print("waves", len(raw), "cycles (s_memtime = shader clock); us assume 2.1 GHz")
US = 1.0 / 2100.0
blk = np.repeat(np.arange(nblk), 8)[:len(raw)]
ok = raw[:, 3] > 0
for name, a, b in (("load wait", 0, 1), ("compute", 1, 2), ("store+drain", 2, 3), ("lifetime", 0, 3)):
    v = (raw[ok, b] - raw[ok, a])
    print(f"{name:12s} cycles: min {v.min():7.0f} median {np.median(v):7.0f} p90 {np.percentile(v,90):7.0f} max {v.max():7.0f}   (median {np.median(v)*US:5.2f} us)")
# absolute timeline from s_memrealtime (100 MHz, chip-wide)
r0 = (raw[ok, 4] - raw[ok, 4].min()) / 100.0
r1 = (raw[ok, 5] - raw[ok, 4].min()) / 100.0
cyc_per_us = np.median((raw[ok, 3] - raw[ok, 0]) / np.maximum(r1 - r0, 1e-9))
print(f"shader clock ~ {cyc_per_us:.0f} cycles/us")
print(f"wave start  (us since first wave): p1 {np.percentile(r0,1):5.2f} median {np.median(r0):5.2f} p99 {np.percentile(r0,99):5.2f} max {r0.max():5.2f}")
print(f"wave end    (us since first wave): p1 {np.percentile(r1,1):5.2f} median {np.median(r1):5.2f} p99 {np.percentile(r1,99):5.2f} max {r1.max():5.2f}")
data_at = r0 + (raw[ok,1]-raw[ok,0]) / cyc_per_us
comp_at = r0 + (raw[ok,2]-raw[ok,0]) / cyc_per_us
print(f"data arrived: p1 {np.percentile(data_at,1):5.2f} median {np.median(data_at):5.2f} p99 {np.percentile(data_at,99):5.2f}")
print(f"compute done: p1 {np.percentile(comp_at,1):5.2f} median {np.median(comp_at):5.2f} p99 {np.percentile(comp_at,99):5.2f}")
