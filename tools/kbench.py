#!/usr/bin/env python3
"""DEV TOOL: kernel-only timing sweeps of the fused STFT->dB kernel (frames per launch, fft size, channels)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jadespectrogram_amd as jsg

def run(n, hop, frames, channels, steps, nbuf=None, mix=0, warm=20):
    H = n // 2 + 1
    pitch = (H + 31) // 32 * 32
    n_samples = frames * hop + (n - hop)
    per = channels * n_samples * 4 + frames * pitch * 4
    nbuf = nbuf or max(2, int(300e6 // per) + 1)
    win = jsg.window(1, n); plan = jsg.Plan(n, win)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    d_in = [torch.rand((channels, n_samples), device="cuda", generator=g) * 2 - 1 for _ in range(nbuf)]
    d_out = [torch.empty((frames, pitch), device="cuda") for _ in range(nbuf)]
    import ctypes, time
    fb = max(1, n // hop)
    L = [jsg.StftLaunch(plan, d_in[b], hop, frames, d_out[b], feedblocks=fb, mix_mode=mix) for b in range(nbuf)]
    if not os.environ.get("JSG_KBENCH_GRAPH"):
        # eager launches on one stream (what bench.py does); host issue cost ~4.6 us per launch bounds small launches
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(warm): L[i % nbuf].launch(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): L[i % nbuf].launch(st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / steps
        return _result(n, hop, frames, channels, us, nbuf)
    # JSG_KBENCH_GRAPH=1: GPU-side time per launch with the host taken out (one hipGraph holding a rotation of the
    # buffers; graph nodes run strictly one after the other, so this reads ~0.3-2 us above back-to-back eager launches)
    s2 = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    reps_in_graph = max(1, 24 // nbuf)
    with torch.cuda.stream(s2):
        st = ctypes.c_void_p(s2.cuda_stream)
        for b in range(nbuf): L[b].launch(st)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s2):
            for _ in range(reps_in_graph):
                for b in range(nbuf): L[b].launch(st)
    torch.cuda.synchronize()
    per_graph = reps_in_graph * nbuf
    n_rep = max(3, steps // per_graph)
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_rep): g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (n_rep * per_graph) * 1e6
    return _result(n, hop, frames, channels, us, nbuf)


def _result(n, hop, frames, channels, us, nbuf):
    H = n // 2 + 1
    ffts = frames * channels
    algo = (4 * hop * channels + 4 * H) * frames
    return dict(n=n, hop=hop, frames=frames, ch=channels, us=round(us, 2), Mfft_s=round(ffts / us, 1),
                GBs=round(algo / us / 1e3, 1), frac=round(algo / us / 1e3 / 8000, 4), nbuf=nbuf)

if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--set", default="frames")
    a = ap.parse_args()
    if a.set == "frames":
        for fr in (1024, 2048, 4096, 8192, 16384, 65536, 262144):
            print(json.dumps(run(1024, 512, fr, 1, max(50, min(2000, 4000000 // fr)))), flush=True)
    elif a.set == "sizes":
        for n in (512, 1024, 2048, 4096, 8192):
            print(json.dumps(run(n, n // 2, 65536 * 1024 // n, 1, 50)), flush=True)
        print(json.dumps(run(2048, 512, 16384, 8, 50)), flush=True)      # C3
        print(json.dumps(run(1024, 512, 4096, 8, 200)), flush=True)      # C4 shard, mixed
        print(json.dumps(run(4096, 512, 8192, 2, 50)), flush=True)       # C5 stft part
