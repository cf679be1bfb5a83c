#!/usr/bin/env python3
"""Headline benchmark: STFT frames/sec (1024-pt, 50 % hop) on N MI355X, with the kernel's HBM roofline fraction.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" is one launch of the fused STFT->dB kernel over one batch of BASELINE.json configs[1]:
mono 48 kHz, 1024-point FFT, hop 512, Hann, 4096 frames per launch, input stream and dB ring resident in HBM.
Every rank works on its own independent streams (weak scaling, no data-path collective: the path shards by
channel/stream); RCCL is used only for the barriers and the max-over-ranks time.

To keep the numbers honest against the 256 MiB Infinity Cache, the steps rotate over NBUF distinct input/output
batches (> 256 MiB in total), so every launch streams its 8.4 MB in and 8.4 MB out from/to HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FFT, HOP, FRAMES = 1024, 512, 4096
H = N_FFT // 2 + 1
ALGO_BYTES_PER_FRAME = 4 * HOP + 4 * H          # input counted once + one dB column = 4100 B
HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_audio(channels: int, n_samples: int, fs: float = 48000.0, seed: int = 1234):
    """SURVEY 8d synthetic input: x_c[n] = 0.5 sin(2 pi f_c n / fs) + 0.1 u[n], f_c = 220*2^(c/12), u ~ U(-1,1)."""
    import numpy as np
    out = np.zeros((channels, n_samples), dtype=np.float32)
    t = np.arange(n_samples, dtype=np.float64)
    for c in range(channels):
        u = np.random.default_rng(seed + c).uniform(-1.0, 1.0, n_samples)
        out[c] = (0.5 * np.sin(2.0 * np.pi * 220.0 * 2.0 ** (c / 12.0) * t / fs) + 0.1 * u).astype(np.float32)
    return out


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget: float = 12.0, check=None):
    """The oracle's CPU path timed on this host (rank 0, N=1 only): reported baseline, not the target.
    This leg is the only place where bench.py touches oracle/ (test infrastructure); `check` = (samples, dB) of a few
    frames the GPU just produced, compared against the oracle before timing it."""
    import numpy as np
    from oracle import jsg_oracle as oracle
    try:
        from oracle import oracle_c
        port = oracle_c.load()
    except Exception:
        port = None
    win = oracle.window(oracle.WIN_HANN, N_FFT)
    if check is not None:
        x, got = check
        fr = np.stack([x[j * HOP:j * HOP + N_FFT] * win for j in range(got.shape[0])]).astype(np.float32)
        assert np.abs(got - oracle.to_db(oracle.power_spectrum(fr))).max() < 5e-3, "bench output drifted from the oracle"
    if port is not None:
        frames = 20000
        x = synth_audio(1, frames * HOP + N_FFT)
        t0 = time.perf_counter(); done = 0
        while time.perf_counter() - t0 < seconds_budget:
            port.stft_db(x, N_FFT, HOP, frames, win)
            done += frames
        dt = time.perf_counter() - t0
        res = {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{done} frames of the bench workload through oracle/jsg_oracle_c.c (scalar float32 C port "
                         f"of Spectrogram::processSynchronBlock, 1 thread) in {dt:.1f} s",
               "cpu_model": _cpu_model()}
        # the same port with OpenMP over frames on every core this process may use (1 thread and all cores)
        cores = min(len(os.sched_getaffinity(0)), 64)
        if cores > 1:
            t0 = time.perf_counter(); done = 0
            while time.perf_counter() - t0 < seconds_budget / 3:
                port.stft_db(x, N_FFT, HOP, frames, win, threads=cores)
                done += frames
            dt = time.perf_counter() - t0
            res["all_cores"] = {"value": done / dt, "unit": "frames/s", "cores": cores,
                                "sample": f"{done} frames, OpenMP over frames, in {dt:.1f} s"}
        return res
    frames = 8192
    x = synth_audio(1, frames * HOP + N_FFT)
    t0 = time.perf_counter(); done = 0
    while time.perf_counter() - t0 < seconds_budget:
        oracle.stft_db_reference(x[:, N_FFT:], N_FFT, HOP, 2, win)
        done += (x.shape[1] - N_FFT) // N_FFT * 2
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} frames through oracle/jsg_oracle.py (numpy float64 rfft restatement) in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--nbuf", type=int, default=24, help="distinct batches rotated through (24 x 16.8 MB > 256 MiB)")
    ap.add_argument("--streams", type=int, default=8, help="HIP streams the independent launches are spread over")
    ap.add_argument("--blocks-per-cu", type=int, default=1, help="workgroups per CU of each launch in the concurrent pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import jadespectrogram_amd as jsg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # JSG_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then share devices);
    # the driver's runs use the default: one rank per GPU, RCCL ("nccl") over xGMI
    backend = os.environ.get("JSG_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    # ---- workload: NBUF independent mono batches per rank, resident in HBM ----
    win = jsg.window(jsg.capi.WIN_HANN, N_FFT)
    plan = jsg.Plan(N_FFT, win)
    n_samples = FRAMES * HOP + (N_FFT - HOP)
    pitch = (H + 31) // 32 * 32
    d_in, d_out = [], []
    base = synth_audio(1, n_samples + args.nbuf * 64, seed=1234 + 1000 * rank)   # SURVEY 8d signal
    for b in range(args.nbuf):
        d_in.append(torch.from_numpy(np.ascontiguousarray(base[:, b * 64:b * 64 + n_samples])).cuda())
        d_out.append(torch.empty((FRAMES, pitch), dtype=torch.float32, device="cuda"))
    # The K steps of the timed region are K independent launches (step i works on batch i % nbuf).  They are issued
    # by ONE C call (jsg_stft_db_launch_many: no per-step FFI cost) round-robin on `n_streams` HIP streams, so the
    # dispatch ramp / first-data latency / store drain of one launch overlap the compute of the others.  A second
    # pass issues the same K steps in order on a single stream: that is the classic per-kernel view used for
    # `roofline` (HIP events on the stream the kernel runs on; agrees with rocprofv3's per-dispatch duration).
    import ctypes
    from jadespectrogram_amd import capi
    from jadespectrogram_amd.spectrogram import _stft_args
    lib = capi.lib()
    n_streams = max(1, args.streams)
    while args.nbuf % n_streams:        # a batch must always land on the same stream (its ring is rewritten in order)
        n_streams -= 1
    total = max(args.steps, args.warmup)
    # concurrent pass: one workgroup per CU, two frames per wavefront with prefetch (fewer, longer-lived workgroups
    # overlap better across launches); in-order pass: the default geometry (best for one launch alone)
    arr = (capi.StftArgs * total)()
    arr_inorder = (capi.StftArgs * total)()
    for i in range(total):
        for dst_arr, bpc in ((arr, args.blocks_per_cu if n_streams > 1 else 0), (arr_inorder, 0)):
            a_i = _stft_args(plan, d_in[i % args.nbuf], HOP, FRAMES, d_out[i % args.nbuf], feedblocks=2, blocks_per_cu=bpc)
            ctypes.memmove(ctypes.byref(dst_arr, i * ctypes.sizeof(capi.StftArgs)), ctypes.byref(a_i), ctypes.sizeof(capi.StftArgs))
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    sarr = (ctypes.c_void_p * n_streams)(*[st.cuda_stream for st in streams])
    one = torch.cuda.Stream()
    one_arr = (ctypes.c_void_p * 1)(one.cuda_stream)

    def run(count, handles, n, which=None):
        capi.check(lib.jsg_stft_db_launch_many(plan._p, which if which is not None else arr, count, handles, n))

    def barrier():
        if dist is not None:
            dist.barrier()

    torch.cuda.synchronize()
    run(args.warmup, sarr, n_streams)                       # W untimed warm-up steps
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, sarr, n_streams)                        # EXACTLY K timed steps
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier(); torch.cuda.synchronize()
    wall = t1 - t0

    # in-order pass for the per-kernel roofline (not part of `value`)
    run(min(args.warmup, 50), one_arr, 1, arr_inorder)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(one)
    run(args.steps, one_arr, 1, arr_inorder)
    ev1.record(one)
    torch.cuda.synchronize()
    ev_ms = ev0.elapsed_time(ev1)                       # events on the stream the kernel runs on
    barrier(); torch.cuda.synchronize()
    if dist is not None:
        t = torch.tensor([wall, ev_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])

    # what was just timed, for the parity spot check inside the cpu_baseline leg
    check = None
    if rank == 0:
        b = (args.steps - 1) % args.nbuf
        check = (d_in[b][0, :N_FFT + 3 * HOP].cpu().numpy(), d_out[b][:4, :H].cpu().numpy())

    # context for the roofline: a plain device-to-device copy of the SAME byte count (8.4 MB in, 8.4 MB out per launch),
    # rotating over the same number of distinct buffers -- what this launch size can reach at all on this GPU
    copy_us = None
    if rank == 0:
        nflt = ALGO_BYTES_PER_FRAME * FRAMES // 8
        csrc = [torch.rand(nflt, device="cuda") for _ in range(args.nbuf)]
        cdst = [torch.empty(nflt, device="cuda") for _ in range(args.nbuf)]
        for i in range(50):
            cdst[i % args.nbuf].copy_(csrc[i % args.nbuf])
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for i in range(500):
            cdst[i % args.nbuf].copy_(csrc[i % args.nbuf])
        c1.record()
        torch.cuda.synchronize()
        copy_us = c0.elapsed_time(c1) * 1e3 / 500
        del csrc, cdst

    frames_total = world * args.steps * FRAMES
    # per-kernel view: HIP events over the K in-order launches on one stream (rocprofv3's per-dispatch duration,
    # profiles/r01_kernel_stats.csv, reads ~0.5 us longer: the tracer runs the dispatches isolated)
    kernel_s = ev_ms * 1e-3 / args.steps
    achieved = ALGO_BYTES_PER_FRAME * FRAMES / kernel_s / 1e9
    traffic, rocprof_us = None, None
    prof = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            traffic = pj.get("hbm_bytes_per_launch")            # PMC FETCH_SIZE x2 (gfx950) + WRITE_SIZE, per launch
            rocprof_us = pj.get("avg_ns", 0.0) / 1e3 or None    # rocprofv3 --kernel-trace --stats of this command
        except Exception:
            traffic = None
    out = {
        "metric": "STFT frames/sec (1024-pt, 50% hop)", "value": frames_total / wall, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: mono 48 kHz, 1024-pt FFT, 512 hop, Hann, 4096 frames/launch, "
                               "input+dB ring resident in HBM", "frames_per_launch": FRAMES, "channels_per_gpu": 1,
                   "distinct_batches": args.nbuf, "hip_streams_per_gpu": n_streams,
                   "parallelism": f"{world} GPU(s), independent batches, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "stft_db_kernel<1024>", "avg_launch_us": kernel_s * 1e6,
                     "rocprof_isolated_dispatch_us": rocprof_us,
                     "concurrent_achieved": ALGO_BYTES_PER_FRAME * FRAMES * args.steps / wall / 1e9,
                     "concurrent_frac": ALGO_BYTES_PER_FRAME * FRAMES * args.steps / wall / 1e9 / HBM_PEAK_GBS,
                     "note": "achieved/frac: one launch at a time on one stream; concurrent_*: the timed region itself "
                             "(independent launches overlapped on hip_streams_per_gpu streams)",
                     "memcpy_same_bytes_us": copy_us,
                     "frac_of_memcpy_rate": (copy_us / (kernel_s * 1e6)) if copy_us else None,
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * FRAMES},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(check=check)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
