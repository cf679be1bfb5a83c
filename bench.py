#!/usr/bin/env python3
"""Headline benchmark: STFT frames/sec (1024-pt, 50 % hop) on N MI355X, with the kernel's HBM roofline fraction.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

What a step is
--------------
A step is a FIXED, stated group of launches of the fused kernel over synthetic batches that are resident in HBM, so the
result does not depend on how many steps the driver asks for:

  c2 (default, BASELINE.json configs[1]): one step = 1024 launches x 4096 frames (mono 48 kHz, 1024-point FFT, hop 512,
      Hann) = 4 194 304 frames.  The launches of a step are independent batches; they are handed to the library in ONE call,
      jsg_stft_db_launch_batches: stream-ordered with respect to the caller's stream like a single launch, spread over the
      caller's stream plus three streams of the library and issued by two host threads, so the ramp-up and drain of one launch
      overlap the others (include/jsg.h).  `value` = frames of all ranks / wall time of the K steps.
      `--streams N` instead issues over N caller-owned streams (jsg_stft_db_launch_many_threads) -- sweeps and the tracer.
  c3 (configs[2]): one step = 64 launches x 4096 columns of 8-channel 2048-point frames, 75 % overlap, AbsMean mix,
      through the same call (two working streams: these kernels run one workgroup per CU).
  c5 (configs[4]): independent images of 1875 columns (10 s), stereo 96 kHz, 4096-point FFT, 87.5 % overlap, fused STFT ->
      palette index -> ARGB image in ONE kernel.  The images have one geometry and lie at a fixed stride, so a step hands them to the
      library as strided batches (jsg_stft_image_launch_strided): one step = 3 launches x a whole rotation of images (~39) =
      ~117 images; the workgroups of a launch walk through the columns of all its images.  `--images-per-launch 1 [--streams S]`
      launches the images one by one instead (jsg_stft_image_launch; 128 per step, three hipGraphs on three streams by default).
  `--streams 1` times every configuration in order on one stream (that is also what `roofline` reports, see below).

The batches rotate over ~1 GB of distinct buffers (about four times the 256 MiB Infinity Cache; `--nbuf` overrides), so every
launch streams from and to HBM.  Before the W warm-up steps the same launches run for about 0.3 s so that the clocks have settled.

`roofline` is the per-kernel view: the same launches, one at a time in order on ONE stream (a hipGraph replay, timed
with HIP events on that stream); achieved = algorithmic bytes per launch / average launch duration.  rocprofv3's
per-dispatch duration of `bench.py --streams 1` (profiles/) is the cross-check.

Every rank works on its own batches (weak scaling; the path shards by channel / stream, there is no data-path
collective); RCCL carries only the barriers and the max-over-ranks time.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues per device (default 4, read when the runtime
# starts) in creation order, and torch's own streams take part: with too few queues two of the bench's busy streams land on ONE
# hardware queue and serialise (round 3, nine streams in the process: 0.79 / 0.41 / 0.84 / 1.08 / 1.08e9 frames/s at 4 / 6 / 8 / 12 /
# 16 queues).  What must stay at four is the number of BUSY queues (a fifth makes the command processor time-slice: 0.35e9).
if not os.environ.get("JSG_KEEP_HW_QUEUES"):      # (libjsg.so applies the same default when it is loaded before the HIP runtime starts)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROTATION_BYTES = 1.0e9       # distinct input + output bytes the timed launches rotate over (>> 256 MiB Infinity Cache)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # ... and about 6.3 TB/s achievable (measured streaming copy)

# name -> workload (algorithmic bytes count every input sample once and every output value once)
CONFIGS = {
    "c2": dict(n=1024, hop=512, channels=1, frames=4096, fs=48000.0, colour=False, launches_per_step=1024, streams=4,
               metric="STFT frames/sec (1024-pt, 50% hop)", unit="frames/s",
               workload="configs[1]: mono 48 kHz, 1024-pt FFT, 512 hop, Hann, 4096 frames/launch, input + dB ring resident in HBM"),
    "c3": dict(n=2048, hop=512, channels=8, frames=4096, fs=48000.0, colour=False, launches_per_step=64, streams=2,
               metric="STFT frames/sec (2048-pt, 75% overlap, 8 channels mixed to one column)", unit="frames/s",
               workload="configs[2]: 8-channel 48 kHz, 2048-pt FFT, 512 hop (75 % overlap), Hann, AbsMean mix, 4096 columns/launch"),
    "c5": dict(n=4096, hop=512, channels=2, frames=1875, fs=96000.0, colour=True, launches_per_step=128, streams=3, batch_launches_per_step=3,
               metric="STFT->ARGB columns/sec (4096-pt, 87.5% overlap, stereo 96 kHz)", unit="columns/s",
               workload="configs[4]: stereo 96 kHz, 4096-pt FFT, 512 hop (87.5 % overlap), AbsMean, Jade LUT -50..50 dB -> ARGB image, "
                        "independent images of 1875 columns (10 s), fused STFT -> palette index -> ARGB"),
}


def algorithmic_bytes_per_launch(c) -> int:
    H = c["n"] // 2 + 1
    per_column = 4 * c["hop"] * c["channels"] + (4 * H if c["colour"] else 4 * H)   # input once + one dB column, or one ARGB column
    return per_column * c["frames"]


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


def kernel_source_sha() -> str:
    """Hash of the sources the kernels and their launcher are built from (the three .hip units, the kernel header and the build
    flags in _build.py): a profile under profiles/ describes this build exactly when it carries the same hash, whatever else was
    committed in between (host-side engine code, headers' comments, tests, documents)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "jadespectrogram_amd")
    files = [os.path.join(base, "csrc", f) for f in ("jsg_stft_kernel.h", "jsg_stft_a.hip", "jsg_stft_b.hip", "jsg_kernels.hip")] + [os.path.join(base, "_build.py")]
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _git_commit() -> str:
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:   # the GPU box has no .git: the build step left a note next to the library it built
        return json.load(open(os.path.join(ROOT, "jadespectrogram_amd", "_build_info.json"))).get("commit", "unknown") + " (build)"
    except Exception:
        return "unknown"


def cpu_baseline(c, seconds_budget: float = 10.0):
    """The oracle's CPU path (oracle/jsg_oracle_c.c, a scalar float32 C port of Spectrogram::processSynchronBlock, plus
    CColorPalette::getRGBColor for c5) timed on this host -- rank 0, N = 1 only.  A reported baseline, not the target.
    This leg and parity_report() are the only places where bench.py touches oracle/ (test infrastructure)."""
    import numpy as np
    from oracle import jsg_oracle as oracle
    from oracle import oracle_c
    port = oracle_c.load()
    n, hop, C = c["n"], c["hop"], c["channels"]
    win = oracle.window(oracle.WIN_HANN, n)
    fb = n // hop
    units_per_col = 1 if c["colour"] else C          # frames (FFTs) for c2/c3, columns for c5
    cols = max(256, min(20000, int(1.5e5 * 1024 / n / C)))
    x = synth_audio(C, cols * hop + n, fs=c["fs"])
    out = np.empty((cols, n // 2 + 1), np.float32)
    pal = lut = None
    if c["colour"]:
        pal = oracle.OracleColorPalette(256, oracle.CM_JADE)
        pal.set_value_range(-50.0, 50.0)
        lut = oracle.compute_colors(256, oracle.CM_JADE)

    def once(threads):
        port.stft_db(x, n, hop, cols, win, feedblocks=fb, mix=0, threads=threads, out=out)
        if c["colour"]:
            port.colour_columns(out, lut, float(pal.vmin), float(pal.vmax), float(pal.mult), threads=threads)

    def timed(threads, budget):
        once(threads)                                 # page in, build the plan
        rates = []
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end or len(rates) < 3:
            t0 = time.perf_counter()
            once(threads)
            rates.append(cols * units_per_col / (time.perf_counter() - t0))
        return float(np.median(rates)), len(rates)

    v1, k1 = timed(1, seconds_budget)
    res = {"value": v1, "unit": c["unit"], "cores": 1, "kind": "port",
           "sample": f"median of {k1} passes over {cols} columns ({cols * C} FFTs) of the bench workload through oracle/jsg_oracle_c.c "
                     f"(scalar float32 C port of Spectrogram::processSynchronBlock{' + getRGBColor' if c['colour'] else ''}), 1 thread, "
                     f"about {seconds_budget:.0f} s",
           "cpu_model": _cpu_model()}
    cores = min(len(os.sched_getaffinity(0)), 64)
    if cores > 1:
        v2, k2 = timed(cores, seconds_budget / 3)
        res["all_cores"] = {"value": v2, "unit": c["unit"], "cores": cores,
                            "sample": f"median of {k2} passes, OpenMP over columns, plan and buffers reused across passes"}
    return res


# Vector instructions per FFT of the kernels the configurations time (SQ_INSTS_VALU / FFTs, profiles/r03_*_hbm_traffic.json ->
# derived) and the shader clock the chip holds under that load (C3: SQ_WAVE_CYCLES x 4 / waves = 67.7 k cycles per wave over a
# 40.4 us dispatch = 1.68 GHz -- well under the 2.4 GHz maximum: packed-math kernels are power-limited): the issue roof of the
# compute-bound configurations (a wave64 VALU instruction holds its SIMD for 4 cycles; 256 CUs x 4 SIMDs).
VALU_PER_FFT = {"c2": 221.0, "c3": 372.0, "c5": 1112.0}
CLOCK_GHZ_UNDER_LOAD = 1.7


def valu_roof(c, units_per_launch, inorder_us):
    key = {1024: "c2", 2048: "c3", 4096: "c5"}[c["n"]]
    ffts = units_per_launch * (c["channels"] if c["colour"] else 1)
    roof = 1024 * CLOCK_GHZ_UNDER_LOAD * 1e9 / (4.0 * VALU_PER_FFT[key])          # FFT/s with every SIMD issuing every cycle
    got = ffts / (inorder_us * 1e-6)
    return {"bound": "valu_issue", "achieved": got, "peak": roof, "unit": "FFT/s", "frac": got / roof,
            "note": f"{VALU_PER_FFT[key]:.0f} vector instructions per FFT (PMC) x 4 cycles on 1024 SIMDs at ~{CLOCK_GHZ_UNDER_LOAD} GHz under load; "
                    "C3 and C5 are bound here, not by HBM (DESIGN.md section 6)"}


def boundary_latency_subprocess():
    """What the drop-in sees, part 1 (rank 0, N = 1; started BEFORE this process touches the GPU): the audio-thread call
    jsg_process_block -- reference call path PluginProcessor.cpp:145-150 -> Spectrogram::processSynchronBlock -- timed in its own
    small C++ process (tests/cpp/producer_latency_test.cpp, C-ABI only) on the C5 geometry (stereo 96 kHz, 4096 points, ring
    1875 x 2049) while a consumer thread reads the ring and recolours the 15 MB image without pause (Spectrogram.cpp:590-608)."""
    import tempfile
    try:
        libdir = os.path.join(ROOT, "jadespectrogram_amd")
        exe = os.path.join(tempfile.gettempdir(), "jsg_bench_producer_latency")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "producer_latency_test.cpp"), "-o", exe, "-L", libdir, "-ljsg",
                               f"-Wl,-rpath,{libdir}", "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = subprocess.run([exe, "400", "300"], capture_output=True, text=True, timeout=120)
        info = json.loads(r.stdout.strip().splitlines()[-1])
        return {"call": "jsg_process_block (host planar pointers in, enqueue only), one 4096-sample stereo block per call, 400 calls, 300 us apart",
                "under": f"a consumer thread alternating jsg_get_mem and jsg_display_update on the 1875 x 2049 ring without pause ({info['reads']} reads meanwhile)",
                "p50_us": info["p50_us"], "p99_us": info["p99_us"], "max_us_after_first_call": info["max_after_first_us"],
                "first_call_us": info["first_call_us"],
                "ring_bit_identical_to_undisturbed_batch_run": info["differing_floats"] == 0 and info["differing_pixels"] == 0}
    except Exception as e:   # a reported figure, never a reason to lose the bench line
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def boundary_pcie_rate(jsg, c):
    """What the drop-in sees, part 2: frames/s through the engine fed from HOST memory (jsg_process_blocks: pinned-staging-free
    H2D copy of a whole batch + one launch), 4096-frame batches of the configuration -- PCIe-inclusive, never the `value`."""
    import numpy as np
    import torch
    n, hop, C = c["n"], c["hop"], c["channels"]
    blocks = c["frames"] * hop // n                       # fft-size blocks that make c["frames"] columns
    s = jsg.Spectrogram(C)
    s.setSamplerate(c["fs"]); s.setmemoryTime_s(60.0); s.setFFTSize(n)
    s._c(jsg.capi.lib().jsg_set_feed_percent_ext(s._h, 100.0 * hop / n))
    x = synth_audio(C, blocks * n, fs=c["fs"], seed=99)
    xp = torch.from_numpy(x).pin_memory().numpy()
    out = {}
    for name, buf in (("pageable", x), ("pinned", xp)):
        for _ in range(3):
            s.processBlocks(buf)
        s.sync()
        K = 40
        t0 = time.perf_counter()
        for _ in range(K):
            s.processBlocks(buf)
        s.sync()
        dt = time.perf_counter() - t0
        out[name] = {"frames_per_s": K * blocks * (n // hop) * C / dt, "us_per_batch": dt / K * 1e6, "h2d_GBps": K * x.nbytes / dt / 1e9}
    s.close()
    return {"call": f"jsg_process_blocks, host buffers in, batches of {blocks * (n // hop)} columns x {C} channel(s), dB ring on the device",
            "host_memory": out, "note": "PCIe-inclusive: the H2D copy of the samples is inside the timed loop; never the `value`"}


def parity_report(jsg, c, plan, d_in_host, win):
    """What the tolerances of the parity tests mean on THIS workload (rank 0, N = 1), measured on the launch geometry of the timed
    region -- the same frame count, channel count and automatic kernel selection, so the numbers describe the kernel that is
    timed (named in "kernel") -- against the float64 DFT on a spread of its columns: the share of bins whose relative power
    error exceeds plain 1e-5 and how far below their frame's peak they sit; and the number of colour indices that differ end to
    end (GPU power -> GPU dB -> GPU index against oracle power -> oracle dB -> oracle index) with the default Jade/256/(-50,50)
    palette.  Checker code: oracle/ (test infrastructure)."""
    import numpy as np
    import torch
    from oracle import jsg_oracle as oracle
    n, hop, C = c["n"], c["hop"], c["channels"]
    H = n // 2 + 1
    F = c["frames"]                                   # the timed launch geometry
    x = d_in_host[:, :(F - 1) * hop + n]
    d_x = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    pitch = (H + 31) // 32 * 32
    d_pow = torch.empty((F, pitch), dtype=torch.float32, device="cuda")
    mix = jsg.capi.MIX_ABSMEAN
    kernel = jsg.stft_kernel_name(plan, d_x, hop, F, d_pow, feedblocks=n // hop, mix_mode=mix)
    jsg.stft_db(plan, d_x, hop, F, d_pow, feedblocks=n // hop, mix_mode=mix, linear_out=True)
    d_db = torch.empty((F, pitch), dtype=torch.float32, device="cuda")
    jsg.stft_db(plan, d_x, hop, F, d_db, feedblocks=n // hop, mix_mode=mix)
    d_lut = torch.from_numpy(jsg.colormap_lut(256, jsg.capi.CM_JADE)).cuda()
    d_img = torch.zeros((H, F), dtype=torch.int32, device="cuda")
    d_scr = torch.zeros((F, (H + 63) // 64 * 64), dtype=torch.uint8, device="cuda")
    jsg.stft_image(plan, d_x, hop, F, d_lut, -50.0, 50.0, d_img, d_scr, feedblocks=n // hop, mix_mode=mix)
    d_idx = torch.zeros((H, F), dtype=torch.uint8, device="cuda")      # the image's palette indices (colour loop on the dB ring)
    jsg.colormap(d_db, d_lut, -50.0, 50.0, d_index=d_idx, n_cols=F, height=H)
    torch.cuda.synchronize()
    cols = np.unique(np.linspace(0, F - 1, min(F, 512)).astype(np.int64))   # the columns the oracle checks
    got_p = d_pow[:, :H].cpu().numpy()[cols].astype(np.float64)
    got_idx = d_idx.cpu().numpy()[::-1, :].T[cols]                      # [column][bin] (the image is flipped: y = H-1-bin)
    idx = (cols * hop)[:, None] + np.arange(n)[None, :]
    frames = (x[:, idx] * win[None, None, :]).astype(np.float32)                       # [C][F'][n]
    p64 = oracle.power_spectrum_f64(frames)                                            # float64 DFT of the float32 frames
    ref_mixed32 = oracle.mix_channels(p64.astype(np.float32), oracle.MIX_ABSMEAN)      # the reference's float32 channel mix
    ref_p = ref_mixed32.astype(np.float64)
    rel = np.abs(got_p - ref_p) / np.maximum(ref_p, 1e-300)
    bad = rel > 1e-5
    peak = ref_p.max(axis=1, keepdims=True)
    level_db = 10.0 * np.log10(np.maximum(ref_p, 1e-300) / peak)
    strong = ref_p > 1e-2 * peak
    pal = oracle.OracleColorPalette(256, oracle.CM_JADE)
    pal.set_value_range(-50.0, 50.0)
    ref_idx = pal.index(oracle.to_db(ref_mixed32)).astype(np.uint8)
    flips = int((got_idx != ref_idx).sum())
    # the fused image (jsg_stft_image_launch) must show the same pixels as dB ring + colour loop: compare as ARGB
    lut = jsg.colormap_lut(256, jsg.capi.CM_JADE).astype(np.uint32) | np.uint32(0xFF000000)
    img_cols = d_img.cpu().numpy().view(np.uint32)[::-1, :].T[cols]
    fused_differs = int((img_cols != lut[got_idx]).sum())
    db_err = np.abs(d_db[:, :H].cpu().numpy()[cols].astype(np.float64) - oracle.to_db(ref_mixed32).astype(np.float64))
    return {"kernel": kernel, "launch_checked": f"{F} columns x {C} channel(s), automatic kernel selection (the timed geometry)",
            "columns_checked_against_float64": int(len(cols)), "bins_checked": int(rel.size),
            "frac_bins_rel_power_err_gt_1e-5": float(bad.mean()),
            "those_bins_level_below_frame_peak_db": {"median": float(np.median(level_db[bad])) if bad.any() else None,
                                                     "highest": float(level_db[bad].max()) if bad.any() else None},
            "max_rel_power_err_bins_within_20dB_of_peak": float(rel[strong].max()),
            "max_err_relative_to_frame_peak": float((np.abs(got_p - ref_p) / peak).max()),
            "max_abs_db_err": float(db_err.max()),
            "colour_index_flips_end_to_end": flips, "pixels_checked": int(got_idx.size),
            "fused_image_pixels_differing_from_two_kernel_image": fused_differs,
            "note": "float64 DFT of the float32 windowed frames is the yardstick; indices: Jade, 256 colours, -50..50 dB"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--launches-per-step", type=int, default=0, help="launches in one step (default: the configuration's)")
    ap.add_argument("--nbuf", type=int, default=0, help="distinct batches rotated through (default: enough for > 256 MiB)")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams of the timed region (default: 4 for c2, 2 for c3, 3 for c5; 1 = in order)")
    ap.add_argument("--images-per-launch", type=int, default=0, help="c5: images of one jsg_stft_image_launch_strided call (default: the whole rotation, "
                                                                     "~39; 1 = one jsg_stft_image_launch per image, on --streams streams)")
    ap.add_argument("--issue-threads", type=int, default=2, help="host threads issuing the launches of a step (c2, streams > 1)")
    ap.add_argument("--blocks-per-cu", type=int, default=1, help="workgroups per CU of each launch when launches overlap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sub-run", action="store_true", help="(internal) this process is the default-environment child of another bench.py")
    ap.add_argument("--no-parity", action="store_true", help="skip the `parity` block (the profile scripts: its few launches would mix into the tracer's averages)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the `boundary` block (jsg_process_block latency, PCIe-inclusive rate)")
    ap.add_argument("--gate", action="store_true", help="with --no-graph: hold the stream with a gate kernel while the host enqueues a step's "
                                                        "launches, so that they run back to back even under a tracer (short kernels: c2)")
    ap.add_argument("--no-graph", action="store_true", help="issue the in-order launches from the host instead of replaying a hipGraph "
                                                            "(rocprofv3 does not see kernels inside graph replays, and its counter passes crash on them)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work at all: exercises the N-rank plumbing (barriers, reductions, "
                                                           "JSON) on a machine without GPUs; value is null")
    args = ap.parse_args()
    c = dict(CONFIGS[args.config])
    # c5: the images of a step are independent and of one geometry -> by default they go to the library as strided batches, one kernel
    # launch for a whole rotation of images (jsg_stft_image_launch_strided); --images-per-launch 1 (or --streams > 1) launches them one by one
    batch = bool(c["colour"]) and args.images_per_launch != 1 and args.streams <= 1
    lps = args.launches_per_step or (c["batch_launches_per_step"] if batch else c["launches_per_step"])
    n_streams = 1 if batch else max(1, args.streams or c["streams"])

    import numpy as np
    import torch

    # child processes (git, and make if the C oracle is stale) are started BEFORE anything initialises the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    commit = _git_commit()
    if not args.dry_run and not args.no_cpu_baseline and world == 1:   # the CPU baseline leg runs at N = 1 only
        from oracle import oracle_c
        oracle_c.load()
    default_env = None
    if not args.dry_run and not args.no_boundary and world == 1 and not args.sub_run and not c["colour"] and not args.streams:
        # the same timed region in a child process WITHOUT the eight hardware queues (GPU_MAX_HW_QUEUES unset, the library told to keep
        # its hands off): what a host sees that neither sets the variable nor loads libjsg.so before the HIP runtime starts
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        env["JSG_KEEP_HW_QUEUES"] = "1"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
                                "--no-boundary", "--sub-run"], env=env, capture_output=True, text=True, timeout=300)
            sl = json.loads(r.stdout.strip().splitlines()[-1])
            default_env = {"GPU_MAX_HW_QUEUES": "unset (runtime default 4)", "value": sl["value"], "unit": sl["unit"],
                           "timed_region_frac_of_8p0": sl["roofline"]["timed_region_frac_of_8p0"], "steps": 10}
        except Exception as e:
            default_env = {"error": f"{type(e).__name__}: {e}"[:200]}
    boundary = None
    if not args.dry_run and not args.no_boundary and world == 1:
        boundary = {"process_block_latency": boundary_latency_subprocess()}

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # JSG_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then share devices);
    # the driver's runs use the default: one rank per GPU, RCCL ("nccl") over xGMI
    backend = "gloo" if args.dry_run else os.environ.get("JSG_BENCH_BACKEND", "nccl")
    dist = None
    dev_index = 0
    if not args.dry_run:
        dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        if not args.dry_run:
            torch.cuda.synchronize()

    n, hop, C, F = c["n"], c["hop"], c["channels"], c["frames"]
    H = n // 2 + 1
    algo = algorithmic_bytes_per_launch(c)
    units_per_launch = F if c["colour"] else F * C     # columns for c5, frames (FFTs) otherwise

    run_step = None
    ipl, nrot = 1, 0                                  # images per launch (c5 batches), distinct launch argument sets
    inorder_us = eager_us = None
    copy_us = None
    parity = None
    nbuf = 0
    if not args.dry_run:
        import jadespectrogram_amd as jsg
        from jadespectrogram_amd import capi
        from jadespectrogram_amd.spectrogram import _stft_args
        lib = capi.lib()
        win = jsg.window(jsg.capi.WIN_HANN, n)
        plan = jsg.Plan(n, win)
        n_samples = F * hop + (n - hop)
        pitch = (H + 31) // 32 * 32
        idx_pitch = (H + 63) // 64 * 64
        img_pitch = (F + 31) // 32 * 32
        per_batch = C * n_samples * 4 + ((0 if batch else F * idx_pitch) + H * img_pitch * 4 if c["colour"] else F * pitch * 4)
        # Rotation: distinct batches worth ~1 GB, about four times the 256 MiB Infinity Cache.  (Rounds 1-2 rotated over 0.3 GB; the
        # --nbuf sweep of round 3 -- profiles/r03_c2_nbuf_sweep.json -- showed that a good part of those reads still hit the cache:
        # C2 1.35e9 frames/s at 0.35 GB, 1.12e9 at 0.7 GB, 1.10e9 at 1.4 GB.)
        nbuf = args.nbuf or max(2, int(ROTATION_BYTES // per_batch) + 1)
        while n_streams > 1 and nbuf % n_streams:   # a batch must always land on the same stream (its ring is rewritten in order)
            nbuf += 1
        if batch:                                   # the rotation is cut into launches of `ipl` images each
            ipl = min(args.images_per_launch or nbuf, nbuf)
            nbuf = (nbuf + ipl - 1) // ipl * ipl
            nrot = nbuf // ipl
            algo *= ipl
            units_per_launch *= ipl
        base = synth_audio(C, n_samples + nbuf * 64, fs=c["fs"], seed=1234 + 1000 * rank)   # SURVEY 8d signal
        d_in, d_out, d_img, d_scr = [], [], [], []
        d_in_all = torch.empty((nbuf, C, n_samples), dtype=torch.float32, device="cuda")      # contiguous: image b = d_in_all[b] (strided launches)
        d_img_all = torch.zeros((nbuf, H, img_pitch), dtype=torch.int32, device="cuda") if c["colour"] else None
        for b in range(nbuf):
            d_in_all[b].copy_(torch.from_numpy(np.ascontiguousarray(base[:, b * 64:b * 64 + n_samples])))
            d_in.append(d_in_all[b])
            if c["colour"]:
                d_img.append(d_img_all[b])
                d_scr.append(torch.zeros((F, idx_pitch), dtype=torch.uint8, device="cuda") if not batch else None)
            else:
                d_out.append(torch.empty((F, pitch), dtype=torch.float32, device="cuda"))
        if not batch:
            nrot = nbuf
        d_lut = torch.from_numpy(jsg.colormap_lut(256, jsg.capi.CM_JADE)).cuda() if c["colour"] else None
        fb = n // hop
        kernel_label = None
        if batch:
            two = jsg.stft_image_strided_needs_scratch(plan, d_in_all[:ipl], hop, F, d_lut, -50.0, 50.0, d_img_all[:ipl, :, :F], None, feedblocks=fb,
                                                       mix_mode=jsg.capi.MIX_ABSMEAN)
            assert not two, "the strided C5 launch is expected to take the single-kernel form"
            kernel_label = (f"stft_db_kernel<Cfg4096B, AbsMean, ARGB out> (jsg_stft_image_launch_strided: {ipl} images in one kernel launch, "
                            "the workgroups colour their columns)")
        elif c["colour"]:
            two = jsg.stft_image_needs_scratch(plan, d_in[0], hop, F, d_lut, -50.0, 50.0, d_img[0][:, :F], d_scr[0], feedblocks=fb,
                                               mix_mode=jsg.capi.MIX_ABSMEAN)
            kernel_label = ("stft_db_kernel<4096, AbsMean, index out> + colormap_kernel (jsg_stft_image_launch, two kernels)" if two else
                            "stft_db_kernel<Cfg4096B, AbsMean, ARGB out> (jsg_stft_image_launch, one kernel: the workgroup colours its columns)")
        one = torch.cuda.Stream()

        def launch(b, stream_handle, bpc=0):
            if batch:       # launch b of the rotation: images [b * ipl, (b + 1) * ipl)
                jsg.stft_image_strided(plan, d_in_all[b * ipl:(b + 1) * ipl], hop, F, d_lut, -50.0, 50.0, d_img_all[b * ipl:(b + 1) * ipl, :, :F], None,
                                       feedblocks=fb, mix_mode=jsg.capi.MIX_ABSMEAN, stream=stream_handle)
            elif c["colour"]:
                jsg.stft_image(plan, d_in[b], hop, F, d_lut, -50.0, 50.0, d_img[b][:, :F], d_scr[b], feedblocks=fb,
                               mix_mode=jsg.capi.MIX_ABSMEAN, stream=stream_handle)
            else:
                jsg.stft_db(plan, d_in[b], hop, F, d_out[b], feedblocks=fb, mix_mode=jsg.capi.MIX_ABSMEAN, blocks_per_cu=bpc,
                            stream=stream_handle)

        # ---- the in-order group of one step as a hipGraph on ONE stream (per-kernel view; the timed region of c3 / c5) ----
        with torch.cuda.stream(one):
            for b in range(min(nrot, lps)):
                launch(b, one.cuda_stream)
            torch.cuda.synchronize()
        gate = {"cycles_per_ms": 0.0, "cycles": 0}
        if args.gate:
            # --gate: a gate kernel (torch.cuda._sleep: spins for a fixed number of cycles, i.e. it always ends) holds a stream
            # while the host enqueues a whole step behind it, so the step's launches run BACK TO BACK on the GPU even when the host
            # is slow (under rocprofv3 it needs ~11 us per launch).  Calibrated here: spin cycles per millisecond.
            with torch.cuda.stream(one):
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(one); torch.cuda._sleep(20_000_000); g1.record(one)
                torch.cuda.synchronize()
            gate["cycles_per_ms"] = 20_000_000 / max(g0.elapsed_time(g1), 1e-3)
        if args.no_graph:
            # Host-issued launches (rocprofv3 does not see kernels inside graph replays).  With --gate they still run back to
            # back; without it every 5 us dispatch of c2 starts on an idle GPU and reads ~1 us longer (6.3 vs 5.7 us).  For the
            # long kernels of c3 / c5 the tracer's own per-dispatch work shows up instead when they run back to back, so they are
            # traced ungated.

            def gated(issue):
                def run():
                    with torch.cuda.stream(one):
                        if gate["cycles"] and args.gate:
                            torch.cuda._sleep(gate["cycles"])
                        t_i = time.perf_counter()
                        issue()
                        t_i = (time.perf_counter() - t_i) * 1e3
                    # next time the gate outlasts the host's issue time by half (bounded: at most 0.5 s)
                    gate["cycles"] = int(min(500.0, 1.5 * t_i + 0.2) * gate["cycles_per_ms"])
                return run
            if c["colour"]:
                def _issue():
                    for i in range(lps):
                        launch(i % nrot, one.cuda_stream)
                replay_inorder = gated(_issue)
            else:
                arr1 = (capi.StftArgs * lps)()
                for i in range(lps):
                    a_i = _stft_args(plan, d_in[i % nbuf], hop, F, d_out[i % nbuf], feedblocks=fb, mix_mode=jsg.capi.MIX_ABSMEAN)
                    ctypes.memmove(ctypes.byref(arr1, i * ctypes.sizeof(capi.StftArgs)), ctypes.byref(a_i), ctypes.sizeof(capi.StftArgs))
                one_arr = (ctypes.c_void_p * 1)(one.cuda_stream)

                replay_inorder = gated(lambda: capi.check(lib.jsg_stft_db_launch_many(plan._p, arr1, lps, one_arr, 1)))
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(one):
                with torch.cuda.graph(graph, stream=one):
                    for i in range(lps):
                        launch(i % nrot, one.cuda_stream)
            torch.cuda.synchronize()

            def replay_inorder():
                with torch.cuda.stream(one):
                    graph.replay()

        if n_streams > 1 and c["colour"] and args.no_graph:
            # ---- the same independent images, host-issued round-robin over the streams (for the tracer: it does not see kernels
            #      inside graph replays); --gate holds every stream while the host enqueues the step ----
            streams = [torch.cuda.Stream() for _ in range(n_streams)]

            def run_step():
                if args.gate and gate["cycles"]:
                    for st in streams:
                        with torch.cuda.stream(st):
                            torch.cuda._sleep(gate["cycles"])
                t_i = time.perf_counter()
                for i in range(lps):
                    launch(i % nbuf, streams[i % n_streams].cuda_stream)
                if args.gate:
                    gate["cycles"] = int(min(500.0, 1.5 * (time.perf_counter() - t_i) * 1e3 + 0.2) * gate["cycles_per_ms"])
        elif n_streams > 1 and c["colour"]:
            # ---- independent images on n_streams streams: one hipGraph per stream (image i goes to stream i % n_streams, so a
            #      batch's scratch and image are always rewritten in order), replayed together: the colour kernel of one image
            #      runs beside the STFT kernel of the next ----
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            graphs = []
            for si, st in enumerate(streams):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(st):
                    with torch.cuda.graph(gph, stream=st):
                        for i in range(si, lps, n_streams):
                            launch(i % nbuf, st.cuda_stream)
                graphs.append(gph)
            torch.cuda.synchronize()

            def run_step():
                for st, gph in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        gph.replay()
        elif n_streams > 1:
            # ---- the overlapped group: lps independent launches from ONE C call over n_streams streams ----
            arr = (capi.StftArgs * lps)()
            for i in range(lps):
                a_i = _stft_args(plan, d_in[i % nbuf], hop, F, d_out[i % nbuf], feedblocks=fb, mix_mode=jsg.capi.MIX_ABSMEAN,
                                 blocks_per_cu=args.blocks_per_cu)
                ctypes.memmove(ctypes.byref(arr, i * ctypes.sizeof(capi.StftArgs)), ctypes.byref(a_i), ctypes.sizeof(capi.StftArgs))
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            sarr = (ctypes.c_void_p * n_streams)(*[st.cuda_stream for st in streams])
            use_pool = not args.streams and not args.gate    # default: the LIBRARY's launch pool (its own streams and issuing threads)

            def run_step():
                if use_pool:
                    capi.check(lib.jsg_stft_db_launch_batches(plan._p, arr, lps, ctypes.c_void_p(one.cuda_stream)))
                    return
                # --gate (for the tracer, tools/profile_overlap.sh): every stream is held by a bounded spin kernel while the host
                # enqueues the step, so the launches overlap on the GPU exactly as they do when the host keeps up
                if args.gate and gate["cycles"]:
                    for st in streams:
                        with torch.cuda.stream(st):
                            torch.cuda._sleep(gate["cycles"])
                t_i = time.perf_counter()
                capi.check(lib.jsg_stft_db_launch_many_threads(plan._p, arr, lps, sarr, n_streams, max(1, args.issue_threads)))
                if args.gate:   # next time the gates outlast the host's issue time by half (bounded: at most 0.5 s)
                    gate["cycles"] = int(min(500.0, 1.5 * (time.perf_counter() - t_i) * 1e3 + 0.2) * gate["cycles_per_ms"])
        else:
            run_step = replay_inorder
    else:
        def run_step():
            time.sleep(0.001)

    # ---- settle the clocks (about 0.3 s of the same work), W warm-up steps, then EXACTLY K timed steps ----
    prewarm_s = 0.0
    if not args.dry_run:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            run_step()
            sync()
        prewarm_s = time.perf_counter() - t_pre
    for _ in range(args.warmup):
        run_step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync()
    t1 = time.perf_counter()
    barrier(); sync()
    wall = t1 - t0

    if not args.dry_run:
        # ---- per-kernel view: the same launches in order on one stream, HIP events on that stream ----
        reps = max(3, min(args.steps, 20))
        replay_inorder(); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.no_graph:   # the gate is part of replay_inorder(): the events go between the gate and the launches
            tot = 0.0
            for _ in range(reps):
                with torch.cuda.stream(one):
                    if gate["cycles"] and args.gate:
                        torch.cuda._sleep(gate["cycles"])
                    ev0.record(one)
                    if c["colour"]:
                        for i in range(lps):
                            launch(i % nrot, one.cuda_stream)
                    else:
                        capi.check(lib.jsg_stft_db_launch_many(plan._p, arr1, lps, one_arr, 1))
                    ev1.record(one)
                torch.cuda.synchronize()
                tot += ev0.elapsed_time(ev1)
            inorder_us = tot * 1e3 / (reps * lps)
        else:
            with torch.cuda.stream(one):
                ev0.record(one)
                for _ in range(reps):
                    replay_inorder()
                ev1.record(one)
            torch.cuda.synchronize()
            inorder_us = ev0.elapsed_time(ev1) * 1e3 / (reps * lps)
        if not c["colour"] and not args.no_graph:   # host-issued launches on the same stream, for comparison (adds the runtime's per-launch handling)
            k = min(lps, 512)
            with torch.cuda.stream(one):
                for i in range(64):
                    launch(i % nrot, one.cuda_stream)
                torch.cuda.synchronize()
                ev0.record(one)
                for i in range(k):
                    launch(i % nrot, one.cuda_stream)
                ev1.record(one)
            torch.cuda.synchronize()
            eager_us = ev0.elapsed_time(ev1) * 1e3 / k
    barrier(); sync()
    if dist is not None:
        t = torch.tensor([wall, inorder_us or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, inorder_us = float(t[0]), (float(t[1]) or None)

    if rank == 0 and not args.dry_run and not args.no_graph:
        # context for the roofline: a plain device-to-device copy of the SAME byte count, same rotation, same graph timing --
        # what a launch of this size can reach at all on this GPU
        nflt = algo // 8
        csrc = [torch.rand(nflt, device="cuda") for _ in range(nrot)]
        cdst = [torch.empty(nflt, device="cuda") for _ in range(nrot)]
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(one):
            cdst[0].copy_(csrc[0]); torch.cuda.synchronize()
            with torch.cuda.graph(g2, stream=one):
                for i in range(min(lps, 256)):
                    cdst[i % nrot].copy_(csrc[i % nrot])
            g2.replay(); torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(one)
            for _ in range(4):
                g2.replay()
            c1.record(one)
        torch.cuda.synchronize()
        copy_us = c0.elapsed_time(c1) * 1e3 / (4 * min(lps, 256))
        del csrc, cdst, g2
    if rank == 0 and not args.dry_run and world == 1 and not args.no_parity:
        parity = parity_report(jsg, c, plan, base, win)
        if batch:   # the timed strided launch against one-by-one launches of the same images (same kernel plan), three images of the batch
            with torch.cuda.stream(one):
                launch(0, one.cuda_stream)
                tmp = torch.zeros((H, img_pitch), dtype=torch.int32, device="cuda")
                differing = 0
                for k in sorted({0, ipl // 2, ipl - 1}):
                    tmp.zero_()
                    jsg.stft_image(plan, d_in[k], hop, F, d_lut, -50.0, 50.0, tmp[:, :F], None, feedblocks=fb, mix_mode=jsg.capi.MIX_ABSMEAN,
                                   plan_select=2, stream=one.cuda_stream)
                    differing += int((tmp != d_img_all[k]).sum())
            parity["strided_batch_pixels_differing_from_single_launches"] = differing

    units_total = world * args.steps * lps * units_per_launch
    gate_note = " (the gate is inside the timed region: this mode is for the tracer)"
    if batch:
        issue_text = ("strided batches of independent images, one kernel launch per batch (jsg_stft_image_launch_strided), "
                      + ("host-issued" if args.no_graph else "hipGraph replay") + ", in order on one stream")
    elif n_streams == 1:
        issue_text = ("hipGraph replay, in order" if not args.no_graph else
                      "host-issued behind a gate kernel, in order and back to back on one stream" + gate_note if args.gate else "host-issued, in order on one stream")
    elif c["colour"]:
        issue_text = f"{n_streams} hipGraphs per step (independent images, one stream each), replayed together"
    else:
        issue_text = ("one jsg_stft_db_launch_batches call per step on one caller stream: the library forks onto its own streams (4; 2 for the "
                      "one-workgroup-per-CU kernels), issues from 2 host threads and joins" if (not args.streams and not args.gate) else
                      f"one C call per step, {max(1, args.issue_threads)} host thread(s)")
        if args.gate:
            issue_text += ", every stream held by a gate kernel while the host enqueues the step" + gate_note
    out = {
        "metric": c["metric"], "value": None if args.dry_run else units_total / wall, "unit": c["unit"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": c["workload"],
                   "step": (f"{lps} launches x {ipl} images x {F} columns = {lps * units_per_launch} columns per step and GPU" if batch else
                            f"{lps} launches x {F} {'columns' if c['colour'] or C > 1 else 'frames'} = {lps * units_per_launch} {c['unit'].split('/')[0]} per step and GPU"),
                   "launches_per_step": lps, "images_per_launch": ipl if c["colour"] else None, "frames_per_launch": F * C * ipl, "columns_per_launch": F * ipl,
                   "channels_per_gpu": C,
                   "distinct_batches": nbuf, "hip_streams_per_gpu": n_streams, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "issue": issue_text,
                   "prewarm_s": round(prewarm_s, 3),
                   "parallelism": f"{world} GPU(s), independent batches, no data-path collective"},
    }
    if args.dry_run:
        out["dry_run"] = True
    if inorder_us:
        achieved = algo / (inorder_us * 1e-6) / 1e9
        conc = algo * lps * args.steps / wall / 1e9
        # HBM traffic (PMC) and the tracer's per-dispatch duration come from the rocprofv3 passes of tools/profile_bench.sh
        # (profiles/r03_<cfg>_hbm_traffic.json): bench.py does not run counters itself.  The file carries the hash of the kernel
        # sources it was recorded with; when that differs from this build's the figures are flagged as stale and `frac` falls back
        # to this run's own event timing.
        traffic, tsrc, rocprof_us = None, None, None
        prof = os.path.join(ROOT, "profiles", f"r03_{args.config}_hbm_traffic.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                same = pj.get("kernel_source_sha") == kernel_source_sha() and int(pj.get("images_per_launch") or 1) == ipl
                traffic = pj.get("hbm_bytes_per_launch")
                rocprof_us = pj.get("avg_us") if same else None
                tsrc = {"file": os.path.relpath(prof, ROOT), "recorded_at_commit": pj.get("commit"), "kernel_source_sha": pj.get("kernel_source_sha"),
                        "matches_this_build": same, "rocprof_avg_dispatch_us": pj.get("avg_us"),
                        "note": "rocprofv3 passes of tools/profile_bench.sh (kernel trace; FETCH_SIZE x2 on gfx950 + WRITE_SIZE in separate PMC passes), "
                                "recorded earlier; NOT measured by this run" + ("" if same else " -- STALE: the kernel sources have changed since")}
            except Exception:
                traffic = None
        frac_events = achieved / HBM_PEAK_GBS
        frac_rocprof = (algo / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None
        out["roofline"] = {
            "bound": "hbm", "achieved": (frac_rocprof or frac_events) * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": frac_rocprof or frac_events,
            "frac_source": ("rocprofv3 per-dispatch average of the same in-order launches (profiles/, same kernel sources)" if frac_rocprof
                            else "this run's HIP events (no matching rocprofv3 profile under profiles/)"),
            "frac_event_timed": frac_events, "frac_rocprof": frac_rocprof,
            "frac_of_8p0": frac_rocprof or frac_events, "frac_of_6p3": (frac_rocprof or frac_events) * HBM_PEAK_GBS / HBM_ACHIEVABLE_GBS,
            "traffic": traffic, "traffic_source": tsrc,
            "kernel": (kernel_label if c["colour"] else f"stft_db_kernel<{(parity or {}).get('kernel', n)}>"),
            "avg_launch_us": inorder_us, "avg_launch_us_host_issued": eager_us,
            "how": f"HIP events on the launch stream around {'host-issued runs' if args.no_graph else 'hipGraph replays'} of the step's {lps} launches, one at a time in order",
            "algorithmic_bytes_per_launch": algo,
            "timed_region_achieved": conc, "timed_region_frac_of_8p0": conc / HBM_PEAK_GBS, "timed_region_frac_of_6p3": conc / HBM_ACHIEVABLE_GBS,
            "note": "achieved/frac: per-kernel view (in order, one stream); timed_region_*: algorithmic bytes of the K timed steps / their wall "
                    "time (independent launches overlapped on hip_streams_per_gpu streams)",
            "memcpy_same_bytes_us": copy_us, "frac_of_memcpy_rate": (copy_us / inorder_us) if copy_us else None,
            "second_roof": valu_roof(c, units_per_launch, inorder_us),
            "commit": commit,
        }
    if parity is not None:
        out["parity"] = parity
    if default_env is not None:
        out["config"]["same_region_default_environment"] = default_env
    if boundary is not None and rank == 0:
        boundary["pcie_inclusive_rate"] = boundary_pcie_rate(jsg, c)
        out["boundary"] = boundary
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(c)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
