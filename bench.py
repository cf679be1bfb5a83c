#!/usr/bin/env python3
"""Headline benchmark: STFT frames/sec (1024-pt, 50 % hop) on N MI355X, with the kernel's HBM roofline fraction.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it through torch.distributed.run with one
rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE in the environment).  Started WITHOUT a launcher (`python bench.py --gpus 8`), the parent
process -- before it imports torch or touches a GPU -- starts the N ranks itself (spawn_ranks) and exits with their code.  Rank 0 prints ONE
JSON line.

What a step is
--------------
A step is a FIXED, stated number of kernel dispatches over synthetic batches that are resident in HBM, all on ONE stream, in
order, issued by plain C calls -- no extra streams, no hardware-queue setting, no issuing threads, no hipGraph:

  c2 (default at N = 1, BASELINE.json configs[1]: mono 48 kHz, 1024-point FFT, hop 512, Hann, batches of 4096 frames): one step =
      16 dispatches of jsg_stft_db_launch_strided, each covering the whole rotation of 64 independent 4096-frame batches
      (64 x 4096 frames per dispatch) = 4 194 304 frames per step.  The workgroups of a dispatch walk through the batches.
      The columns are written in the REFERENCE's layout m_mem[col][bin] (Spectrogram.h:144: n/2+1 contiguous floats per column); the
      tail-plane layout of jsg_stft_args.out_tail is timed beside it (roofline.frac_tail_plane), as is the literal one batch per
      dispatch (roofline.frac_one_batch_per_dispatch).
  c4 (default at N > 1, configs[3]: 64 channels sharded 8 per GPU): every rank holds 8 channels and writes one dB column PER CHANNEL
      (JSG_MIX_PER_CHANNEL; the reference's channel loop Spectrogram.cpp:52-59 without the mix of :64-76); one step = 16 dispatches x
      8 batches x 8 channels x 4096 frames; 4100 algorithmic bytes per channel-frame.  The N = 1 line carries one GPU's shard as extra.c4.
  c3 (configs[2]: 8 channels, 2048 points, 75 % overlap, AbsMean): one step = 4 dispatches x 12 batches x 4096 columns.
  c5 (configs[4]: stereo 96 kHz, 4096 points, 87.5 % overlap -> ARGB): one step = 3 dispatches of jsg_stft_image_launch_strided,
      each a whole rotation of ~44 independent 1875-column images.
  `--mode single` times one dispatch per batch instead (the literal "4096 frames/launch" of configs[1]; also what the
  `one_batch_per_launch` block of every default line reports, from a short in-order run).

The batches rotate over ~1 GB of distinct buffers (about four times the 256 MiB Infinity Cache; `--nbuf` overrides), so every
dispatch streams from and to HBM.  Before the W warm-up steps the same work runs for about 0.5 s so that the clocks have settled.

`value` = units of all ranks / wall time of the K timed steps (barrier + synchronize on both sides, max over ranks).
`roofline`: the timed region IS a sequence of back-to-back dispatches of one kernel on one stream, so its per-dispatch duration is
measured live with HIP events on that stream around the K steps: achieved = algorithmic bytes per dispatch / that duration, and
`timed_region_frac` (bytes of the K steps / wall) is the same number up to the host's last synchronize.  rocprofv3's per-dispatch
average of the same command is committed under profiles/ (tools/profile_bench.sh) and quoted beside it when it was recorded with
the same kernel sources.  `calibration` = what a tuned float4 copy (jsg_calib_copy_launch) reaches on this box: the measured roof.
`power` = the same steps kept running for --power-seconds (after the timed region, N = 1) with the card's socket power and core clock
sampled from its hwmon files: every kernel of this path runs the socket into its 1400 W cap, the core clock gives.

Every rank works on its own batches (weak scaling; the path shards by channel / stream, there is no data-path collective); RCCL
carries only the barriers and the max-over-ranks time.  At N = 1 the default line also carries the c4, c3 and c5 results (`extra`, and as
scalars roofline.extra_<cfg>_frac / _value), from child processes that run BEFORE this process touches the GPU.
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

PROFILE_ROUND = "r06"        # profiles/<round>_<cfg>_hbm_traffic.json: the rocprofv3 passes this build's figures are compared with
ROTATION_BYTES = 1.0e9       # distinct input + output bytes the timed launches rotate over (>> 256 MiB Infinity Cache)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_GUIDE_COPY_GBS = 6290.0  # ... and 6.29 TB/s for a float4 copy in the guide; bench.py measures its own (calibrate_copy)

# name -> workload (algorithmic bytes count every input sample once and every output value once)
CONFIGS = {
    "c2": dict(n=1024, hop=512, channels=1, frames=4096, fs=48000.0, colour=False, dispatches_per_step=16, batches_per_dispatch=64,
               metric="STFT frames/sec (1024-pt, 50% hop)", unit="frames/s",
               workload="configs[1]: mono 48 kHz, 1024-pt FFT, 512 hop, Hann, batched 4096 frames/launch -- K such batches per kernel dispatch "
                        "(see step), inputs + dB rings resident in HBM"),
    "c3": dict(n=2048, hop=512, channels=8, frames=4096, fs=48000.0, colour=False, dispatches_per_step=4, batches_per_dispatch=12,
               metric="STFT frames/sec (2048-pt, 75% overlap, 8 channels mixed to one column)", unit="frames/s",
               workload="configs[2]: 8-channel 48 kHz, 2048-pt FFT, 512 hop (75 % overlap), Hann, AbsMean mix, 4096 columns/launch"),
    # configs[3]: the shard ONE GPU holds of the 64-channel stream (8 channels per GPU at N = 8), one column PER CHANNEL (the reference's channel loop,
    # Spectrogram.cpp:52-59, without the mix of :64-76 -- channels are independent up to there): rows = batch x channel of a strided dispatch
    "c4": dict(n=1024, hop=512, channels=8, frames=4096, fs=48000.0, colour=False, per_channel=True, dispatches_per_step=16, batches_per_dispatch=8,
               metric="STFT frames/sec (1024-pt, 50% hop)", unit="frames/s",
               workload="configs[3]: 64-channel synthetic 48 kHz, 1024-pt FFT, 512 hop, Hann, sharded 8 channels per GPU (this GPU's shard: 8 channels, "
                        "one dB column per channel and frame, JSG_MIX_PER_CHANNEL), batches of 4096 frames per channel, no data-path collective"),
    "c5": dict(n=4096, hop=512, channels=2, frames=1875, fs=96000.0, colour=True, dispatches_per_step=3, batches_per_dispatch=44,
               metric="STFT->ARGB columns/sec (4096-pt, 87.5% overlap, stereo 96 kHz)", unit="columns/s",
               workload="configs[4]: stereo 96 kHz, 4096-pt FFT, 512 hop (87.5 % overlap), AbsMean, Jade LUT -50..50 dB -> ARGB image, "
                        "independent images of 1875 columns (10 s), fused STFT -> palette index -> ARGB"),
}


def algorithmic_bytes_per_batch(c) -> int:
    H = c["n"] // 2 + 1
    if c.get("per_channel"):                                                       # every channel-frame: its hop of input + its own dB column (4100 B at 1024 / 512)
        return (4 * c["hop"] + 4 * H) * c["channels"] * c["frames"]
    per_column = 4 * c["hop"] * c["channels"] + (4 * H if c["colour"] else 4 * H)   # input once + one dB column, or one ARGB column
    return per_column * c["frames"]


def synth_audio(channels: int, n_samples: int, fs: float = 48000.0, seed: int = 1234, first_channel: int = 0):
    """SURVEY 8d synthetic input: x_c[n] = 0.5 sin(2 pi f_c n / fs) + 0.1 u[n], f_c = 220*2^(c/12), u ~ U(-1,1); c counts from
    `first_channel` (a rank's shard of a wider stream)."""
    import numpy as np
    out = np.zeros((channels, n_samples), dtype=np.float32)
    t = np.arange(n_samples, dtype=np.float64)
    for c in range(channels):
        g = first_channel + c
        u = np.random.default_rng(seed + g).uniform(-1.0, 1.0, n_samples)
        out[c] = (0.5 * np.sin(2.0 * np.pi * 220.0 * 2.0 ** (g / 12.0) * t / fs) + 0.1 * u).astype(np.float32)
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
        if c.get("per_channel"):                      # one column per channel: the channels are independent mono streams
            for ch in range(C):
                port.stft_db(x[ch:ch + 1], n, hop, cols, win, feedblocks=fb, mix=0, threads=threads, out=out)
        else:
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


# Vector instructions per FFT of the kernels the configurations time (SQ_INSTS_VALU / FFTs, profiles/r05_*_hbm_traffic.json ->
# derived) and the shader clock the chip holds under that load (C3: SQ_WAVE_CYCLES x 4 / waves = 67.7 k cycles per wave over a
# 40.4 us dispatch = 1.68 GHz -- well under the 2.4 GHz maximum: packed-math kernels are power-limited): the issue roof of the
# compute-bound configurations (a wave64 VALU instruction holds its SIMD for 4 cycles; 256 CUs x 4 SIMDs).
VALU_PER_FFT = {"c2": 188.0, "c3": 352.0, "c5": 989.0}   # round 5: split-radix 16- and 32-point stages, the display kernel's trims (366 and 1099 before)
CLOCK_GHZ_UNDER_LOAD = 1.7


def valu_roof(c, units_per_launch, inorder_us, sclk_mhz=None):
    """Vector-issue roof: every SIMD issuing one vector instruction every 4 cycles, at the core clock the card SUSTAINS under this
    workload (measured by the `power` leg when the telemetry is readable: the socket sits at its power cap and the clock gives,
    DESIGN.md section 6) -- else at the ~1.7 GHz of earlier rounds' estimate."""
    key = {1024: "c2", 2048: "c3", 4096: "c5"}[c["n"]]
    ghz = sclk_mhz / 1e3 if sclk_mhz else CLOCK_GHZ_UNDER_LOAD
    ffts = units_per_launch * (c["channels"] if c["colour"] else 1)
    roof = 1024 * ghz * 1e9 / (4.0 * VALU_PER_FFT[key])          # FFT/s with every SIMD issuing every cycle
    got = ffts / (inorder_us * 1e-6)
    return {"bound": "valu_issue", "achieved": got, "peak": roof, "unit": "FFT/s", "frac": got / roof, "clock_GHz": ghz,
            "clock_source": "hwmon, median over the sustained leg of this run" if sclk_mhz else "estimate",
            "note": f"{VALU_PER_FFT[key]:.0f} vector instructions per FFT (PMC) x 4 cycles on 1024 SIMDs at the core clock under this load; "
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
        r = subprocess.run([exe, "20000", "250", "256"], capture_output=True, text=True, timeout=180)
        info = json.loads(r.stdout.strip().splitlines()[-1])
        return {"call": "jsg_process_block (wait-free: lock-free ring of page-locked memory, the engine's worker thread makes the HIP calls), one 4096-sample "
                        "stereo block per call, 20 000 calls, 250 us apart",
                "under": f"a consumer thread alternating jsg_get_mem and jsg_display_update on the 1875 x 2049 ring without pause ({info['reads']} reads meanwhile)",
                "p50_us": info["p50_us"], "p99_us": info["p99_us"], "p9999_us": info["p9999_us"], "max_us_after_first_call": info["max_after_first_us"],
                "first_call_us": info["first_call_us"], "calls_over_50us": info["calls_over_50us"], "dropped_blocks": info["dropped_blocks"],
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
    if c.get("per_channel"):
        return parity_report_per_channel(jsg, c, plan, d_in_host, win)
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
    cpu32 = float32_cpu_fft_share_beyond_1e5(oracle, frames, ref_p, oracle.MIX_ABSMEAN)
    pal = oracle.OracleColorPalette(256, oracle.CM_JADE)
    pal.set_value_range(-50.0, 50.0)
    ref_idx = pal.index(oracle.to_db(ref_mixed32)).astype(np.uint8)
    flips = int((got_idx != ref_idx).sum())
    # the fused image (jsg_stft_image_launch) must show the same pixels as dB ring + colour loop: compare as ARGB
    lut = jsg.colormap_lut(256, jsg.capi.CM_JADE).astype(np.uint32) | np.uint32(0xFF000000)
    img_cols = d_img.cpu().numpy().view(np.uint32)[::-1, :].T[cols]
    fused_differs = int((img_cols != lut[got_idx]).sum())
    db_err = np.abs(d_db[:, :H].cpu().numpy()[cols].astype(np.float64) - oracle.to_db(ref_mixed32).astype(np.float64))
    # the bit-reproducible mode on the SAME display launch (jsg_stft_args.exact_log): the image of jsg_stft_image_launch against the CPU
    # mirror of the kernel (oracle/jsg_mirror.c: its float32 arithmetic and the shared logarithm) -> CColorPalette -- 0 flips expected
    exact_flips = None
    try:
        from oracle import mirror as mirror_mod
        mm = mirror_mod.load()
        d_img_x = torch.zeros((H, F), dtype=torch.int32, device="cuda")
        jsg.stft_image(plan, d_x, hop, F, d_lut, -50.0, 50.0, d_img_x, d_scr, feedblocks=n // hop, mix_mode=mix, exact_log=True)
        torch.cuda.synchronize()
        sub = cols[:: max(1, len(cols) // 128)]                                     # (a spread of columns: the mirror is scalar C)
        mdb = np.stack([mm.columns(kernel, x, hop, 1, win, feedblocks=n // hop, mix=0, first_frame=int(cc), exact_db=True)[0] for cc in sub])
        want = (lut[pal.index(mdb).astype(np.uint8)])
        got_x = d_img_x.cpu().numpy().view(np.uint32)[::-1, :].T[sub]
        exact_flips = {"pixels_differing": int((got_x != want).sum()), "pixels_checked": int(want.size), "columns_checked": int(len(sub))}
    except Exception as e:   # a reported figure, never a reason to lose the bench line
        exact_flips = {"error": f"{type(e).__name__}: {e}"[:200]}
    return {"kernel": kernel, "launch_checked": f"{F} columns x {C} channel(s), automatic kernel selection (the timed geometry)",
            "columns_checked_against_float64": int(len(cols)), "bins_checked": int(rel.size),
            "frac_bins_rel_power_err_gt_1e-5": float(bad.mean()),
            "float32_cpu_fft_frac_bins_rel_power_err_gt_1e-5": cpu32,
            "those_bins_level_below_frame_peak_db": {"median": float(np.median(level_db[bad])) if bad.any() else None,
                                                     "highest": float(level_db[bad].max()) if bad.any() else None},
            "max_rel_power_err_bins_within_20dB_of_peak": float(rel[strong].max()),
            "max_err_relative_to_frame_peak": float((np.abs(got_p - ref_p) / peak).max()),
            "max_abs_db_err": float(db_err.max()),
            "colour_index_flips_end_to_end": flips, "pixels_checked": int(got_idx.size),
            "fused_image_pixels_differing_from_two_kernel_image": fused_differs,
            "exact_log_display_launch_vs_cpu_mirror": exact_flips,
            "note": "float64 DFT of the float32 windowed frames is the yardstick; indices: Jade, 256 colours, -50..50 dB"}


def float32_cpu_fft_share_beyond_1e5(oracle, frames_f32, ref_p64, mix):
    """Context for north_star's "within 1e-5 relative (float32) of the reference CPU path": what ANY float32 FFT on a CPU does on the same
    windowed frames -- scipy's pocketfft in single precision (the reference's own `spectrum::power` is float in, float out; its source is not
    in the reference tree) -> |X|^2 in float32 -> the reference's float32 channel mix -- against the same float64 yardstick: the share of bins
    whose relative power error exceeds plain 1e-5.  A per-bin 1e-5 bound on bins far below the frame's peak is not a property a float32
    transform has, on either side of the boundary."""
    import numpy as np
    try:
        import scipy.fft
        X = scipy.fft.rfft(np.ascontiguousarray(frames_f32, dtype=np.float32), axis=-1)          # complex64: single-precision pocketfft
        p32 = (X.real * X.real + X.imag * X.imag).astype(np.float32)
        mixed = oracle.mix_channels(p32, mix).astype(np.float64) if mix is not None else p32.astype(np.float64)
        rel = np.abs(mixed - ref_p64) / np.maximum(ref_p64, 1e-300)
        return float((rel > 1e-5).mean())
    except Exception:   # a reported figure, never a reason to lose the bench line
        return None


def parity_report_per_channel(jsg, c, plan, d_in_host, win):
    """parity_report for the per-channel workload (c4): one launch of the timed geometry (all channels, JSG_MIX_PER_CHANNEL), every channel's columns
    against the float64 DFT of its own float32 windowed frames, and its palette indices end to end.  Checker code: oracle/ (test infrastructure)."""
    import numpy as np
    import torch
    from oracle import jsg_oracle as oracle
    n, hop, C, F = c["n"], c["hop"], c["channels"], c["frames"]
    H = n // 2 + 1
    x = d_in_host[:, :(F - 1) * hop + n]
    d_x = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    pitch = (H + 31) // 32 * 32
    mix = jsg.capi.MIX_PER_CHANNEL
    d_pow = torch.empty((C, F, pitch), dtype=torch.float32, device="cuda")
    d_db = torch.empty((C, F, pitch), dtype=torch.float32, device="cuda")
    kernel = jsg.stft_kernel_name(plan, d_x, hop, F, d_pow, feedblocks=n // hop, mix_mode=mix)
    jsg.stft_db(plan, d_x, hop, F, d_pow, feedblocks=n // hop, mix_mode=mix, linear_out=True)
    jsg.stft_db(plan, d_x, hop, F, d_db, feedblocks=n // hop, mix_mode=mix)
    torch.cuda.synchronize()
    cols = np.unique(np.linspace(0, F - 1, min(F, 128)).astype(np.int64))
    idx = (cols * hop)[:, None] + np.arange(n)[None, :]
    frames = (x[:, idx] * win[None, None, :]).astype(np.float32)                       # [C][F'][n]
    ref_p = oracle.power_spectrum_f64(frames).astype(np.float32).astype(np.float64)   # [C][F'][H]
    got_p = d_pow[:, :, :H].cpu().numpy()[:, cols].astype(np.float64)
    rel = np.abs(got_p - ref_p) / np.maximum(ref_p, 1e-300)
    bad = rel > 1e-5
    cpu32 = float32_cpu_fft_share_beyond_1e5(oracle, frames, ref_p, None)
    peak = ref_p.max(axis=2, keepdims=True)
    level_db = 10.0 * np.log10(np.maximum(ref_p, 1e-300) / peak)
    strong = ref_p > 1e-2 * peak
    pal = oracle.OracleColorPalette(256, oracle.CM_JADE)
    pal.set_value_range(-50.0, 50.0)
    ref_db = oracle.to_db(ref_p.astype(np.float32))
    got_db = d_db[:, :, :H].cpu().numpy()[:, cols]
    flips = int((pal.index(got_db) != pal.index(ref_db)).sum())
    return {"kernel": kernel, "launch_checked": f"{F} columns x {C} channels, one column per channel (the timed geometry, one batch)",
            "columns_checked_against_float64": int(len(cols) * C), "bins_checked": int(rel.size),
            "frac_bins_rel_power_err_gt_1e-5": float(bad.mean()),
            "float32_cpu_fft_frac_bins_rel_power_err_gt_1e-5": cpu32,
            "those_bins_level_below_frame_peak_db": {"median": float(np.median(level_db[bad])) if bad.any() else None,
                                                     "highest": float(level_db[bad].max()) if bad.any() else None},
            "max_rel_power_err_bins_within_20dB_of_peak": float(rel[strong].max()),
            "max_err_relative_to_frame_peak": float((np.abs(got_p - ref_p) / peak).max()),
            "max_abs_db_err": float(np.abs(got_db.astype(np.float64) - ref_db.astype(np.float64)).max()),
            "colour_index_flips_end_to_end": flips, "pixels_checked": int(got_db.size),
            "note": "float64 DFT of the float32 windowed frames is the yardstick; indices: GPU dB and oracle dB through the oracle's CColorPalette "
                    "(Jade, 256 colours, -50..50 dB)"}


def power_report(torch, lib, stream, run_step, units_per_step, bytes_per_step, seconds, dev_index):
    """What the card draws while the path runs: the same steps for `seconds`, and the calibration copy for a shorter while, with the
    socket power and the core clock sampled from the card's hwmon files (tools/hwmon.py; read-only).  The STFT kernels run the socket
    INTO ITS POWER CAP (1400 W) with the core clock pulled down to 1.7-1.9 GHz, while a pure copy at 0.82 of 8 TB/s stays near 1100 W at
    the full 2.4 GHz: DESIGN.md section 6, 'power'.  Returns None when the telemetry files are not there."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from hwmon import Hwmon, Watch
    except ImportError:
        return None
    hw = Hwmon(dev_index)
    if not hw.ok:
        return None

    def leg(fn, per_call_units, per_call_bytes, secs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        calls, t0 = 0, time.perf_counter()
        with Watch(hw, settle_s=min(1.0, secs / 3)) as w:
            e0.record(stream)
            while time.perf_counter() - t0 < secs:
                for _ in range(4):
                    fn()
                calls += 4
                if calls % 16 == 0:
                    stream.synchronize()
            e1.record(stream)
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        r = {"seconds": round(ms / 1e3, 2), "GBps": per_call_bytes * calls / ms / 1e6, "frac_of_8TBps": per_call_bytes * calls / ms / 1e6 / HBM_PEAK_GBS}
        if per_call_units:
            r["units_per_s"] = per_call_units * calls / ms * 1e3
        r.update(w.summary())
        return r

    half = 1 << 30
    pool = torch.empty(2 * half // 4, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    src, dst, h = ctypes.c_void_p(pool.data_ptr()), ctypes.c_void_p(pool.data_ptr() + half), ctypes.c_void_p(stream.cuda_stream)
    out = {"cap_W": hw.cap_W(),
           "path_sustained": leg(run_step, units_per_step, bytes_per_step, seconds),
           "copy_sustained": leg(lambda: lib.jsg_calib_copy_launch(src, dst, half, h), 0, 2 * half, max(1.5, seconds / 2)),
           "source": os.path.join(hw.dir, hw.power_file),
           "note": "the same steps as the timed region, kept running; socket power and core clock are medians of 0.1 s samples after a settling "
                   "time.  At the cap the core clock is what gives: the path is power-bound there (DESIGN.md section 6)"}
    del pool
    return out


def calibrate_copy(lib, torch, stream):
    """The roof on THIS box: jsg_calib_copy_launch (float4, non-temporal loads and stores, non-looping grid) on buffers that rotate over
    2 x 1 GiB -- 1 GiB per launch, and the byte counts of one 65 536-frame launch (134 MB each way) and of one C2 batch (8.4 MB each way).
    Bytes read + bytes written / time of back-to-back launches on one stream (HIP events)."""
    pool = 1 << 30
    src = torch.empty(pool // 4, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    dst = torch.empty(pool // 4, dtype=torch.float32, device="cuda")
    out = {}
    for name, nbytes in (("1GiB", pool), ("134MB", 134348800), ("8.4MB", 8396800)):
        nrot = max(1, pool // nbytes)
        reps = max(6, min(300, (6 << 30) // nbytes))

        def go(i):
            off = (i % nrot) * nbytes
            rc = lib.jsg_calib_copy_launch(ctypes.c_void_p(src.data_ptr() + off), ctypes.c_void_p(dst.data_ptr() + off), nbytes, ctypes.c_void_p(stream.cuda_stream))
            assert rc == 0
        with torch.cuda.stream(stream):
            for i in range(3):
                go(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(reps):
                go(i)
            e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[name] = {"bytes_each_way": nbytes, "us": round(us, 2), "GBps": round(2 * nbytes / us / 1e3, 1), "frac_of_8p0": round(2 * nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
    del src, dst
    return {"kernel": "jsg_calib_copy_launch: float4 copy, non-temporal loads + stores, one thread per 16 bytes, read + written bytes / time",
            "peak_copy_GBps": out["1GiB"]["GBps"], "sizes": out,
            "guide": "MI355X_MICROARCH.md: 6.29 TB/s measured for a float4 copy (0.79 of the 8.0 TB/s spec)"}


def run_extra_config(cfg):
    """c3 / c5 as a child process (started before this process touches the GPU): their headline numbers for the driver's line."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-boundary",
                            "--no-extra", "--no-calibration", "--power-seconds", "2", "--sub-run"], capture_output=True, text=True, timeout=400)
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "step": j["config"]["step"],
                "workload": j["config"]["workload"], "kernel": j["roofline"]["kernel"],
                "roofline": {k: j["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_source", "avg_dispatch_us", "algorithmic_bytes_per_dispatch",
                                                           "timed_region_frac", "frac_rocprof", "traffic", "second_roof") if k in j["roofline"]},
                "parity": {k: j["parity"][k] for k in ("kernel", "frac_bins_rel_power_err_gt_1e-5", "max_err_relative_to_frame_peak", "colour_index_flips_end_to_end",
                                                       "pixels_checked", "exact_log_display_launch_vs_cpu_mirror", "strided_pixels_differing_from_single_launches",
                                                       "strided_columns_differing_from_single_launches")
                           if k in j.get("parity", {})},
                "power": ({"cap_W": j["power"].get("cap_W"), "path_sustained": j["power"].get("path_sustained")} if j.get("power") else None)}
    except Exception as e:   # a reported figure, never a reason to lose the bench line
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def spawn_ranks(n: int, argv) -> int:
    """Start `n` ranks of this script on this node (one per GPU) and return their exit code.  No torch import, no GPU call in this process."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="default: c2 (BASELINE configs[1]) on one GPU, c4 (configs[3]: 8 channels per GPU, one column per channel) with --gpus N > 1")
    ap.add_argument("--dispatches-per-step", type=int, default=0, help="kernel dispatches in one step (default: the configuration's)")
    ap.add_argument("--nbuf", type=int, default=0, help="batches (c5: images) per dispatch = distinct batches rotated through (default: ~1 GB worth)")
    ap.add_argument("--mode", choices=("strided", "single"), default="strided",
                    help="strided: one dispatch covers all batches of the rotation (default); single: one dispatch per batch, in order")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="workgroups per CU of a dispatch (0: the library's choice)")
    ap.add_argument("--layout", choices=("auto", "tail", "reference"), default="auto",
                    help="dB column layout of the timed dispatches: reference (= auto) = the reference's m_mem[col][bin]: bin n/2 inline, n/2+1 contiguous "
                         "floats per column (pitch rounded up to 32 floats); tail = jsg_stft_args.out_tail: columns of exactly n/2 floats + a dense "
                         "plane of bin n/2 (same values; the c2 line reports it as a side leg, roofline.frac_tail_plane)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the `parity` block (the profile scripts: its few launches would mix into the tracer's averages)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the `boundary` block (jsg_process_block latency, PCIe-inclusive rate)")
    ap.add_argument("--no-extra", action="store_true", help="skip the c3 / c5 child runs of the default line")
    ap.add_argument("--no-calibration", action="store_true", help="skip the copy-roof calibration")
    ap.add_argument("--no-power", action="store_true", help="skip the sustained leg that samples socket power and core clock")
    ap.add_argument("--power-seconds", type=float, default=3.0, help="length of the sustained leg (N=1 only; after the timed region)")
    ap.add_argument("--no-single", action="store_true", help="skip the one-batch-per-dispatch leg")
    ap.add_argument("--sub-run", action="store_true", help="(internal) this process is a child of another bench.py")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work at all: exercises the N-rank plumbing (barriers, reductions, "
                                                           "JSON) on a machine without GPUs; value is null")
    args = ap.parse_args()
    if args.config is None:
        args.config = "c2" if args.gpus <= 1 else "c4"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher around it: this parent -- which has not imported torch and never touches a GPU -- starts
        # the N ranks as fresh processes (torch.distributed.run -> N x `bench.py --gpus N ...`), lets rank 0's JSON line through and exits with
        # their code.  (Never an exec from a process that has initialised the GPU; the parent has not.)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    c = dict(CONFIGS[args.config])
    dps = args.dispatches_per_step or c["dispatches_per_step"]

    import numpy as np
    import torch

    # child processes (git, make for a stale C oracle, the c3 / c5 runs, the latency test) are started BEFORE anything initialises the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    commit = _git_commit()
    if not args.dry_run and not args.no_cpu_baseline and world == 1:   # the CPU baseline leg runs at N = 1 only
        from oracle import oracle_c
        oracle_c.load()
    extra = None
    if not args.dry_run and not args.no_extra and world == 1 and not args.sub_run and args.config == "c2":
        extra = {cfg: run_extra_config(cfg) for cfg in ("c4", "c3", "c5")}
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
    # JSG_BENCH_DIST_SINGLE=1: a ONE-rank process group on the real backend (RCCL on the GPU box) -- rehearses exactly the calls the N > 1 runs
    # make (init with device_id, barrier, max-reduction of a CUDA tensor, destroy) where only one GPU is there (tests/test_bench_cli.py)
    if world > 1 or os.environ.get("JSG_BENCH_DIST_SINGLE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.setdefault("MASTER_PORT", str(port)); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
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
    fb = n // hop
    algo_batch = algorithmic_bytes_per_batch(c)
    units_per_batch = F if c["colour"] else F * C      # columns for c5, frames (FFTs) otherwise
    strided = args.mode == "strided"

    run_step = None
    nbuf = args.nbuf or c["batches_per_dispatch"]
    bpd = nbuf if strided else 1                      # batches per dispatch
    dispatch_us = single_us = None
    parity = calibration = None
    kernel_label = None
    if not args.dry_run:
        import jadespectrogram_amd as jsg
        from jadespectrogram_amd import capi
        lib = capi.lib()
        win = jsg.window(jsg.capi.WIN_HANN, n)
        plan = jsg.Plan(n, win)
        n_samples = (F * hop + (n - hop) + 3) // 4 * 4
        per_ch = bool(c.get("per_channel"))
        rows = C if per_ch else 1                         # output planes of one batch
        use_tail = (not c["colour"]) and args.layout == "tail"      # (auto = reference: the reference's m_mem[col][bin], Spectrogram.h:144)
        ref_pitch = (H + 31) // 32 * 32
        pitch = n // 2 if use_tail else ref_pitch
        img_pitch = (F + 31) // 32 * 32
        per_batch = C * n_samples * 4 + (H * img_pitch * 4 if c["colour"] else rows * (F * pitch * 4 + (F * 4 if use_tail else 0)))
        if not args.nbuf and abs(nbuf * per_batch - ROTATION_BYTES) > 0.25 * ROTATION_BYTES:   # (a changed geometry: keep ~1 GB)
            nbuf = max(2, int(ROTATION_BYTES // per_batch) + 1)
            bpd = nbuf if strided else 1
        base = synth_audio(C, n_samples + nbuf * 64, fs=c["fs"], seed=1234 + 1000 * rank, first_channel=(rank * C if per_ch else 0))   # SURVEY 8d signal
        d_in = torch.empty((nbuf, C, n_samples), dtype=torch.float32, device="cuda")          # batch b = d_in[b]
        for b in range(nbuf):
            d_in[b].copy_(torch.from_numpy(np.ascontiguousarray(base[:, b * 64:b * 64 + n_samples])))
        d_img = torch.zeros((nbuf, H, img_pitch), dtype=torch.int32, device="cuda") if c["colour"] else None
        out_shape = (nbuf, C, F, pitch) if per_ch else (nbuf, F, pitch)
        d_out = None if c["colour"] else torch.empty(out_shape, dtype=torch.float32, device="cuda")
        d_tail = torch.empty((nbuf, rows, F), dtype=torch.float32, device="cuda") if use_tail else None
        d_lut = torch.from_numpy(jsg.colormap_lut(256, jsg.capi.CM_JADE)).cuda() if c["colour"] else None
        mixk = dict(feedblocks=fb, mix_mode=(jsg.capi.MIX_PER_CHANNEL if per_ch else jsg.capi.MIX_ABSMEAN))
        one = torch.cuda.Stream()
        if c["colour"]:
            if strided:
                two = jsg.stft_image_strided_needs_scratch(plan, d_in[:bpd], hop, F, d_lut, -50.0, 50.0, d_img[:bpd, :, :F], None, **mixk)
            else:
                two = jsg.stft_image_needs_scratch(plan, d_in[0], hop, F, d_lut, -50.0, 50.0, d_img[0][:, :F], None, plan_select=2, **mixk)
            assert not two, "the C5 launches are expected to take the single-kernel form"
            kernel_label = "stft_db_kernel<Cfg4096B, AbsMean, ARGB out> (the workgroups colour their own columns)"
        else:
            kname = jsg.stft_db_strided_kernel_name(plan, d_in[:bpd], hop, F, d_out[:bpd], d_tail=(d_tail[:bpd] if use_tail else None), **mixk)
            runs = bpd > 1 and kname == "Cfg1024" and 2 * hop == n and (C == 1 or per_ch)      # the launcher's rule (jsg_kernels.hip: `runs`)
            kernel_label = (f"stft_db_kernel<{kname}, {'one channel per column' if C == 1 or per_ch else 'AbsMean'}, dB out{', strided' if bpd > 1 else ''}"
                            f"{', runs of 2 consecutive columns per wavefront' if runs else ''}{', tail plane' if use_tail else ''}>")

        def dispatch(stream_handle, b=0):
            if c["colour"] and strided:
                jsg.stft_image_strided(plan, d_in, hop, F, d_lut, -50.0, 50.0, d_img[:, :, :F], None, stream=stream_handle, **mixk)
            elif c["colour"]:
                jsg.stft_image(plan, d_in[b], hop, F, d_lut, -50.0, 50.0, d_img[b][:, :F], None, plan_select=2, stream=stream_handle, **mixk)
            elif strided:
                jsg.stft_db_strided(plan, d_in, hop, F, d_out, d_tail=d_tail, blocks_per_cu=args.blocks_per_cu, stream=stream_handle, **mixk)
            else:
                jsg.stft_db(plan, d_in[b], hop, F, d_out[b], d_tail=(d_tail[b] if use_tail else None), blocks_per_cu=args.blocks_per_cu, stream=stream_handle, **mixk)

        def run_step():
            h = one.cuda_stream
            for i in range(dps):
                dispatch(h, i % nbuf)
    else:
        def run_step():
            time.sleep(0.001)

    # ---- settle the clocks (about 0.5 s of the same work), W warm-up steps, then EXACTLY K timed steps ----
    prewarm_s = 0.0
    ev0 = ev1 = None
    if not args.dry_run:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.5:
            run_step()
            sync()
        prewarm_s = time.perf_counter() - t_pre
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.warmup):
        run_step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    if ev0 is not None:
        ev0.record(one)
    for _ in range(args.steps):
        run_step()
    if ev1 is not None:
        ev1.record(one)
    sync()
    t1 = time.perf_counter()
    barrier(); sync()
    wall = t1 - t0
    if ev0 is not None:
        dispatch_us = ev0.elapsed_time(ev1) * 1e3 / (args.steps * dps)     # per-dispatch duration of the timed region itself

    if not args.dry_run and strided and not args.no_single and not c["colour"]:
        # ---- the literal "4096 frames/launch" of the configuration: one dispatch per batch, in order on the same stream, a hipGraph
        #      replay of the whole rotation (the host out of the picture), HIP events on that stream ----
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(one):
            for b in range(nbuf):
                jsg.stft_db(plan, d_in[b], hop, F, d_out[b], d_tail=(d_tail[b] if use_tail else None), stream=one.cuda_stream, **mixk)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=one):
                for b in range(nbuf):
                    jsg.stft_db(plan, d_in[b], hop, F, d_out[b], d_tail=(d_tail[b] if use_tail else None), stream=one.cuda_stream, **mixk)
            g.replay(); g.replay()
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(one)
            for _ in range(8):
                g.replay()
            s1.record(one)
        torch.cuda.synchronize()
        single_us = s0.elapsed_time(s1) * 1e3 / (8 * nbuf)
    other_layout_us = None
    if not args.dry_run and strided and not args.no_single and not c["colour"] and not c.get("per_channel") and world == 1:
        # ---- the same dispatches with the OTHER column layout, measured in this process on this box.  The timed region writes the reference's
        #      m_mem[col][bin] (bin n/2 inline: columns of n/2+1 floats at a pitch rounded up to 32 floats); the other one is the tail plane of
        #      jsg_stft_args.out_tail (columns of exactly n/2 floats = whole 128-byte lines + a dense plane of bin n/2; same values) -- or the
        #      reverse under --layout tail ----
        if use_tail:
            d_o, d_ot = torch.empty((nbuf, F, ref_pitch), dtype=torch.float32, device="cuda"), None
        else:
            d_o, d_ot = torch.empty((nbuf, F, n // 2), dtype=torch.float32, device="cuda"), torch.empty((nbuf, 1, F), dtype=torch.float32, device="cuda")
        with torch.cuda.stream(one):
            t_ref = time.perf_counter()
            while time.perf_counter() - t_ref < 0.3:        # the same settling as the main region: ~0.3 s of the work, then 5 timed steps
                for _ in range(dps):
                    jsg.stft_db_strided(plan, d_in, hop, F, d_o, d_tail=d_ot, blocks_per_cu=args.blocks_per_cu, stream=one.cuda_stream, **mixk)
                torch.cuda.synchronize()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record(one)
            for _ in range(5 * dps):
                jsg.stft_db_strided(plan, d_in, hop, F, d_o, d_tail=d_ot, blocks_per_cu=args.blocks_per_cu, stream=one.cuda_stream, **mixk)
            r1.record(one)
        torch.cuda.synchronize()
        other_layout_us = r0.elapsed_time(r1) * 1e3 / (5 * dps)
        if use_tail:
            layouts_same = bool(torch.equal(d_o[..., :n // 2], d_out) and torch.equal(d_o[..., n // 2], d_tail[:, 0, :]))
        else:
            layouts_same = bool(torch.equal(d_out[..., :n // 2], d_o) and torch.equal(d_out[..., n // 2], d_ot[:, 0, :]))
        del d_o, d_ot
    barrier(); sync()
    if dist is not None:
        t = torch.tensor([wall, dispatch_us or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dispatch_us = float(t[0]), (float(t[1]) or None)

    power = None
    if rank == 0 and not args.dry_run and world == 1 and not args.no_power:
        try:   # (a reported figure, never a reason to lose the bench line)
            power = power_report(torch, lib, one, run_step, units_per_batch * bpd * dps, algo_batch * bpd * dps, args.power_seconds, dev_index)
        except Exception as e:
            power = {"error": f"{type(e).__name__}: {e}"[:200]}
            torch.cuda.synchronize()
    if rank == 0 and not args.dry_run and world == 1 and not args.no_calibration:
        calibration = calibrate_copy(lib, torch, one)
    if rank == 0 and not args.dry_run and world == 1 and not args.no_parity:
        parity = parity_report(jsg, c, plan, base, win)
        # the timed strided dispatch against one-by-one launches of the same batches (2048 / 4096 points: the same kernel plan)
        with torch.cuda.stream(one):
            if c["colour"]:
                dispatch(one.cuda_stream)
                tmp = torch.zeros((H, img_pitch), dtype=torch.int32, device="cuda")
                differing = 0
                for k in sorted({0, nbuf // 2, nbuf - 1}):
                    tmp.zero_()
                    jsg.stft_image(plan, d_in[k], hop, F, d_lut, -50.0, 50.0, tmp[:, :F], None, plan_select=2, stream=one.cuda_stream, **mixk)
                    differing += int((tmp != d_img[k]).sum())
                parity["strided_pixels_differing_from_single_launches"] = differing
            elif strided:
                d_out.fill_(-7.0)
                dispatch(one.cuda_stream)
                pin = 0 if n not in (2048, 4096) else (2 if "B," in kernel_label else 1)
                tmp = torch.empty(((C, F, ref_pitch) if per_ch else (F, ref_pitch)), dtype=torch.float32, device="cuda")      # single launches in the REFERENCE layout
                differing = 0
                for k in sorted({0, nbuf // 2, nbuf - 1}):
                    tmp.fill_(-7.0)
                    jsg.stft_db(plan, d_in[k], hop, F, tmp, plan_select=pin, stream=one.cuda_stream, **mixk)
                    if use_tail:
                        differing += int((tmp[..., :n // 2] != d_out[k]).sum()) + int((tmp[..., n // 2] != d_tail[k].reshape(tmp[..., n // 2].shape)).sum())
                    else:
                        differing += int((tmp != d_out[k]).sum())
                parity["strided_columns_differing_from_single_launches"] = differing
        torch.cuda.synchronize()

    units_total = world * args.steps * dps * bpd * units_per_batch
    algo = algo_batch * bpd                            # algorithmic bytes of one dispatch
    per_ch = bool(c.get("per_channel"))
    unit_word = (f"{C} channels x {F} frames" if per_ch else ("columns" if c["colour"] or C > 1 else "frames"))
    workload = c["workload"]
    if per_ch and world > 1:
        workload += f" -- this run: {world} GPUs x {C} channels = {world * C} channels"
    out = {
        "metric": c["metric"], "value": None if args.dry_run else units_total / wall, "unit": c["unit"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "step": f"{dps} dispatches x {bpd} {'images' if c['colour'] else 'batches'} x {'' if per_ch else F}{'' if per_ch else ' '}{unit_word} = {dps * bpd * units_per_batch} {c['unit'].split('/')[0]} per step and GPU",
                   "dispatches_per_step": dps, "batches_per_dispatch": bpd, "frames_per_batch": F * C, "columns_per_batch": F, "frames_per_dispatch": F * C * bpd,
                   "channels_per_gpu": C, "channels_total": world * C if per_ch else C, "per_gpu_value": None if args.dry_run else units_total / wall / world,
                   "distinct_batches": nbuf, "rotation_bytes": None if args.dry_run else int(nbuf * per_batch),
                   "hip_streams_per_gpu": 1, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "column_layout": None if args.dry_run else ("n/2 floats per column + dense plane of bin n/2 (jsg_stft_args.out_tail)" if use_tail else
                                                               ("ARGB image rows" if c["colour"] else "reference: n/2+1 floats per column, pitch rounded up to 32 floats")),
                   "issue": (("jsg_stft_image_launch_strided" if c["colour"] else "jsg_stft_db_launch_strided") +
                             f": K = {bpd} independent {'images' if c['colour'] else f'{F}-frame batches'} per kernel dispatch, the dispatches of a step back to back "
                             "on ONE stream, issued by plain C calls (no hipGraph, no extra streams, no hardware-queue setting)") if strided else
                            "one dispatch per batch, in order on one stream, host-issued",
                   "prewarm_s": round(prewarm_s, 3),
                   "parallelism": f"{world} GPU(s), independent batches, no data-path collective"},
    }
    if args.dry_run:
        out["dry_run"] = True
    if dispatch_us:
        achieved = algo / (dispatch_us * 1e-6) / 1e9
        conc = algo * dps * args.steps / wall / 1e9
        # HBM traffic (PMC) and the tracer's per-dispatch duration come from the rocprofv3 passes of tools/profile_bench.sh
        # (profiles/r06_<cfg>_hbm_traffic.json): bench.py does not run counters itself.  The file carries the hash of the kernel sources
        # it was recorded with; when that differs from this build's the figures are flagged as stale.
        traffic, tsrc, rocprof_us = None, None, None
        prof = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{args.config}_hbm_traffic.json")
        if os.path.exists(prof) and strided:
            try:
                pj = json.load(open(prof))
                same = pj.get("kernel_source_sha") == kernel_source_sha() and int(pj.get("batches_per_dispatch") or 1) == bpd
                traffic = pj.get("hbm_bytes_per_launch")
                rocprof_us = pj.get("avg_us") if same else None
                tsrc = {"file": os.path.relpath(prof, ROOT), "recorded_at_commit": pj.get("commit"), "kernel_source_sha": pj.get("kernel_source_sha"),
                        "matches_this_build": same, "rocprof_avg_dispatch_us": pj.get("avg_us"),
                        "note": "rocprofv3 passes of tools/profile_bench.sh over this same command (kernel trace; FETCH_SIZE x2 on gfx950 + WRITE_SIZE in separate "
                                "PMC passes), recorded earlier; NOT measured by this run" + ("" if same else " -- STALE: the kernel sources have changed since")}
            except Exception:
                traffic = None
        frac = achieved / HBM_PEAK_GBS
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
            "frac_source": "this run: algorithmic bytes per dispatch / average dispatch duration of the timed region (HIP events on the launch stream around the K steps)",
            "avg_dispatch_us": dispatch_us, "algorithmic_bytes_per_dispatch": algo, "algorithmic_bytes_per_batch": algo_batch,
            "timed_region_achieved": conc, "timed_region_frac": conc / HBM_PEAK_GBS,
            "frac_rocprof": (algo / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None,
            "frac_of_measured_copy_roof": (achieved / calibration["peak_copy_GBps"]) if calibration else None,
            "traffic": traffic, "traffic_source": tsrc, "kernel": kernel_label,
            "second_roof": valu_roof(c, units_per_batch * bpd, dispatch_us, ((power or {}).get("path_sustained") or {}).get("sclk_MHz_median")),
            "commit": commit,
        }
        r = out["roofline"]
        # ---- the qualifiers of the headline as SCALARS of `roofline` (a record that keeps only scalars keeps these) ----
        r["column_layout"] = "tail_plane" if use_tail else ("argb_image" if c["colour"] else "reference")
        r["second_roof_frac"] = r["second_roof"]["frac"]
        r["second_roof_clock_GHz"] = r["second_roof"]["clock_GHz"]
        r["traffic_over_algorithmic"] = (traffic / algo) if traffic else None
        r["traffic_matches_this_build"] = bool(tsrc and tsrc["matches_this_build"])
        if not c["colour"]:
            r["frac_reference_layout" if not use_tail else "frac_tail_plane"] = frac
        if other_layout_us:
            of = algo / (other_layout_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            r["frac_tail_plane" if not use_tail else "frac_reference_layout"] = of
            r["other_column_layout"] = {
                "what": ("the same strided dispatches writing columns of n/2 floats + a dense plane of bin n/2 (jsg_stft_args.out_tail) instead of the "
                         "reference's m_mem[col][bin] (bin n/2 inline: a 4-byte piece in one more 128-byte line per column); same values; a host that "
                         "wants the reference shape back pays one more pass (jsg_columns_from_tail_layout_launch)") if not use_tail else
                        "the same strided dispatches writing the reference's column layout (bin n/2 inline) instead of the tail plane; same values",
                "layout": "reference" if use_tail else "tail_plane",
                "avg_dispatch_us": other_layout_us, "frac": of, "units_per_s": units_per_batch * bpd / (other_layout_us * 1e-6),
                "values_identical_to_the_timed_layout": layouts_same}
        if single_us:
            r["frac_one_batch_per_dispatch"] = algo_batch / (single_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            r["one_batch_per_dispatch_units_per_s"] = units_per_batch / (single_us * 1e-6)
            r["one_batch_per_dispatch"] = {
                "what": f"the same batches, ONE jsg_stft_db_launch per {F}-frame batch, in order on one stream (hipGraph replay of the rotation, HIP events): "
                        "the literal 'batched 4096 frames/launch' of BASELINE configs[1]",
                "avg_dispatch_us": single_us, "algorithmic_bytes_per_dispatch": algo_batch, "frac": r["frac_one_batch_per_dispatch"],
                "units_per_s": units_per_batch / (single_us * 1e-6)}
        if extra is not None:
            for cfg in ("c3", "c4", "c5"):
                e = extra.get(cfg) or {}
                r[f"extra_{cfg}_frac"] = (e.get("roofline") or {}).get("frac")
                r[f"extra_{cfg}_value"] = e.get("value")
                r[f"extra_{cfg}_second_roof_frac"] = ((e.get("roofline") or {}).get("second_roof") or {}).get("frac")
    if power is not None:
        out["power"] = power
    if calibration is not None:
        out["calibration"] = calibration
    if parity is not None:
        out["parity"] = parity
    if extra is not None:
        out["extra"] = extra
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
