"""GPU parity: the HIP path (through the C-ABI) against the oracle on the same seeded inputs."""
import os

import numpy as np
import pytest

from parity_util import assert_db_close, assert_power_close, mixed_power_f64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _padded(x, n):
    return np.concatenate([np.zeros((x.shape[0], n), np.float32), x], axis=1)


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("kind", ["mix", "noise"])
def test_linear_power_vs_float64_dft(jsg, oracle, torch_cuda, n, kind):
    """|X|^2 of every frame against the float64 DFT of the float32 windowed frame (mono, Hann, 50 % hop)."""
    torch = torch_cuda
    hop, K = n // 2, 12
    x = oracle.synth_audio(1, K * n, kind=kind)
    win = oracle.window(oracle.WIN_HANN, n)
    plan = jsg.Plan(n, win)
    F = 2 * K
    d_in = torch.from_numpy(_padded(x, n)).cuda()
    d_out = torch.full((F, n // 2 + 1 + 7), -1.0, device="cuda")
    jsg.stft_db(plan, d_in, hop, F, d_out, linear_out=True)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert (got[:, n // 2 + 1:] == -1.0).all(), "wrote outside the column"
    ref = oracle.stft_db_reference(x, n, hop, 2, win, return_power=True)[0]
    worst = assert_power_close(got[:, :n // 2 + 1], ref, f"n={n} {kind}")
    assert worst < 1e-5, worst


CASES = [
    # n, feed, channels, mix, window
    (1024, 1, 1, 0, 1),      # C1/C2: mono 1024 / 512 hop / Hann
    (1024, 1, 2, 0, 1),      # the plugin's stereo bus
    (2048, 2, 8, 0, 1),      # C3: 8 ch, 2048, 75 % overlap
    (1024, 1, 8, 0, 3),      # C4 per-GPU shard, BlackmanHarris
    (2048, 1, 2, 0, 1),      # the plugin's prepareToPlay defaults
    (1024, 0, 2, 1, 0),      # Max, Rect, no overlap
    (1024, 2, 3, 2, 2),      # Min, Hamming
    (4096, 1, 2, 3, 4),      # Left, FlatTop
    (512, 1, 2, 4, 5),       # Right, HannPoisson, two frames per wave
    (8192, 1, 1, 0, 1),
    (1024, 3, 2, 0, 1),      # perc10: irregular hop 102 x9 + 106
    (2048, 3, 1, 0, 1),      # perc10 with an odd hop (205): 4-byte aligned pair loads
    (512, 3, 1, 0, 1),
    (1024, 1, 3, 0, 1),      # AbsMean over a non-power-of-two channel count: IEEE division path
    (1024, 1, 5, 0, 1),
    (1024, 1, 64, 0, 1),     # the full 64-channel mix of configs[3] on one GPU
    (1024, 0, 2, 0, 1),      # perc100: no overlap
    # 2048 points with three or more channels mixed into a column: the two-stage plan
    (2048, 1, 4, 1, 0),      # Max, Rect
    (2048, 2, 5, 2, 2),      # Min, Hamming
    (2048, 0, 3, 0, 3),      # AbsMean over three channels (IEEE division), no overlap
    (2048, 3, 4, 0, 1),      # perc10: odd hop 205
    # 4096 points with three or more channels: the one-wavefront-per-frame plan with factorised tables (Cfg4096B)
    (4096, 1, 3, 0, 1),      # AbsMean over three channels (IEEE division)
    (4096, 2, 4, 1, 5),      # Max, HannPoisson
    (4096, 0, 6, 2, 3),      # Min, BlackmanHarris, no overlap
    (4096, 3, 8, 0, 2),      # perc10: hop 410
]


@pytest.mark.parametrize("n,feed,channels,mix,win", CASES)
def test_engine_db_columns(jsg, oracle, n, feed, channels, mix, win):
    K = 6
    x = oracle.synth_audio(channels, K * n, seed=77)
    s = jsg.Spectrogram(channels)
    s.setSamplerate(48000.0); s.setmemoryTime_s(2.0); s.setFFTSize(n); s.setfeed_percent(feed)
    s.setWindow(win); s.setMixMode(mix)
    o = oracle.OracleSpectrogram(channels)
    o.set_samplerate(48000.0); o.set_memory_time_s(2.0); o.set_fft_size(n); o.set_feed_percent(feed)
    o.set_window(win); o.mode = mix
    assert (s.getMemorySize(), s.getSpectrumSize(), s.getFeedSamples()) == (o.memsize_blocks, o.freqsize, o.hop)
    assert (s.getWindow().view(np.uint32) == o.window.view(np.uint32)).all()
    s.processBlocks(x)
    mem = np.zeros((s.getMemorySize(), s.getSpectrumSize()), np.float32)
    nv, pos = s.getMem(mem)
    F = K * o.feedblocks
    assert pos == F % o.memsize_blocks and nv == oracle.NEW_ENTRY_SENTINEL + F
    ref = oracle.stft_db_reference(x, n, o.hop, o.feedblocks, o.window, mode=mix)
    pw = mixed_power_f64(oracle, x, n, o.hop, o.feedblocks, o.window, mix)
    assert (mem[0] == np.float32(-110.0)).all(), "first column of a fresh engine is the all-zero frame"
    peak = None
    if mix != 0:   # a selecting mix: the yardstick is the largest per-channel peak of the frame (parity_util.assert_db_close)
        pc = oracle.stft_db_reference(x, n, o.hop, o.feedblocks, o.window, return_power=True)
        peak = pc.astype(np.float64).max(axis=(0, 2))[:, None]
    assert_db_close(mem[:F], ref, pw, f"n={n} feed={feed} C={channels} mix={mix} win={win}", peak=peak)
    assert (mem[F:] == np.float32(-120.0)).all(), "untouched ring columns keep the -120 dB fill"
    s.close()


def test_per_channel_mode_equals_single_channel_engines(jsg, oracle):
    C, n, K = 4, 1024, 4
    x = oracle.synth_audio(C, K * n, seed=5)
    s = jsg.Spectrogram(C)
    s.setSamplerate(48000.0); s.setFFTSize(n); s.setfeed_percent(1); s.setMixMode(jsg.capi.MIX_PER_CHANNEL)
    s.processBlocks(x)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    mem = np.zeros((C * W, H), np.float32)
    s.getMem(mem)
    mem = mem.reshape(C, W, H)
    for c in range(C):
        m = jsg.Spectrogram(1)
        m.setSamplerate(48000.0); m.setFFTSize(n); m.setfeed_percent(1)
        m.processBlocks(x[c:c + 1])
        one = np.zeros((W, H), np.float32)
        m.getMem(one)
        assert (one.view(np.uint32) == mem[c].view(np.uint32)).all(), "sharding by channel must be bit-identical"
        m.close()
    s.close()


def test_block_by_block_stream_pause_and_getmem(jsg, oracle):
    """processSynchronBlock per block with getMem in between, a pause, and more than one ring wrap."""
    C, n = 2, 1024
    s = jsg.Spectrogram(C); o = oracle.OracleSpectrogram(C)
    for e in (s,):
        e.setSamplerate(48000.0); e.setmemoryTime_s(0.25); e.setFFTSize(n); e.setfeed_percent(1)
    o.set_samplerate(48000.0); o.set_memory_time_s(0.25); o.set_fft_size(n); o.set_feed_percent(1)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    assert W == o.memsize_blocks == 23
    x = oracle.synth_audio(C, 40 * n, seed=9)
    mem_g = np.zeros((W, H), np.float32); mem_o = np.zeros((W, H), np.float32)
    assert s.getMem(np.zeros((W + 1, H), np.float32))[0] == -1          # reference: size mismatch -> -1
    for b in range(40):
        if b == 10:
            s.setPauseMode(True); o.set_pause_mode(True)
        if b == 14:
            s.setPauseMode(False); o.set_pause_mode(False)
        blk = x[:, b * n:(b + 1) * n]
        assert s.processSynchronBlock(blk) == 0
        o.process_synchron_block(blk)
        if b % 4 == 3 or b == 39:
            nv_g, pos_g = s.getMem(mem_g)
            nv_o, pos_o = o.get_mem(mem_o)
            assert (nv_g, pos_g) == (nv_o, pos_o), b
            d = np.abs(mem_g.astype(np.float64) - mem_o.astype(np.float64))
            assert d.max() < 2e-3, (b, d.max())   # coarse here; exact tolerances are checked in test_engine_db_columns
    # batch == block-by-block, bit for bit
    s2 = jsg.Spectrogram(C)
    s2.setSamplerate(48000.0); s2.setmemoryTime_s(0.25); s2.setFFTSize(n); s2.setfeed_percent(1)
    s2.processBlocks(x[:, :10 * n]); s2.processBlocks(x[:, 14 * n:])
    mem2 = np.zeros((W, H), np.float32)
    s2.getMem(mem2)
    full = np.zeros((W, H), np.float32)
    o2 = jsg.Spectrogram(C)
    o2.setSamplerate(48000.0); o2.setmemoryTime_s(0.25); o2.setFFTSize(n); o2.setfeed_percent(1)
    for b in list(range(10)) + list(range(14, 40)):
        o2.processSynchronBlock(x[:, b * n:(b + 1) * n])
    o2.getMem(full)
    assert (full.view(np.uint32) == mem2.view(np.uint32)).all()
    for e in (s, s2, o2):
        e.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("JSG_FUZZ_SCENARIOS", "10")))))
def test_seeded_random_engine_scenarios(jsg, oracle, seed):
    """Random walks over the engine's setter / process / pause / getMem surface, mirrored on the oracle engine: the column
    counters and write positions must agree exactly after every getMem (every setter wipes the history like the
    reference's buildmem), the ring contents within the coarse bound used above (fine tolerances: test_engine_db_columns)."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(1, 5))
    s = jsg.Spectrogram(C); o = oracle.OracleSpectrogram(C)
    n = int(rng.choice([512, 1024, 2048]))
    s.setSamplerate(48000.0); o.set_samplerate(48000.0)
    s.setmemoryTime_s(0.2); o.set_memory_time_s(0.2)
    s.setFFTSize(n); o.set_fft_size(n)
    x = oracle.synth_audio(C, 64 * 4096, seed=seed, kind="mix")
    at = 0
    for step in range(60):
        ev = rng.choice(["block"] * 12 + ["getmem"] * 3 + ["pause", "feed", "window", "mix", "fft", "memtime"])
        if ev == "block":
            if at + n > x.shape[1]:
                at = 0
            blk = x[:, at:at + n]; at += n
            assert s.processSynchronBlock(blk) == 0
            o.process_synchron_block(blk)
        elif ev == "pause":
            p = bool(rng.integers(0, 2)); s.setPauseMode(p); o.set_pause_mode(p)
        elif ev == "feed":
            f = int(rng.integers(0, 4)); s.setfeed_percent(f); o.set_feed_percent(f)
        elif ev == "window":
            w = int(rng.integers(0, 6)); s.setWindow(w); o.set_window(w)
        elif ev == "mix":
            m = int(rng.integers(0, 5 if C > 1 else 4)); s.setMixMode(m); o.mode = m
        elif ev == "fft":
            n = int(rng.choice([512, 1024, 2048, 4096])); s.setFFTSize(n); o.set_fft_size(n)
        elif ev == "memtime":
            t = float(rng.choice([0.1, 0.2, 0.5])); s.setmemoryTime_s(t); o.set_memory_time_s(t)
        if ev == "getmem" or step == 59:
            W, H = s.getMemorySize(), s.getSpectrumSize()
            assert (W, H, s.getFeedSamples()) == (o.memsize_blocks, o.freqsize, o.hop), (seed, step)
            mg = np.zeros((W, H), np.float32); mo = np.zeros((W, H), np.float32)
            assert s.getMem(mg) == o.get_mem(mo), (seed, step)
            # this test is about the state machine: bins within 60 dB of their column's peak to 2e-3 dB, the weak rest
            # (deep window side lobes, where two float32 FFTs differ visibly) only coarsely
            d = np.abs(mg.astype(np.float64) - mo.astype(np.float64))
            strong = mo > (mo.max(axis=1, keepdims=True) - 60.0)
            assert d[strong].max() < 2e-3 and d.max() < 0.5, (seed, step, float(d[strong].max()), float(d.max()))
    s.close()


@pytest.mark.parametrize("n", [2048, 4096])
def test_engine_ring_does_not_depend_on_batching_with_many_channels(jsg, oracle, n):
    """Four channels at 2048 / 4096 points: a stateless launch of this size would switch to the large-workgroup kernel, the
    engine must not (its ring may not depend on how the host cut the stream into calls): one call with 1024 blocks, calls
    of 7 blocks and single blocks leave the same bits."""
    C, blocks = 4, 1024
    x = oracle.synth_audio(C, blocks * n, seed=3)
    rings = []
    for step in (blocks, 7, 1):
        s = jsg.Spectrogram(C)
        s.setSamplerate(48000.0); s.setmemoryTime_s(20.0); s.setFFTSize(n); s.setfeed_percent(2)
        for b in range(0, blocks, step):
            s.processBlocks(x[:, b * n:min(blocks, b + step) * n])
        mem = np.zeros((s.getMemorySize(), s.getSpectrumSize()), np.float32)
        nv, pos = s.getMem(mem)
        rings.append((nv, pos, mem))
        s.close()
    for nv, pos, mem in rings[1:]:
        assert (nv, pos) == rings[0][:2]
        assert (mem.view(np.uint32) == rings[0][2].view(np.uint32)).all()


def test_silence_and_full_scale(jsg, oracle):
    n = 1024
    s = jsg.Spectrogram(1)
    s.setSamplerate(48000.0); s.setFFTSize(n); s.setfeed_percent(1)
    s.processBlocks(np.zeros((1, 4 * n), np.float32))
    mem = np.zeros((s.getMemorySize(), 513), np.float32)
    s.getMem(mem)
    assert (mem[:8] == np.float32(-110.0)).all()        # 10*log10(1e-11f)
    t = np.arange(4 * n)
    x = np.sin(2 * np.pi * 1000.0 * t / 48000.0).astype(np.float32)[None]
    s.processBlocks(x)
    s.getMem(mem)
    assert abs(mem[8 + 4, 21] - 51.7976) < 2e-3 and abs(mem[8 + 4, 22] - 49.8594) < 2e-3
    s.close()


def test_power_scale_and_custom_window(jsg, oracle):
    n = 1024
    x = oracle.synth_audio(1, 4 * n, seed=3)
    s = jsg.Spectrogram(1)
    s.setSamplerate(48000.0); s.setFFTSize(n); s.setfeed_percent(1); s.setPowerScale(1.0 / n)
    s.processBlocks(x)
    mem = np.zeros((s.getMemorySize(), 513), np.float32); s.getMem(mem)
    win = oracle.window(1, n)
    ref = oracle.stft_db_reference(x, n, 512, 2, win, power_scale=1.0 / n)
    pw = mixed_power_f64(oracle, x, n, 512, 2, win, 0, power_scale=1.0 / n)
    assert_db_close(mem[:8], ref, pw, "power_scale 1/N")
    w2 = (win * np.linspace(0.5, 1.5, n)).astype(np.float32)
    s.setPowerScale(1.0); s.setFFTSize(n); s.setfeed_percent(1); s.setWindowTable(w2)
    s.processBlocks(x); s.getMem(mem)
    ref = oracle.stft_db_reference(x, n, 512, 2, w2)
    pw = mixed_power_f64(oracle, x, n, 512, 2, w2, 0)
    assert_db_close(mem[:8], ref, pw, "custom window")
    s.close()


def test_unsupported_sizes_fail_loudly(jsg):
    s = jsg.Spectrogram(1)
    with pytest.raises(jsg.JsgError) as ei:
        s.setFFTSize(1000)
    assert ei.value.code == jsg.capi.JSG_ERR_UNSUPPORTED
    with pytest.raises(jsg.JsgError):
        s.setMixMode(4)     # Right with one channel: out-of-bounds read in the reference
    s.close()


def test_batch_larger_than_ring_keeps_newest_columns(jsg, oracle):
    """One processBlocks call with more frames than ring columns: only the newest W columns survive, in ring order
    (a single launch must never write one column twice)."""
    C, n = 1, 1024
    s = jsg.Spectrogram(C); o = oracle.OracleSpectrogram(C)
    s.setSamplerate(48000.0); s.setmemoryTime_s(0.2); s.setFFTSize(n); s.setfeed_percent(2)
    o.set_samplerate(48000.0); o.set_memory_time_s(0.2); o.set_fft_size(n); o.set_feed_percent(2)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    assert W == o.memsize_blocks == 38
    x = oracle.synth_audio(C, 25 * n, seed=13)          # 100 frames > 38 columns
    s.processBlocks(x[:, :3 * n])                        # 12 columns first, so the ring position is not 0
    s.processBlocks(x[:, 3 * n:])                        # 88 frames in one launch
    for b in range(25):
        o.process_synchron_block(x[:, b * n:(b + 1) * n])
    mem = np.zeros((W, H), np.float32)
    nv, pos = s.getMem(mem)
    assert pos == o.mem_counter == 100 % W
    assert np.abs(mem.astype(np.float64) - o.mem.astype(np.float64)).max() < 2e-3
    s.close()


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("frames", [1, 3, 17])
def test_ragged_frame_counts(jsg, oracle, torch_cuda, n, frames):
    """Frame counts that do not fill a workgroup (tail lanes duplicate the last frame and must not store)."""
    torch = torch_cuda
    hop = n // 4
    x = oracle.synth_audio(2, (frames - 1) * hop + n, seed=frames, kind="noise")
    win = oracle.window(oracle.WIN_HAMMING, n)
    plan = jsg.Plan(n, win)
    H = n // 2 + 1
    d_out = torch.full((frames + 2, H + 3), 7.0, device="cuda")          # two guard columns, three guard bins
    jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, frames, d_out[:frames + 0], feedblocks=4)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert (got[frames:] == 7.0).all() and (got[:, H:] == 7.0).all(), "stored outside the requested columns"
    idx = (np.arange(frames) * hop)[:, None] + np.arange(n)[None, :]
    fr = (x[:, idx] * win[None, None, :]).astype(np.float32)
    pw = oracle.mix_channels(oracle.power_spectrum_f64(fr).astype(np.float32), oracle.MIX_ABSMEAN)
    assert_db_close(got[:frames, :H], oracle.to_db(pw), pw.astype(np.float64), f"n={n} frames={frames}")


@pytest.mark.parametrize("n,channels", [(2048, 4), (512, 2), (512, 1), (4096, 5)])
@pytest.mark.parametrize("frames", [1, 2, 5, 16, 33])
def test_ragged_frame_counts_two_frames_per_wavefront(jsg, oracle, torch_cuda, n, channels, frames):
    """The 32-lane plans (512 points; 2048 points with >= 2 channels) put two frames into one wavefront and trade
    register halves before the column store: odd frame counts (the upper frame of the last wavefront is a duplicate of
    the lower one) and a ring whose wrap falls between the two frames of a wavefront."""
    torch = torch_cuda
    hop = n // 4
    x = oracle.synth_audio(channels, (frames - 1) * hop + n, seed=frames + n, kind="noise")
    win = oracle.window(oracle.WIN_HANN, n)
    plan = jsg.Plan(n, win)
    H = n // 2 + 1
    W = frames + 2
    pos = W - 1                                                      # frame 0 -> last column, frame 1 -> column 0
    d_out = torch.full((W, H + 5), 7.0, device="cuda")
    # plan_select=2: the "B" kernels of 2048 / 4096 points (automatic selection keeps them for launches that fill the GPU)
    jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, frames, d_out, feedblocks=4, ring_pos=pos, plan_select=2)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    cols = (pos + np.arange(frames)) % W
    untouched = np.setdiff1d(np.arange(W), cols)
    assert (got[untouched] == 7.0).all() and (got[:, H:] == 7.0).all(), "stored outside the requested columns"
    idx = (np.arange(frames) * hop)[:, None] + np.arange(n)[None, :]
    fr = (x[:, idx] * win[None, None, :]).astype(np.float32)
    pw = oracle.mix_channels(oracle.power_spectrum_f64(fr).astype(np.float32), oracle.MIX_ABSMEAN)
    assert_db_close(got[cols, :H], oracle.to_db(pw), pw.astype(np.float64), f"n={n} C={channels} frames={frames}")


@pytest.mark.parametrize("n", [2048, 4096])
def test_2048_and_4096_point_plans_agree(jsg, oracle, torch_cuda, n):
    """2048 and 4096 points have two kernels each: the small-workgroup plan and the "B" plan (two-stage / one wavefront per
    frame with factorised tables; picked automatically for launches that fill the GPU -- at 2048 points from two channels per column on).
    Linear power of both, for every channel count, against the float64 DFT; and the same stream through both (channels
    duplicated so that the mixes are equal) stays within the float32 bound of one another."""
    torch = torch_cuda
    hop, K = 512, 10
    x = oracle.synth_audio(4, K * n, seed=11, kind="mix")
    win = oracle.window(oracle.WIN_BLACKMANHARRIS, n)
    plan = jsg.Plan(n, win)
    F = (K - 1) * 4 + 1
    H = n // 2 + 1
    idx = (np.arange(F) * hop)[:, None] + np.arange(n)[None, :]
    fr = (x[:, idx] * win[None, None, :]).astype(np.float32)
    p64 = oracle.power_spectrum_f64(fr)                              # [C][F][H]
    for C in (1, 2, 3, 4):
        for sel in (1, 2):
            d_out = torch.empty((F, H), device="cuda")
            jsg.stft_db(plan, torch.from_numpy(x[:C].copy()).cuda(), hop, F, d_out, mix_mode=jsg.capi.MIX_SUM, linear_out=True,
                        plan_select=sel)
            torch.cuda.synchronize()
            worst = assert_power_close(d_out.cpu().numpy(), p64[:C].sum(axis=0), f"{n} points, {C} channels summed, kernel {sel}")
            assert worst < 1e-5, (C, sel, worst)
    # one channel four times: the AbsMean of four equal channels is the channel itself up to one rounding of the sum
    one = torch.from_numpy(x[:1].copy()).cuda()
    d1 = torch.empty((F, H), device="cuda"); d4 = torch.empty((F, H), device="cuda")
    jsg.stft_db(plan, one, hop, F, d1, linear_out=True, plan_select=1)
    jsg.stft_db(plan, one.repeat(4, 1).contiguous(), hop, F, d4, linear_out=True, plan_select=2)
    torch.cuda.synchronize()
    a, b = d1.cpu().numpy().astype(np.float64), d4.cpu().numpy().astype(np.float64)
    peak = p64[0].max(axis=-1, keepdims=True)
    assert (np.abs(a - b) <= 2 * (1e-5 * p64[0] + 1e-6 * peak)).all()


def _random_geometries(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.choice([512, 1024, 2048, 2048, 4096, 8192]))
        channels = int(rng.integers(1, 10))
        mix = int(rng.choice([0, 0, 1, 2, 3, 4]))
        if mix == 4 and channels < 2:
            mix = 3
        feedblocks = int(rng.choice([1, 2, 4, 8, 10]))
        hop = n // feedblocks if feedblocks != 10 else [51, 102, 205, 410, 819][[512, 1024, 2048, 4096, 8192].index(n)]
        frames = int(rng.integers(1, 70))
        extra = int(rng.integers(0, 4))
        sel = int(rng.choice([0, 1, 2])) if n in (1024, 2048, 4096) else 0      # which of the two 1024- (round 6) / 2048- / 4096-point kernels
        out.append((n, channels, mix, feedblocks, hop, frames, extra, sel))
    return out


# JSG_FUZZ_CASES / JSG_FUZZ_SEED widen the sweep for a one-off campaign (tools/README.md); the suite runs 48 fixed cases
@pytest.mark.parametrize("n,channels,mix,feedblocks,hop,frames,extra,sel",
                         _random_geometries(int(os.environ.get("JSG_FUZZ_CASES", "48")), int(os.environ.get("JSG_FUZZ_SEED", "20260"))))
def test_seeded_random_geometries(jsg, oracle, torch_cuda, n, channels, mix, feedblocks, hop, frames, extra, sel):
    """Seeded sweep over plan x channel count x mix x hop pattern x frame count x ring position (every kernel
    instantiation, both 2048-point plans, the perc10 pattern of Spectrogram.cpp:50-55,216) against the float64 DFT."""
    torch = torch_cuda
    starts = np.array([(j // feedblocks) * n + (j % feedblocks) * hop for j in range(frames)])
    x = oracle.synth_audio(channels, int(starts[-1]) + n, seed=n + frames, kind="noise" if frames % 2 else "mix")
    win = oracle.window((frames + channels) % 6, n)
    plan = jsg.Plan(n, win)
    H = n // 2 + 1
    W = frames + extra
    pos = (W - 1) if extra else 0
    d_out = torch.full((W, H + 2), 7.0, device="cuda")
    jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, frames, d_out, feedblocks=feedblocks, mix_mode=mix, ring_pos=pos,
                plan_select=sel)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    cols = (pos + np.arange(frames)) % W
    assert (got[np.setdiff1d(np.arange(W), cols)] == 7.0).all() and (got[:, H:] == 7.0).all(), "stored outside the requested columns"
    fr = (x[:, starts[:, None] + np.arange(n)[None, :]] * win[None, None, :]).astype(np.float32)
    p64 = oracle.power_spectrum_f64(fr)                                   # [C][F][H]
    pw = oracle.mix_channels(p64.astype(np.float32), mix)
    peak = p64.max(axis=(0, 2))[:, None]                                  # largest per-channel peak of every frame
    assert_db_close(got[cols, :H], oracle.to_db(pw), pw.astype(np.float64),
                    f"n={n} C={channels} mix={mix} fb={feedblocks} hop={hop} F={frames}", peak=peak)
    # round 5, on the same case: the tail-plane layout holds the very same values (every second case), and where the pair plan applies
    # (2048 points, AbsMean over an even channel count) its columns hold the bound too
    if (frames + channels) % 2 == 0:
        d_dense = torch.full((W, n // 2), 7.0, device="cuda")
        d_tail = torch.full((1, W), 7.0, device="cuda")
        jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, frames, d_dense, feedblocks=feedblocks, mix_mode=mix, ring_pos=pos,
                    plan_select=sel, d_tail=d_tail)
        torch.cuda.synchronize()
        assert torch.equal(d_dense, d_out[:, :n // 2]) and torch.equal(d_tail[0, cols], d_out[cols, n // 2]), "tail-plane layout differs"
        assert bool((d_tail[0, np.setdiff1d(np.arange(W), cols)] == 7.0).all())
    if n == 2048 and mix == 0 and channels % 2 == 0:
        d_pair = torch.full((W, H + 2), 7.0, device="cuda")
        jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, frames, d_pair, feedblocks=feedblocks, mix_mode=mix, ring_pos=pos, plan_select=3)
        torch.cuda.synchronize()
        gp = d_pair.cpu().numpy()
        assert (gp[np.setdiff1d(np.arange(W), cols)] == 7.0).all() and (gp[:, H:] == 7.0).all()
        assert_db_close(gp[cols, :H], oracle.to_db(pw), pw.astype(np.float64), f"pair plan C={channels} fb={feedblocks} hop={hop} F={frames}", peak=peak)


def test_empty_inputs_and_bad_geometry(jsg, oracle, torch_cuda):
    torch = torch_cuda
    n = 1024
    plan = jsg.Plan(n, oracle.window(1, n))
    d_in = torch.zeros((1, 4 * n), device="cuda")
    d_out = torch.full((4, 544), 3.0, device="cuda")
    jsg.stft_db(plan, d_in, 512, 0, d_out)                                # zero frames: a no-op, not an error
    torch.cuda.synchronize()
    assert bool((d_out == 3.0).all())
    with pytest.raises(jsg.JsgError):                                      # more frames than ring columns would race
        jsg.stft_db(plan, d_in, 512, 5, d_out)
    with pytest.raises(jsg.JsgError):                                      # column pitch smaller than n/2+1
        jsg.stft_db(plan, d_in, 512, 2, torch.zeros((4, 512), device="cuda"))
    # a frame count that reaches past the end of the input rows is refused, not read (4n samples hold
    # 7 frames at hop n/2; the 8th would read n/2 floats past the allocation)
    big = torch.zeros((16, 544), device="cuda")
    jsg.stft_db(plan, d_in, 512, 7, big)
    with pytest.raises(jsg.JsgError, match="past the end"):
        jsg.stft_db(plan, d_in, 512, 8, big)
    with pytest.raises(jsg.JsgError, match="past the end"):
        jsg.stft_db(plan, d_in, 512, 4, big, first_frame=4)
    with pytest.raises(jsg.JsgError, match="past the end"):               # irregular (perc10-style) framing: 10 frames per block
        jsg.stft_db(plan, d_in, 102, 16, torch.zeros((64, 544), device="cuda"), feedblocks=10, first_frame=30)
    with pytest.raises(jsg.JsgError):                                      # per-channel output must have one plane per channel
        jsg.stft_db(plan, torch.zeros((2, 4 * n), device="cuda"), 512, 2, big, mix_mode=jsg.capi.MIX_PER_CHANNEL)
    torch.cuda.synchronize()
    s = jsg.Spectrogram(1)
    s.setFFTSize(n)
    assert s.processBlocks(np.zeros((1, 0), np.float32)) == 0             # empty batch
    mem = np.zeros((s.getMemorySize(), 513), np.float32)
    nv, pos = s.getMem(mem)
    assert pos == 0 and (mem == np.float32(-120.0)).all()
    with pytest.raises(jsg.JsgError):
        s.processSynchronBlock(np.zeros((1, 1000), np.float32))           # not an fft-size block
    s.close()


def test_producer_and_consumer_threads(jsg, oracle):
    """Audio thread (processSynchronBlock) and GUI thread (getMem) use one engine concurrently.  getMem copies only
    the new columns into the caller's buffer, so after the run that buffer must hold the complete final ring (the
    reference races here; the engine serialises producer and consumer internally)."""
    import threading
    C, n, K = 2, 1024, 60
    x = oracle.synth_audio(C, K * n, seed=8)
    s = jsg.Spectrogram(C)
    s.setSamplerate(48000.0); s.setmemoryTime_s(0.5); s.setFFTSize(n); s.setfeed_percent(1)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    assert 2 * K > W
    mem = np.zeros((W, H), np.float32)
    counts, stop = [], threading.Event()

    def consumer():
        while not stop.is_set():
            nv, _ = s.getMem(mem)
            counts.append(nv)

    t = threading.Thread(target=consumer)
    t.start()
    for b in range(K):
        assert s.processSynchronBlock(x[:, b * n:(b + 1) * n]) == 0
    stop.set(); t.join()
    nv, pos = s.getMem(mem)
    counts.append(nv)
    assert pos == (2 * K) % W
    assert sum(c for c in counts if c < oracle.NEW_ENTRY_SENTINEL) <= 2 * K      # no column reported twice
    ref = jsg.Spectrogram(C)
    ref.setSamplerate(48000.0); ref.setmemoryTime_s(0.5); ref.setFFTSize(n); ref.setfeed_percent(1)
    ref.processBlocks(x)
    full = np.zeros((W, H), np.float32)
    ref.getMem(full)
    assert (mem.view(np.uint32) == full.view(np.uint32)).all()
    s.close(); ref.close()


def test_integration_md_ctypes_snippet(jsg):
    """The raw-ctypes example of INTEGRATION.md section D, verbatim in spirit: plain C-ABI, no package helpers."""
    import ctypes as C
    lib = C.CDLL(jsg.capi.LIB_PATH)
    h = C.c_void_p()
    assert lib.jsg_create(C.byref(h), 2) == 0
    assert lib.jsg_set_samplerate(h, C.c_float(48000.0)) == 0
    assert lib.jsg_set_fft_size(h, 2048) == 0 and lib.jsg_set_feed_percent(h, 1) == 0
    block = np.zeros((2, 2048), np.float32)
    ptrs = (C.c_void_p * 2)(block[0].ctypes.data, block[1].ctypes.data)
    assert lib.jsg_process_block(h, ptrs) == 0
    W, H = lib.jsg_get_memory_size(h), lib.jsg_get_spectrum_size(h)
    assert (W, H) == (47, 1025)
    mem = np.zeros((W, H), np.float32); pos = C.c_int()
    new_vals = lib.jsg_get_mem(h, mem.ctypes.data_as(C.c_void_p), W, C.byref(pos))
    assert new_vals > W and pos.value == 2 and (mem[:2] == np.float32(-110.0)).all()
    assert lib.jsg_get_mem(h, mem.ctypes.data_as(C.c_void_p), W + 1, C.byref(pos)) == -1
    assert lib.jsg_destroy(h) == 0


def test_launches_are_graph_capturable(jsg, oracle, torch_cuda):
    """The launch entry points only enqueue (no malloc / sync inside): a hipGraph holding STFT + colour-loop launches
    replays to the same bits as eager execution."""
    torch = torch_cuda
    n, hop, F = 1024, 512, 64
    win = oracle.window(oracle.WIN_HANN, n)
    plan = jsg.Plan(n, win)
    x = torch.from_numpy(oracle.synth_audio(2, F * hop + n, seed=6)).cuda()
    lut = torch.from_numpy(jsg.colormap_lut(256, 6)).cuda()
    ring_e = torch.zeros((F, 544), device="cuda"); img_e = torch.zeros((513, F), dtype=torch.int32, device="cuda")
    ring_g = torch.zeros((F, 544), device="cuda"); img_g = torch.zeros((513, F), dtype=torch.int32, device="cuda")
    jsg.stft_db(plan, x, hop, F, ring_e)                       # eager (also the warm-up that sets kernel attributes)
    jsg.colormap(ring_e, lut, -50.0, 50.0, d_argb=img_e, height=513)
    torch.cuda.synchronize()
    s2 = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            jsg.stft_db(plan, x, hop, F, ring_g, stream=s2.cuda_stream)
            jsg.colormap(ring_g, lut, -50.0, 50.0, d_argb=img_g, height=513, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert float(ring_g.abs().sum()) == 0.0                    # capture did not execute anything
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ring_g[:, :513], ring_e[:, :513]) and torch.equal(img_g, img_e)
    # a plan whose kernels need > 48 KB of dynamic LDS (2048 points, both plans): capture FIRST, run eagerly afterwards --
    # the LDS attribute is set when the plan is created, not inside the first (captured) launch
    n2 = 2048
    plan2 = jsg.Plan(n2, oracle.window(oracle.WIN_HANN, n2))
    for C in (2, 6):
        x2 = torch.from_numpy(oracle.synth_audio(C, F * hop + n2, seed=8)).cuda()
        ring_c = torch.zeros((F, 1056), device="cuda"); ring_x = torch.zeros((F, 1056), device="cuda")
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s2):
            with torch.cuda.graph(g2, stream=s2):
                jsg.stft_db(plan2, x2, hop, F, ring_c, mix_mode=jsg.capi.MIX_MIN, stream=s2.cuda_stream)
        g2.replay()
        jsg.stft_db(plan2, x2, hop, F, ring_x, mix_mode=jsg.capi.MIX_MIN)
        torch.cuda.synchronize()
        assert torch.equal(ring_c, ring_x)


@pytest.mark.parametrize("scale", [32768.0, 1e-6])
def test_amplitude_extremes(jsg, oracle, scale):
    """PCM-scale and very quiet signals: no overflow, and the 1e-11 floor dominates exactly like in the reference."""
    n = 1024
    x = (oracle.synth_audio(1, 4 * n, seed=19) * np.float32(scale)).astype(np.float32)
    s = jsg.Spectrogram(1)
    s.setSamplerate(48000.0); s.setFFTSize(n); s.setfeed_percent(1)
    s.processBlocks(x)
    mem = np.zeros((s.getMemorySize(), 513), np.float32); s.getMem(mem)
    win = oracle.window(1, n)
    ref = oracle.stft_db_reference(x, n, 512, 2, win)
    pw = mixed_power_f64(oracle, x, n, 512, 2, win, 0)
    assert_db_close(mem[:8], ref, pw, f"scale {scale}")
    assert np.isfinite(mem[:8]).all()
    s.close()


def test_launch_many_on_several_streams_equals_single_launches(jsg, oracle, torch_cuda):
    """jsg_stft_db_launch_many: independent batches issued from one call over several HIP streams give the same bits
    as one launch at a time."""
    import ctypes as C
    from jadespectrogram_amd import capi
    from jadespectrogram_amd.spectrogram import _stft_args
    torch = torch_cuda
    n, hop, F, B = 1024, 512, 96, 6
    plan = jsg.Plan(n, oracle.window(1, n))
    xs = [torch.from_numpy(oracle.synth_audio(1, F * hop + n, seed=40 + b)).cuda() for b in range(B)]
    ref = [torch.zeros((F, 544), device="cuda") for _ in range(B)]
    out = [torch.zeros((F, 544), device="cuda") for _ in range(B)]
    for b in range(B):
        jsg.stft_db(plan, xs[b], hop, F, ref[b])
    torch.cuda.synchronize()
    arr = (capi.StftArgs * B)()
    for b in range(B):
        a = _stft_args(plan, xs[b], hop, F, out[b])
        C.memmove(C.byref(arr, b * C.sizeof(capi.StftArgs)), C.byref(a), C.sizeof(capi.StftArgs))
    streams = [torch.cuda.Stream() for _ in range(3)]
    sarr = (C.c_void_p * 3)(*[s.cuda_stream for s in streams])
    capi.check(capi.lib().jsg_stft_db_launch_many(plan._p, arr, B, sarr, 3))
    torch.cuda.synchronize()
    for b in range(B):
        assert torch.equal(out[b][:, :513], ref[b][:, :513])
    assert capi.lib().jsg_stft_db_launch_many(plan._p, arr, 0, None, 0) == 0          # empty list is a no-op
    assert capi.lib().jsg_stft_db_launch_many(plan._p, arr, 2, None, 0) == 0          # default stream
    torch.cuda.synchronize()
