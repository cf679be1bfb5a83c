"""The bit-exact contract (missing item 1): oracle/jsg_mirror.c restates the GPU kernel's float32 arithmetic operation by
operation, so
  * the LINEAR power of every plan equals the mirror's in every bit (window multiply, FFT in the kernel's factorisation and operation
    order, paired post pass, |X|^2, channel mix),
  * with exact_log the dB columns equal the mirror's in every bit (the logarithm is the shared float32 routine of
    csrc/jsg_exact_math.h instead of the hardware unit), and therefore
  * the palette indices / ARGB pixels of the colour loop equal those computed on the CPU from the mirror's columns: 0 flips end to end.
The float64 oracle stays the accuracy yardstick (tests/test_oracle_golden.py pins the mirror to it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


@pytest.fixture(scope="module")
def mirror():
    from oracle import mirror as m
    return m.load()


def _signal(oracle, C, n_samples, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "synth":
        return oracle.synth_audio(C, n_samples, seed=seed)
    if kind == "noise":
        return rng.uniform(-1, 1, (C, n_samples)).astype(np.float32)
    if kind == "quiet":     # tiny values: products and sums that underflow towards denormals must round the same way on both sides
        return (rng.uniform(-1, 1, (C, n_samples)) * 1e-18).astype(np.float32)
    x = np.zeros((C, n_samples), np.float32)   # "sparse": silence with a few full-scale clicks (exact zeros through the butterflies)
    x[:, rng.integers(0, n_samples, 7)] = 1.0
    return x


CASES = [  # (n, plan_select, kernel name, channels, hop, feedblocks, mix, window, signal)
    (512, 0, "Cfg512", 1, 256, 2, "absmean", 1, "synth"), (512, 0, "Cfg512", 3, 128, 4, "absmean", 3, "noise"),
    (1024, 0, "Cfg1024", 1, 512, 2, "absmean", 1, "synth"), (1024, 0, "Cfg1024", 2, 512, 2, "absmean", 1, "noise"),
    (1024, 0, "Cfg1024", 3, 256, 4, "absmean", 2, "synth"), (1024, 0, "Cfg1024", 5, 102, 10, "max", 4, "noise"),
    (1024, 0, "Cfg1024", 4, 1024, 1, "min", 0, "synth"), (1024, 0, "Cfg1024", 2, 512, 2, "right", 5, "sparse"),
    (1024, 0, "Cfg1024", 8, 512, 2, "sum", 1, "quiet"),
    # round 6: the two-stage 1024-point plan (split-radix 16 x 32, 16 lanes per frame, four frames per wavefront; plan_select = 2)
    (1024, 2, "Cfg1024B", 1, 512, 2, "absmean", 1, "synth"), (1024, 2, "Cfg1024B", 2, 512, 2, "absmean", 1, "noise"),
    (1024, 2, "Cfg1024B", 3, 256, 4, "absmean", 2, "synth"), (1024, 2, "Cfg1024B", 8, 512, 2, "sum", 1, "quiet"),
    (1024, 2, "Cfg1024B", 2, 512, 2, "right", 5, "sparse"), (1024, 2, "Cfg1024B", 5, 102, 10, "absmean", 4, "noise"),
    (2048, 1, "Cfg2048", 1, 1024, 2, "absmean", 1, "synth"), (2048, 1, "Cfg2048", 8, 512, 4, "absmean", 1, "noise"),
    (2048, 2, "Cfg2048B", 8, 512, 4, "absmean", 1, "synth"), (2048, 2, "Cfg2048B", 1, 1024, 2, "absmean", 3, "sparse"), (2048, 2, "Cfg2048B", 3, 205, 1, "absmean", 2, "noise"),
    (4096, 1, "Cfg4096", 2, 512, 8, "absmean", 1, "synth"), (4096, 1, "Cfg4096", 1, 2048, 2, "left", 4, "noise"),
    (4096, 2, "Cfg4096B", 2, 512, 8, "absmean", 1, "synth"), (4096, 2, "Cfg4096B", 6, 1024, 4, "absmean", 5, "quiet"),
    (8192, 0, "Cfg8192", 1, 4096, 2, "absmean", 1, "synth"), (8192, 0, "Cfg8192", 2, 2048, 4, "max", 2, "noise"),
]


@pytest.mark.parametrize("n,sel,kernel,C,hop,fb,mix,win_kind,signal", CASES)
def test_linear_power_and_exact_db_equal_the_mirror_bit_for_bit(jsg, oracle, mirror, torch_cuda, n, sel, kernel, C, hop, fb, mix, win_kind, signal):
    torch = torch_cuda
    cap = jsg.capi
    m = {"absmean": cap.MIX_ABSMEAN, "max": cap.MIX_MAX, "min": cap.MIX_MIN, "left": cap.MIX_LEFT, "right": cap.MIX_RIGHT, "sum": cap.MIX_SUM}[mix]
    F = 48 if n <= 2048 else 24
    last = F - 1
    n_samples = (last // fb) * n + (last % fb) * hop + n
    x = _signal(oracle, C, n_samples, signal, seed=n + C + hop)
    win = oracle.window(win_kind, n)
    plan = jsg.Plan(n, win)
    H, pitch = n // 2 + 1, (n // 2 + 1 + 31) // 32 * 32
    d_x = torch.from_numpy(x).cuda()
    d_lin = torch.zeros((F, pitch), device="cuda")
    d_db = torch.zeros((F, pitch), device="cuda")
    kw = dict(feedblocks=fb, mix_mode=m, plan_select=sel)
    assert jsg.stft_kernel_name(plan, d_x, hop, F, d_lin, **kw) == kernel
    jsg.stft_db(plan, d_x, hop, F, d_lin, linear_out=True, **kw)
    jsg.stft_db(plan, d_x, hop, F, d_db, exact_log=True, **kw)
    torch.cuda.synchronize()
    ref_lin = mirror.columns(kernel, x, hop, F, win, feedblocks=fb, mix=m)
    ref_db = mirror.columns(kernel, x, hop, F, win, feedblocks=fb, mix=m, exact_db=True)
    got_lin = d_lin[:, :H].cpu().numpy()
    got_db = d_db[:, :H].cpu().numpy()
    bad = got_lin.view(np.uint32) != ref_lin.view(np.uint32)
    assert not bad.any(), (f"{kernel}: {int(bad.sum())} of {bad.size} power values differ from the mirror; first at {tuple(np.argwhere(bad)[0])}: "
                           f"{got_lin[bad][0]!r} vs {ref_lin[bad][0]!r}")
    bad = got_db.view(np.uint32) != ref_db.view(np.uint32)
    assert not bad.any(), f"{kernel}: {int(bad.sum())} of {bad.size} exact-log dB values differ from the mirror"
    # (the mirror itself is held to the float64 oracle in tests/test_oracle_golden.py)


def test_power_scale_and_first_frame(jsg, oracle, mirror, torch_cuda):
    torch = torch_cuda
    n, hop, F, C = 1024, 512, 40, 2
    x = oracle.synth_audio(C, (F + 9) * hop + n, seed=5)
    win = oracle.window(oracle.WIN_HAMMING, n)
    plan = jsg.Plan(n, win, power_scale=1.0 / 1024.0)
    d = torch.zeros((F, 544), device="cuda")
    jsg.stft_db(plan, torch.from_numpy(x).cuda(), hop, F, d, first_frame=7, linear_out=True)
    torch.cuda.synchronize()
    ref = mirror.columns("Cfg1024", x, hop, F, win, power_scale=1.0 / 1024.0, first_frame=7)
    assert (d[:, :513].cpu().numpy().view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("n,C,hop,scheme", [(1024, 1, 512, 6), (1024, 2, 512, 4), (4096, 2, 512, 6), (2048, 8, 512, 2)])
def test_exact_log_gives_zero_index_flips_end_to_end(jsg, oracle, mirror, torch_cuda, n, C, hop, scheme):
    """GPU power -> exact dB -> colour loop on the GPU against mirror power -> the same logarithm -> CColorPalette on the CPU: the
    palette indices and the ARGB pixels are identical (the default hardware logarithm flips a handful per million)."""
    torch = torch_cuda
    F = 1024 if n <= 2048 else 512
    x = oracle.synth_audio(C, (F - 1) * hop + n, seed=n + scheme)
    win = oracle.window(oracle.WIN_HANN, n)
    plan = jsg.Plan(n, win)
    H, pitch = n // 2 + 1, (n // 2 + 1 + 31) // 32 * 32
    d_x = torch.from_numpy(x).cuda()
    d_db = torch.zeros((F, pitch), device="cuda")
    kernel = jsg.stft_kernel_name(plan, d_x, hop, F, d_db, feedblocks=n // hop)
    jsg.stft_db(plan, d_x, hop, F, d_db, feedblocks=n // hop, exact_log=True)
    d_lut = torch.from_numpy(jsg.colormap_lut(256, scheme)).cuda()
    d_img = torch.zeros((H, F), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((H, F), dtype=torch.uint8, device="cuda")
    jsg.colormap(d_db, d_lut, -50.0, 50.0, d_argb=d_img, d_index=d_idx, n_cols=F, height=H)
    torch.cuda.synchronize()
    ref_db = mirror.columns(kernel, x, hop, F, win, feedblocks=n // hop, exact_db=True)
    pal = oracle.OracleColorPalette(256, scheme)
    pal.set_value_range(-50.0, 50.0)
    ref_idx = pal.index(ref_db).astype(np.uint8)                       # [column][bin]
    got_idx = d_idx.cpu().numpy()[::-1, :].T                           # image rows are flipped: y = H - 1 - bin
    assert int((got_idx != ref_idx).sum()) == 0
    ring = np.ascontiguousarray(ref_db)
    assert (d_img.cpu().numpy().view(np.uint32) == oracle.render_all(ring, 0, pal, running=True)[:, :F]).all()


def test_engine_with_exact_log_equals_the_mirror(jsg, oracle, mirror, torch_cuda):
    """class Spectrogram with setExactLog(True): the ring after block-by-block processing (N-zero pre-roll, -110 dB first column) is the
    mirror's, bit for bit, and so is the image of the display tick."""
    n, C, K = 1024, 2, 12
    x = oracle.synth_audio(C, K * n, seed=77)
    s = jsg.Spectrogram(C)
    s.setSamplerate(48000.0); s.setFFTSize(n); s.setfeed_percent(jsg.Spectrogram.FeedPercentage.perc50)
    s.setExactLog(True)
    for k in range(K):
        s.processSynchronBlock([x[c, k * n:(k + 1) * n] for c in range(C)])
    W, H = s.getMemorySize(), s.getSpectrumSize()
    mem = np.zeros((W, H), np.float32)
    nv, pos = s.getMem(mem)          # (a fresh engine reports the reference's "copy everything" sentinel + the new columns, Spectrogram.cpp:18)
    assert nv >= 2 * K and pos == 2 * K
    xz = np.concatenate([np.zeros((C, n), np.float32), x], axis=1)     # the engine's zero pre-roll
    ref = mirror.columns("Cfg1024", xz, 512, 2 * K, oracle.window(oracle.WIN_HANN, n), exact_db=True)
    assert (mem[:2 * K].view(np.uint32) == ref.view(np.uint32)).all()
    assert np.all(mem[0] == np.float32(-110.0)) or abs(float(mem[0, 0]) + 110.0) < 1e-4    # 10 log10(1e-11f)
    d = jsg.SpectrogramDisplay(s)
    img = np.zeros((H, W), np.uint32)
    d.timerCallback(img)
    ring = np.full((W, H), -120.0, np.float32)
    ring[:2 * K] = ref
    pal = oracle.OracleColorPalette(256, oracle.CM_JADE)
    pal.set_value_range(-50.0, 50.0)
    assert int((img != oracle.render_all(ring, pos, pal, running=True)).sum()) == 0
    s.close()


def test_strided_and_per_channel_exact_log(jsg, oracle, mirror, torch_cuda):
    torch = torch_cuda
    n, hop, F, K, C = 1024, 512, 200, 5, 3
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    xs = [oracle.synth_audio(C, (F - 1) * hop + n, seed=900 + b) for b in range(K)]
    d_in = torch.from_numpy(np.stack(xs)).cuda()
    d_out = torch.full((K, C, 256, 544), -7.0, device="cuda")
    jsg.stft_db_strided(plan, d_in, hop, F, d_out, mix_mode=jsg.capi.MIX_PER_CHANNEL, ring_pos=100, exact_log=True)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    cols = (100 + np.arange(F)) % 256
    for b in range(K):
        for c in range(C):
            ref = mirror.columns("Cfg1024", xs[b][c:c + 1], hop, F, oracle.window(oracle.WIN_HANN, n), mix=jsg.capi.MIX_SUM, exact_db=True)
            assert (got[b, c, cols, :513].view(np.uint32) == ref.view(np.uint32)).all(), (b, c)
    untouched = np.setdiff1d(np.arange(256), cols)
    assert (got[:, :, untouched, :] == -7.0).all() and (got[..., 513:] == -7.0).all()
