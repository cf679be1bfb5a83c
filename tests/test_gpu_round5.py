"""Round 5 additions, each against what it must equal bit for bit:
  * tail plane (jsg_stft_args.out_tail): bin n/2 of every column in a dense plane, columns of n/2 floats -- the values of the reference
    layout (m_mem[col][bin], Spectrogram.h:144), every plan, strided batches, per-channel rows, ring wrap;
  * exact_log in the display launches (jsg_stft_image_launch(_strided), both forms): 0 palette-index flips against the CPU mirror through
    the ONE-kernel path on the C5 geometry (reference Spectrogram.cpp:107 feeding CColorpalette.h:32-47);
  * the lossless producer call (jsg_process_block_wait): more blocks than the queue has slots, back to back, none lost."""
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


def _rand_in(torch, shape, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.empty(shape, device="cuda")
    x.uniform_(-1.0, 1.0, generator=g)
    t = torch.arange(shape[-1], device="cuda", dtype=torch.float64)
    return (0.3 * x.double() + 0.5 * torch.sin(2 * np.pi * 1234.5 * t / 48000.0)).float()


TAIL_CASES = [  # (n, plan_select, channels, hop, frames, ring width, ring_pos, mix, linear, exact)
    (512, 0, 1, 256, 333, 400, 390, "absmean", False, False), (512, 0, 2, 128, 64, 64, 0, "absmean", True, False),
    (1024, 0, 1, 512, 4096, 4096, 0, "absmean", False, False), (1024, 0, 1, 512, 1000, 1200, 1100, "absmean", False, True),
    (1024, 0, 3, 256, 77, 128, 120, "max", False, False), (1024, 0, 4, 512, 50, 64, 60, "per_channel", False, False),
    (2048, 1, 2, 512, 300, 512, 500, "absmean", False, False), (2048, 2, 8, 512, 4096, 4096, 7, "absmean", False, False),
    (2048, 2, 3, 1024, 130, 256, 200, "absmean", True, False),
    (4096, 1, 2, 512, 100, 128, 100, "absmean", False, False), (4096, 2, 2, 512, 1875, 1875, 1800, "absmean", False, True),
    (8192, 0, 1, 2048, 40, 64, 50, "left", False, False),
]


@pytest.mark.parametrize("n,sel,C,hop,F,W,pos,mix,linear,exact", TAIL_CASES)
def test_tail_plane_holds_the_values_of_the_reference_layout(jsg, oracle, torch_cuda, n, sel, C, hop, F, W, pos, mix, linear, exact):
    torch = torch_cuda
    cap = jsg.capi
    m = {"absmean": cap.MIX_ABSMEAN, "max": cap.MIX_MAX, "left": cap.MIX_LEFT, "per_channel": cap.MIX_PER_CHANNEL}[mix]
    M, H = n // 2, n // 2 + 1
    pitch = (H + 31) // 32 * 32
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_x = _rand_in(torch, (C, (F - 1) * hop + n), seed=n + F)
    planes = C if m == cap.MIX_PER_CHANNEL else 1
    shape_ref = (C, W, pitch) if planes > 1 else (W, pitch)
    shape_new = (C, W, M) if planes > 1 else (W, M)               # a column is exactly n/2 floats: whole 128-byte lines
    ref = torch.full(shape_ref, -7.0, device="cuda")
    got = torch.full(shape_new, -7.0, device="cuda")
    tail = torch.full((planes, W), -7.0, device="cuda")
    kw = dict(feedblocks=max(1, n // hop), mix_mode=m, ring_pos=pos, linear_out=linear, plan_select=sel, exact_log=exact)
    jsg.stft_db(plan, d_x, hop, F, ref, **kw)
    jsg.stft_db(plan, d_x, hop, F, got, d_tail=tail, **kw)
    torch.cuda.synchronize()
    cols = (pos + torch.arange(F, device="cuda")) % W
    assert torch.equal(got.view(planes, W, M), ref.view(planes, W, pitch)[..., :M]), "bins 0 .. n/2-1 differ (or a column outside the launch was written)"
    assert torch.equal(tail[:, cols], ref.view(planes, W, pitch)[:, cols, M]), "bin n/2 differs"
    untouched = torch.ones(W, dtype=torch.bool, device="cuda"); untouched[cols] = False
    assert (tail[:, untouched] == -7.0).all(), "tail entries of columns outside the launch were written"


@pytest.mark.parametrize("n,sel,C,K,F,mix", [(1024, 0, 1, 9, 700, "absmean"), (1024, 0, 3, 4, 130, "per_channel"), (2048, 0, 8, 2, 4096, "absmean"),
                                             (4096, 0, 2, 6, 1875, "absmean"), (512, 0, 2, 5, 100, "absmean")])
def test_tail_plane_in_strided_batches(jsg, oracle, torch_cuda, n, sel, C, K, F, mix):
    torch = torch_cuda
    cap = jsg.capi
    m = cap.MIX_PER_CHANNEL if mix == "per_channel" else cap.MIX_ABSMEAN
    hop, M, H = 512 if n >= 1024 else 256, n // 2, n // 2 + 1
    pitch = (H + 31) // 32 * 32
    W, pos = F + 5, 3
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_in = _rand_in(torch, (K, C, (F - 1) * hop + n), seed=n + K)
    planes = C if m == cap.MIX_PER_CHANNEL else 1
    ref = torch.full((K, C, W, pitch) if planes > 1 else (K, W, pitch), -7.0, device="cuda")
    got = torch.full((K, C, W, M) if planes > 1 else (K, W, M), -7.0, device="cuda")
    tail = torch.full((K, planes, W), -7.0, device="cuda")
    kw = dict(feedblocks=n // hop, mix_mode=m, ring_pos=pos, plan_select=sel)
    name = jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, ref, **kw)
    jsg.stft_db_strided(plan, d_in, hop, F, ref, **kw)
    jsg.stft_db_strided(plan, d_in, hop, F, got, d_tail=tail, **kw)
    torch.cuda.synchronize()
    r = ref.view(K, planes, W, pitch)
    assert torch.equal(got.view(K, planes, W, M), r[..., :M]), name
    cols = (pos + torch.arange(F, device="cuda")) % W
    assert torch.equal(tail[:, :, cols], r[:, :, cols, M]), name
    untouched = torch.ones(W, dtype=torch.bool, device="cuda"); untouched[cols] = False
    assert (tail[:, :, untouched] == -7.0).all()


def test_tail_plane_is_refused_where_it_does_not_apply(jsg, oracle, torch_cuda):
    torch = torch_cuda
    n, hop, F = 1024, 512, 16
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_x = _rand_in(torch, (1, (F - 1) * hop + n), seed=1)
    with pytest.raises(jsg.capi.JsgError):          # a 512-float column without a tail plane: bin n/2 would land in the next column
        jsg.stft_db(plan, d_x, hop, F, torch.zeros((F, 512), device="cuda"))


@pytest.mark.parametrize("n,C,hop,F,W,scheme,one_kernel", [
    (4096, 2, 512, 1875, 1875, 6, True),       # C5: stereo 96 kHz, 87.5 % overlap, Jade, a ten-second image -- ONE kernel (Cfg4096B, OUTK = 2)
    (1024, 1, 512, 600, 640, 4, True),         # 1024 points: one kernel (Cfg1024I)
    (4096, 2, 512, 300, 300, 6, False),        # a launch that does not fill the rounds: two kernels (Cfg4096, index scratch)
    (2048, 8, 512, 512, 512, 2, False),        # 2048 points: two kernels
])
def test_exact_log_in_the_display_launch_gives_zero_index_flips(jsg, oracle, mirror, torch_cuda, n, C, hop, F, W, scheme, one_kernel):
    """jsg_stft_image_launch with exact_log: GPU power -> shared float32 logarithm -> palette index -> ARGB inside the STFT kernel,
    against mirror power -> the same logarithm -> CColorPalette on the CPU.  Every pixel is identical."""
    torch = torch_cuda
    H = n // 2 + 1
    x = oracle.synth_audio(C, (F - 1) * hop + n, seed=n + scheme + F)
    win = oracle.window(oracle.WIN_HANN, n)
    plan = jsg.Plan(n, win)
    d_x = torch.from_numpy(x).cuda()
    d_lut = torch.from_numpy(jsg.colormap_lut(256, scheme)).cuda()
    d_img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    scratch = torch.zeros((W, (H + 63) // 64 * 64), dtype=torch.uint8, device="cuda")
    kw = dict(feedblocks=n // hop, ring_width=W, exact_log=True)
    needs = jsg.stft_image_needs_scratch(plan, d_x, hop, F, d_lut, -50.0, 50.0, d_img, scratch, **kw)
    assert needs == (not one_kernel)
    jsg.stft_image(plan, d_x, hop, F, d_lut, -50.0, 50.0, d_img, None if one_kernel else scratch, **kw)
    torch.cuda.synchronize()
    d_db = torch.zeros((F, (H + 31) // 32 * 32), device="cuda")
    kernel = "Cfg4096B" if (n == 4096 and one_kernel) else jsg.stft_kernel_name(plan, d_x, hop, F, d_db, feedblocks=n // hop)
    ref_db = mirror.columns(kernel, x, hop, F, win, feedblocks=n // hop, exact_db=True)
    pal = oracle.OracleColorPalette(256, scheme)
    pal.set_value_range(-50.0, 50.0)
    ring = np.full((W, H), -120.0, np.float32)
    ring[:F] = ref_db
    want = oracle.render_all(ring, 0, pal, running=True)
    got = d_img.cpu().numpy().view(np.uint32)
    flips = int((got[:, :F] != want[:, :F]).sum())
    assert flips == 0, f"{flips} of {F * H} pixels differ from the mirror's image ({kernel}, one kernel: {one_kernel})"
    # ... and the default logarithm on the same launch stays what it was: a handful of flips at most
    d_img2 = torch.zeros_like(d_img)
    kw["exact_log"] = False
    jsg.stft_image(plan, d_x, hop, F, d_lut, -50.0, 50.0, d_img2, None if one_kernel else scratch, **kw)
    torch.cuda.synchronize()
    assert int((d_img2.cpu().numpy().view(np.uint32)[:, :F] != want[:, :F]).sum()) <= max(4, F * H // 20000)


def test_exact_log_strided_image_batch_equals_single_exact_launches(jsg, oracle, torch_cuda):
    torch = torch_cuda
    n, C, hop, F, K = 4096, 2, 512, 1875, 3
    H = n // 2 + 1
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_in = _rand_in(torch, (K, C, (F - 1) * hop + n), seed=55)
    d_lut = torch.from_numpy(jsg.colormap_lut(256, 6)).cuda()
    one = torch.zeros((K, H, F), dtype=torch.int32, device="cuda")
    many = torch.zeros((K, H, F), dtype=torch.int32, device="cuda")
    kw = dict(feedblocks=n // hop, ring_width=F, exact_log=True, plan_select=2)
    for b in range(K):
        jsg.stft_image(plan, d_in[b], hop, F, d_lut, -50.0, 50.0, one[b], None, **kw)
    jsg.stft_image_strided(plan, d_in, hop, F, d_lut, -50.0, 50.0, many, None, **kw)
    torch.cuda.synchronize()
    assert torch.equal(one, many)


def test_lossless_producer_call_loses_nothing(jsg, oracle, torch_cuda):
    """More blocks than the engine's queue has slots (64), pushed back to back: the lossless entry point (the Python class's default,
    jsg_process_block_wait) waits for the worker instead of dropping, and the ring equals that of one batch launch.  The wait-free call
    of a live host may drop in the same loop -- and says so."""
    n, C, K = 1024, 2, 300
    x = oracle.synth_audio(C, K * n, seed=4242)
    a = jsg.Spectrogram(C)
    b = jsg.Spectrogram(C)
    for s in (a, b):
        s.setSamplerate(48000.0); s.setmemoryTime_s(10.0); s.setFFTSize(n); s.setfeed_percent(jsg.Spectrogram.FeedPercentage.perc50)
    for k in range(K):
        assert a.processSynchronBlock([x[c, k * n:(k + 1) * n] for c in range(C)]) == 0
    b.processBlocks(x)
    W, H = a.getMemorySize(), a.getSpectrumSize()
    ma, mb = np.zeros((W, H), np.float32), np.zeros((W, H), np.float32)
    na, pa = a.getMem(ma)
    nb, pb = b.getMem(mb)
    assert a.droppedBlocks() == 0 and pa == pb == (2 * K) % W
    assert (ma.view(np.uint32) == mb.view(np.uint32)).all()
    # the wait-free call: whatever it drops it counts, and what it queued is in the ring in order
    c = jsg.Spectrogram(C)
    c.setSamplerate(48000.0); c.setmemoryTime_s(10.0); c.setFFTSize(n); c.setfeed_percent(jsg.Spectrogram.FeedPercentage.perc50)
    rcs = [c.processSynchronBlock([x[ch, k * n:(k + 1) * n] for ch in range(C)], realtime=True) for k in range(K)]
    assert set(rcs) <= {0, 1} and c.droppedBlocks() == sum(rcs)
    mc = np.zeros((W, H), np.float32)
    nc, pc = c.getMem(mc)
    assert pc == (2 * (K - sum(rcs))) % W
    for s in (a, b, c):
        s.close()


# ---- the pair plan (Cfg2048P: VERDICT r4 item 1): a channel pair as ONE complex transform, sum-type mixes over an even channel count ----
PAIR_CASES = [  # (channels, hop, feedblocks, mix, window, signal, frames)
    (2, 512, 4, "absmean", 1, "synth", 96), (4, 512, 4, "absmean", 1, "noise", 96), (8, 512, 4, "absmean", 1, "synth", 96),
    (6, 1024, 2, "absmean", 3, "noise", 64), (4, 2048, 1, "sum", 0, "synth", 40), (8, 205, 10, "absmean", 2, "noise", 50),
    (2, 512, 4, "absmean", 5, "sparse", 48), (16, 512, 4, "absmean", 4, "quiet", 32),
]


def _pair_signal(oracle, C, n_samples, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "synth":
        return oracle.synth_audio(C, n_samples, seed=seed)
    if kind == "noise":
        return rng.uniform(-1, 1, (C, n_samples)).astype(np.float32)
    if kind == "quiet":
        return (rng.uniform(-1, 1, (C, n_samples)) * 1e-18).astype(np.float32)
    x = np.zeros((C, n_samples), np.float32)
    x[:, rng.integers(0, n_samples, 7)] = 1.0
    return x


@pytest.mark.parametrize("C,hop,fb,mix,win_kind,signal,F", PAIR_CASES)
def test_pair_plan_equals_the_mirror_bit_for_bit_and_the_float64_oracle_within_the_bound(jsg, oracle, mirror, torch_cuda, C, hop, fb, mix, win_kind, signal, F):
    """plan_select = 3: Cfg2048P.  Linear power and exact-log dB against oracle/jsg_mirror.c (every bit), linear power against the float64
    DFT of the float32 windowed frames inside the parity bound of tests/parity_util.py (the bound of every other 2048-point kernel)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_util import assert_power_close
    torch = torch_cuda
    cap = jsg.capi
    m = {"absmean": cap.MIX_ABSMEAN, "sum": cap.MIX_SUM}[mix]
    n = 2048
    last = F - 1
    n_samples = (last // fb) * n + (last % fb) * hop + n
    x = _pair_signal(oracle, C, n_samples, signal, seed=C + hop)
    win = oracle.window(win_kind, n)
    plan = jsg.Plan(n, win)
    H, pitch = n // 2 + 1, 1056
    d_x = torch.from_numpy(x).cuda()
    d_lin = torch.zeros((F, pitch), device="cuda")
    d_db = torch.zeros((F, pitch), device="cuda")
    kw = dict(feedblocks=fb, mix_mode=m, plan_select=3)
    assert jsg.stft_kernel_name(plan, d_x, hop, F, d_lin, **kw) == "Cfg2048P"
    jsg.stft_db(plan, d_x, hop, F, d_lin, linear_out=True, **kw)
    jsg.stft_db(plan, d_x, hop, F, d_db, exact_log=True, **kw)
    torch.cuda.synchronize()
    ref_lin = mirror.columns("Cfg2048P", x, hop, F, win, feedblocks=fb, mix=m)
    ref_db = mirror.columns("Cfg2048P", x, hop, F, win, feedblocks=fb, mix=m, exact_db=True)
    got_lin = d_lin[:, :H].cpu().numpy()
    bad = got_lin.view(np.uint32) != ref_lin.view(np.uint32)
    assert not bad.any(), (f"Cfg2048P: {int(bad.sum())} of {bad.size} power values differ from the mirror; first at {tuple(np.argwhere(bad)[0])}: "
                           f"{got_lin[bad][0]!r} vs {ref_lin[bad][0]!r}")
    assert (d_db[:, :H].cpu().numpy().view(np.uint32) == ref_db.view(np.uint32)).all()
    if signal in ("synth", "noise"):
        j = np.arange(F)
        starts = (j // fb) * n + (j % fb) * hop
        idx = starts[:, None] + np.arange(n)[None, :]
        frames = (x[:, idx] * win[None, None, :]).astype(np.float32)
        p64 = oracle.power_spectrum_f64(frames)
        ref = oracle.mix_channels(p64.astype(np.float32), oracle.MIX_ABSMEAN).astype(np.float64) if mix == "absmean" else p64.astype(np.float32).sum(axis=0, dtype=np.float32).astype(np.float64)
        assert_power_close(got_lin, ref, f"Cfg2048P C={C} hop={hop}")


def test_pair_plan_selection(jsg, oracle, torch_cuda):
    """Where the pair plan does not apply -- an odd channel count, a selecting mix, per-channel rows -- plan_select = 3 means 'automatic'
    and today's kernels run; where it applies it is named."""
    torch = torch_cuda
    n, hop, F = 2048, 512, 64
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    out = torch.zeros((F, 1056), device="cuda")
    for C, mix, want in ((3, jsg.capi.MIX_ABSMEAN, "Cfg2048"), (5, jsg.capi.MIX_SUM, "Cfg2048"), (4, jsg.capi.MIX_MAX, "Cfg2048"),
                         (4, jsg.capi.MIX_LEFT, "Cfg2048"), (4, jsg.capi.MIX_ABSMEAN, "Cfg2048P"), (2, jsg.capi.MIX_SUM, "Cfg2048P")):
        d_x = _rand_in(torch, (C, (F - 1) * hop + n), seed=C)
        assert jsg.stft_kernel_name(plan, d_x, hop, F, out, feedblocks=4, mix_mode=mix, plan_select=3) == want, (C, mix)
        jsg.stft_db(plan, d_x, hop, F, out, feedblocks=4, mix_mode=mix, plan_select=3)      # ... and runs
    torch.cuda.synchronize()
    # an odd count through the pair request equals the automatic choice bit for bit (it IS the automatic choice)
    d_x = _rand_in(torch, (3, (F - 1) * hop + n), seed=9)
    a, b = torch.zeros((F, 1056), device="cuda"), torch.zeros((F, 1056), device="cuda")
    jsg.stft_db(plan, d_x, hop, F, a, feedblocks=4, plan_select=3)
    jsg.stft_db(plan, d_x, hop, F, b, feedblocks=4, plan_select=0)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("C,K,F,W,pos,tail", [(8, 3, 4096, 4096, 0, False), (2, 5, 700, 800, 750, False), (4, 4, 2048, 2050, 5, True), (8, 12, 4096, 4096, 0, True)])
def test_pair_plan_strided_batches_equal_single_launches(jsg, oracle, torch_cuda, C, K, F, W, pos, tail):
    torch = torch_cuda
    n, hop, M = 2048, 512, 1024
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_in = _rand_in(torch, (K, C, (F - 1) * hop + n), seed=K + C)
    pitch = M if tail else 1056
    ref = torch.full((K, W, pitch), -7.0, device="cuda")
    got = torch.full((K, W, pitch), -7.0, device="cuda")
    t_ref = torch.full((K, 1, W), -7.0, device="cuda") if tail else None
    t_got = torch.full((K, 1, W), -7.0, device="cuda") if tail else None
    kw = dict(feedblocks=4, ring_pos=pos, plan_select=3)
    assert jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, got, d_tail=t_got, **kw) == "Cfg2048P"
    for b in range(K):
        jsg.stft_db(plan, d_in[b], hop, F, ref[b], d_tail=(t_ref[b] if tail else None), **kw)
    jsg.stft_db_strided(plan, d_in, hop, F, got, d_tail=t_got, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    if tail:
        assert torch.equal(t_got, t_ref)
        cols = (pos + torch.arange(F, device="cuda")) % W
        lay = torch.full((W, 1056), -7.0, device="cuda")
        jsg.stft_db(plan, d_in[0], hop, F, lay, **kw)                       # the reference layout holds the same values
        torch.cuda.synchronize()
        assert torch.equal(lay[:, :M], ref[0]) and torch.equal(lay[cols, M], t_ref[0, 0, cols])


def test_columns_from_tail_layout_gives_the_reference_shape(jsg, oracle, torch_cuda):
    """jsg_columns_from_tail_layout_launch: whole-line columns + the plane of bin n/2 -> the dense [W][n/2+1] columns getMem hands out
    (m_mem[col][bin], Spectrogram.h:144), equal to a launch in the reference layout."""
    torch = torch_cuda
    n, hop, F, W, pos = 1024, 512, 300, 320, 310
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_x = _rand_in(torch, (2, (F - 1) * hop + n), seed=77)
    ref = torch.full((W, 544), -120.0, device="cuda")
    dense = torch.full((W, 512), -120.0, device="cuda")
    tail = torch.full((1, W), -120.0, device="cuda")
    jsg.stft_db(plan, d_x, hop, F, ref, ring_pos=pos)
    jsg.stft_db(plan, d_x, hop, F, dense, d_tail=tail, ring_pos=pos)
    got = torch.full((W, 513), 3.0, device="cuda")
    jsg.columns_from_tail_layout(dense, tail[0], got)
    torch.cuda.synchronize()
    assert torch.equal(got, ref[:, :513])
    with pytest.raises(jsg.capi.JsgError):      # a destination that cannot hold n/2 + 1 floats per column
        jsg.capi.check(jsg.capi.lib().jsg_columns_from_tail_layout_launch(dense.data_ptr(), 512, tail.data_ptr(), W, 513, got.data_ptr(), 512, None))
    # padded rows on BOTH sides (pitch 544 and 544) -- the height comes from n, never from the row widths
    dense_p = torch.full((W, 544), -120.0, device="cuda")
    jsg.stft_db(plan, d_x, hop, F, dense_p, d_tail=tail, ring_pos=pos)
    got_p = torch.full((W, 544), 3.0, device="cuda")
    jsg.columns_from_tail_layout(dense_p, tail[0], got_p, n=n)
    torch.cuda.synchronize()
    assert torch.equal(got_p[:, :513], ref[:, :513]) and bool((got_p[:, 513:] == 3.0).all())
    with pytest.raises(jsg.capi.JsgError):      # padded source rows and no n: refused instead of guessed
        jsg.columns_from_tail_layout(dense_p, tail[0], got_p)


# ---- round 5: the palette index without the two selects, where the value range allows it (color_index2_fast / cmap_is_fast) ----
# The launcher decides per range; the image must be that of the dB columns + the colour kernel (which keeps the reference's form, selects
# included) bit for bit either way.  Ranges: the plugin's default (fast), a negative maximum (max * 0.9999 lies ABOVE max), a range whose
# replaced value does not reach the last colour (narrow range far from zero: the selects are needed), an inverted range, a range the
# signal saturates on both sides, few colours.
@pytest.mark.parametrize("lo,hi,n_colors", [(-50.0, 50.0, 256), (-120.0, -10.0, 256), (999.0, 1000.0, 256), (-30.0, -29.5, 256), (20.0, -80.0, 256),
                                            (-70.0, -35.0, 256), (-50.0, 50.0, 7), (-0.001, 0.001, 256), (-90.0, 0.0, 64)])
@pytest.mark.parametrize("n,C,F,one_kernel", [(4096, 2, 1875, True), (1024, 1, 2048, True), (2048, 2, 600, False), (4096, 2, 300, False)])
def test_display_launch_equals_db_plus_colour_kernel_for_every_value_range(jsg, oracle, torch_cuda, lo, hi, n_colors, n, C, F, one_kernel):
    torch = torch_cuda
    hop = n // 8
    H = n // 2 + 1
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    # a signal whose dB values cross the whole range: a tone on noise, the level swept over 140 dB along the stream
    ns = (F - 1) * hop + n
    t = np.arange(ns, dtype=np.float64)
    rng = np.random.default_rng(n + C + F)
    x = np.stack([(np.sin(2 * np.pi * (0.01 + 0.07 * c) * t) + 0.3 * rng.standard_normal(ns)) * 10.0 ** (-7.0 + 7.5 * t / ns) for c in range(C)]).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    d_lut = torch.from_numpy(jsg.colormap_lut(n_colors, jsg.capi.CM_JADE)).cuda()
    pitch = (F + 31) // 32 * 32
    W = (H + 31) // 32 * 32
    d_db = torch.empty((F, W), device="cuda")
    two = torch.zeros((H, pitch), dtype=torch.int32, device="cuda")
    fused = torch.zeros_like(two)
    scratch = torch.zeros((F, W), dtype=torch.uint8, device="cuda")
    kw = dict(feedblocks=8, mix_mode=jsg.capi.MIX_ABSMEAN)
    jsg.stft_db(plan, d_in, hop, F, d_db, **kw)
    jsg.colormap(d_db, d_lut, lo, hi, d_argb=two[:, :F], col_first=0, x_first=0, height=H)
    needs = jsg.stft_image_needs_scratch(plan, d_in, hop, F, d_lut, lo, hi, fused[:, :F], scratch, **kw)
    assert needs == (not one_kernel)
    jsg.stft_image(plan, d_in, hop, F, d_lut, lo, hi, fused[:, :F], scratch, **kw)
    torch.cuda.synchronize()
    assert torch.equal(fused, two), f"{int((fused != two).sum())} pixels differ"
    if not one_kernel:   # the index scratch of the two-kernel form: the oracle's palette index of the GPU's own dB values
        pal = oracle.OracleColorPalette(n_colors, oracle.CM_JADE)
        pal.set_value_range(lo, hi)
        assert (scratch[:, :H].cpu().numpy() == pal.index(d_db[:, :H].cpu().numpy()).astype(np.uint8)).all()
