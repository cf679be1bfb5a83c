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
