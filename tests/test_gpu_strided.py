"""jsg_stft_db_launch_strided: K independent batches of one geometry in ONE kernel launch must give, bit for bit, the columns of K
single jsg_stft_db_launch calls -- for ragged frame counts, ring wrap, every hop class, mixes, per-channel rows, unaligned inputs, every
plan, grids whose step count is not a multiple of the grid size (surplus steps start over with the first groups)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _batches(torch, K, C, n_samples, seed, lead=0):
    """[K][C][n_samples] float32 on the GPU (a view that starts `lead` floats into its allocation: alignment cases)."""
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    flat = torch.empty(K * C * n_samples + lead + 64, device="cuda")
    flat.uniform_(-1.0, 1.0, generator=g)
    t = torch.arange(K * C * n_samples, device="cuda", dtype=torch.float64)
    flat[lead:lead + K * C * n_samples] = (0.3 * flat[lead:lead + K * C * n_samples].double() + 0.5 * torch.sin(2 * np.pi * 997.0 * t / 48000.0)).float()
    return flat[lead:lead + K * C * n_samples].view(K, C, n_samples)


def _run_case(jsg, oracle, torch, n, C, F, K, hop, fb=None, mix=None, W=None, ring_pos=0, first_frame=0, lead=0, linear=False,
              plan_select=0, expect_kernel=None, window=None, same_input=False, blocks_per_cu=0):
    mix = jsg.capi.MIX_ABSMEAN if mix is None else mix
    fb = fb if fb is not None else max(1, n // hop)
    H, pitch = n // 2 + 1, (n // 2 + 1 + 31) // 32 * 32
    W = W or F
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN if window is None else window, n))
    last = first_frame + F - 1
    n_samples = ((last // fb) * n + (last % fb) * hop + n + 3) // 4 * 4
    d_in = _batches(torch, 1 if same_input else K, C, n_samples, seed=n + 7 * F + K, lead=lead)
    if same_input:
        d_in = d_in.expand(K, C, n_samples)
    per_ch = mix == jsg.capi.MIX_PER_CHANNEL
    shape = (K, C, W, pitch) if per_ch else (K, W, pitch)
    ref = torch.full(shape, -7.0, device="cuda")
    got = torch.full(shape, -7.0, device="cuda")
    kw = dict(feedblocks=fb, mix_mode=mix, ring_pos=ring_pos, first_frame=first_frame, linear_out=linear)
    name = jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, got, plan_select=plan_select, **kw)
    if expect_kernel is not None:
        assert name == expect_kernel, (name, expect_kernel)
    # the single launches, pinned to the plan the whole strided launch takes (2048 / 4096 points: the rule looks at the total size)
    pin = 0
    if n in (1024, 2048, 4096):
        pin = 2 if name.endswith("B") else 1
    for b in range(K):
        jsg.stft_db(plan, d_in[b], hop, F, ref[b], plan_select=pin, **kw)
    jsg.stft_db_strided(plan, d_in, hop, F, got, plan_select=plan_select, blocks_per_cu=blocks_per_cu, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got[..., :H], ref[..., :H]), f"{name}: strided launch differs from {K} single launches"
    assert torch.equal(got, ref), f"{name}: something outside the columns was written"
    return name


@pytest.mark.parametrize("F,K,hop,ring", [(4096, 5, 512, (4096, 0)), (4096, 5, 512, (5000, 3000)), (1000, 7, 512, (1000, 0)), (1001, 9, 256, (1200, 700)),
                                           (16, 300, 512, (16, 5)), (17, 300, 128, (40, 39)), (4096, 3, 64, (4096, 0)), (333, 40, 4, (400, 100))])
def test_strided_equals_single_launches_mono_1024(jsg, oracle, torch_cuda, F, K, hop, ring):
    """1024 points, one channel: whole and ragged rows, ring wrap, overlaps from 50 % to 99.6 %."""
    assert _run_case(jsg, oracle, torch_cuda, 1024, 1, F, K, hop, W=ring[0], ring_pos=ring[1]) == "Cfg1024"


@pytest.mark.parametrize("bpc", [1, 2, 3, 5, 7, 9, 24, 100])
def test_grid_sizes_that_do_not_divide_the_step_count(jsg, oracle, torch_cuda, bpc):
    """30 batches x 125 groups = 3750 steps over grids of 256 x bpc workgroups: surplus steps of the last round recompute the FIRST groups
    (same bits, stored twice) instead of piling up on the last one."""
    _run_case(jsg, oracle, torch_cuda, 1024, 1, 1000, 30, 512, blocks_per_cu=bpc)
    _run_case(jsg, oracle, torch_cuda, 1024, 2, 333, 11, 256, blocks_per_cu=bpc, W=400, ring_pos=399)


@pytest.mark.parametrize("C,mix", [(2, "absmean"), (3, "absmean"), (8, "absmean"), (2, "left"), (2, "right"), (5, "sum"), (8, "per_channel"), (3, "per_channel")])
def test_mixes_and_per_channel_rows(jsg, oracle, torch_cuda, C, mix):
    m = {"absmean": jsg.capi.MIX_ABSMEAN, "left": jsg.capi.MIX_LEFT, "right": jsg.capi.MIX_RIGHT, "sum": jsg.capi.MIX_SUM,
         "per_channel": jsg.capi.MIX_PER_CHANNEL}[mix]
    _run_case(jsg, oracle, torch_cuda, 1024, C, 700, 6, 512, mix=m, W=800, ring_pos=500, expect_kernel="Cfg1024")


def test_linear_power_first_frame_other_windows_and_shared_input(jsg, oracle, torch_cuda):
    _run_case(jsg, oracle, torch_cuda, 1024, 2, 900, 5, 512, linear=True)
    _run_case(jsg, oracle, torch_cuda, 1024, 1, 900, 5, 512, first_frame=37)
    _run_case(jsg, oracle, torch_cuda, 1024, 1, 900, 5, 256, first_frame=3, window=oracle.WIN_FLATTOP)
    _run_case(jsg, oracle, torch_cuda, 1024, 1, 640, 5, 512, same_input=True)     # in_batch_stride = 0


@pytest.mark.parametrize("what", ["unaligned", "hop1024", "perc10", "hop_odd", "short_rows", "max"])
def test_odd_geometries(jsg, oracle, torch_cuda, what):
    """Unaligned rows, no overlap, the reference's irregular perc10 hop, odd hops, rows shorter than a workgroup step; Max / Min go out
    batch by batch: same columns."""
    cap = jsg.capi
    if what == "unaligned":
        _run_case(jsg, oracle, torch_cuda, 1024, 1, 800, 6, 512, lead=1, expect_kernel="Cfg1024")
    elif what == "hop1024":
        _run_case(jsg, oracle, torch_cuda, 1024, 1, 800, 6, 1024, expect_kernel="Cfg1024")
    elif what == "perc10":
        _run_case(jsg, oracle, torch_cuda, 1024, 2, 800, 6, 102, fb=10, expect_kernel="Cfg1024")
    elif what == "hop_odd":
        _run_case(jsg, oracle, torch_cuda, 1024, 1, 800, 6, 205, fb=1, expect_kernel="Cfg1024")
    elif what == "short_rows":
        _run_case(jsg, oracle, torch_cuda, 1024, 1, 9, 50, 512, expect_kernel="Cfg1024")
    else:
        _run_case(jsg, oracle, torch_cuda, 1024, 3, 500, 4, 512, mix=cap.MIX_MAX)
        _run_case(jsg, oracle, torch_cuda, 1024, 3, 500, 4, 512, mix=cap.MIX_MIN)


@pytest.mark.parametrize("n,C,F,K,hop,mix", [(512, 1, 3000, 5, 256, "absmean"), (512, 2, 777, 4, 128, "absmean"), (2048, 1, 1500, 4, 1024, "absmean"),
                                            (2048, 8, 1024, 4, 512, "absmean"), (2048, 8, 300, 3, 512, "absmean"), (2048, 3, 500, 3, 512, "per_channel"),
                                            (4096, 2, 1875, 3, 512, "absmean"), (4096, 2, 100, 3, 2048, "absmean"), (4096, 1, 640, 5, 1024, "left"),
                                            (8192, 1, 300, 3, 4096, "absmean"), (8192, 2, 150, 2, 2048, "sum")])
def test_every_plan_strided(jsg, oracle, torch_cuda, n, C, F, K, hop, mix):
    """Every FFT size through its strided instantiation (both kernels of 2048 / 4096 points, whichever the rule picks for the total)."""
    m = {"absmean": jsg.capi.MIX_ABSMEAN, "left": jsg.capi.MIX_LEFT, "sum": jsg.capi.MIX_SUM, "per_channel": jsg.capi.MIX_PER_CHANNEL}[mix]
    _run_case(jsg, oracle, torch_cuda, n, C, F, K, hop, mix=m, W=F + 13, ring_pos=F // 2)


def test_c3_geometry_takes_the_b_kernel_for_the_whole_launch(jsg, oracle, torch_cuda):
    assert _run_case(jsg, oracle, torch_cuda, 2048, 8, 4096, 2, 512) == "Cfg2048B"


def test_strided_launch_is_graph_capturable_and_one_batch_is_a_plain_launch(jsg, oracle, torch_cuda):
    torch = torch_cuda
    n, F, K, hop = 1024, 2048, 6, 512
    H, pitch = n // 2 + 1, 544
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_in = _batches(torch, K, 1, F * hop + n - hop, seed=5)
    ref = torch.empty((K, F, pitch), device="cuda")
    for b in range(K):
        jsg.stft_db(plan, d_in[b], hop, F, ref[b])
    got = torch.full((K, F, pitch), -7.0, device="cuda")
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            jsg.stft_db_strided(plan, d_in, hop, F, got, stream=st.cuda_stream)
    torch.cuda.synchronize()
    got.fill_(-7.0)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got[..., :H], ref[..., :H])
    one = torch.full((1, F, pitch), -7.0, device="cuda")
    jsg.stft_db_strided(plan, d_in[:1], hop, F, one)
    torch.cuda.synchronize()
    assert torch.equal(one[0, :, :H], ref[0, :, :H])


def test_strided_bad_arguments_are_rejected(jsg, oracle, torch_cuda):
    import ctypes as C
    torch = torch_cuda
    from jadespectrogram_amd.spectrogram import _stft_args
    lib = jsg.capi.lib()
    n, F, hop = 1024, 64, 512
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    d_in = _batches(torch, 3, 1, F * hop + n - hop, seed=1)
    d_out = torch.empty((3, F, 544), device="cuda")
    a = _stft_args(plan, d_in[0], hop, F, d_out[0])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok_in, ok_out = int(d_in.stride(0)), int(d_out.stride(0))
    assert lib.jsg_stft_db_launch_strided(plan._p, C.byref(a), 3, ok_in, ok_out, st) == 0
    assert lib.jsg_stft_db_launch_strided(plan._p, C.byref(a), 0, ok_in, ok_out, st) == 0
    assert lib.jsg_stft_db_launch_strided(plan._p, C.byref(a), -1, ok_in, ok_out, st) == jsg.capi.JSG_ERR_INVALID
    assert lib.jsg_stft_db_launch_strided(plan._p, C.byref(a), 3, -8, ok_out, st) == jsg.capi.JSG_ERR_INVALID
    assert lib.jsg_stft_db_launch_strided(plan._p, C.byref(a), 3, ok_in, ok_out - 544, st) == jsg.capi.JSG_ERR_INVALID      # rings overlap
    assert lib.jsg_stft_db_launch_strided(plan._p, C.byref(a), 3, ok_in, 0, st) == jsg.capi.JSG_ERR_INVALID
    assert lib.jsg_stft_db_launch_strided(None, C.byref(a), 3, ok_in, ok_out, st) == jsg.capi.JSG_ERR_INVALID
    torch.cuda.synchronize()


def _fuzz_cases(default):
    import os
    return int(os.environ.get("JSG_FUZZ_CASES", default))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("JSG_FUZZ_STRIDED_SEEDS", "4"))))
def test_seeded_random_strided_db_batches(jsg, oracle, torch_cuda, seed):
    """Random plan / channel count / mix / hop pattern / frame count / batch count / ring geometry / grid size: the strided dispatch
    against single launches, bit for bit (JSG_FUZZ_CASES widens the sweep)."""
    import os
    rng = np.random.default_rng(1000 + seed + 97 * int(os.environ.get("JSG_FUZZ_SEED", "0")))
    cap = jsg.capi
    cases = max(12, _fuzz_cases(48) // 4)
    for _ in range(cases):
        n = int(rng.choice([512, 1024, 1024, 2048, 4096, 8192]))
        C = int(rng.choice([1, 1, 2, 3, 8]))
        mix = [cap.MIX_ABSMEAN, cap.MIX_ABSMEAN, cap.MIX_SUM, cap.MIX_LEFT, cap.MIX_PER_CHANNEL, cap.MIX_MAX][int(rng.integers(0, 6))]
        if mix == cap.MIX_MAX and C == 1:
            mix = cap.MIX_ABSMEAN
        kind = int(rng.integers(0, 4))
        if kind == 0:
            hop, fb = n // 2, 2
        elif kind == 1:
            hop, fb = n // 4, 4
        elif kind == 2:
            hop, fb = int(0.1 * n + 0.5), 10            # the reference's perc10: irregular last hop
        else:
            hop, fb = int(rng.integers(1, n // 8)) * 4, 1   # a free hop (feedblocks 1: regular only when hop == n)
            fb = 1
        budget = 3.0e6 / n                              # keep a case small: frames x batches x channels
        F = int(rng.integers(1, max(2, int(min(1500, budget / C)))))
        K = int(rng.integers(2, max(3, int(min(40, budget * 4 / (F * C))))))
        W = F + int(rng.integers(0, 50))
        pos = int(rng.integers(0, W))
        bpc = int(rng.choice([0, 0, 1, 3, 7]))
        sel = int(rng.choice([0, 0, 2])) if n == 1024 else 0      # round 6: a third of the 1024-point cases pin the two-stage plan (where it exists)
        _run_case(jsg, oracle, torch_cuda, n, C, F, K, hop, fb=fb, mix=mix, W=W, ring_pos=pos, first_frame=int(rng.integers(0, 5)), blocks_per_cu=bpc,
                  linear=bool(rng.integers(0, 4) == 0), plan_select=sel)
