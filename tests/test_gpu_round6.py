"""Round 6: the two-stage 1024-point plan Cfg1024B (split-radix 16 x 32, 16 lanes per frame, FOUR frames side by side in a wavefront, one
exchange) -- which launches take it, parity against the float64 oracle, and that everything the launcher can route to
it (strided rows, per-channel planes, ring wrap, the tail plane, ragged frame counts, irregular hops, linear output, exact_log) gives the
columns of single pinned launches bit for bit.  Bit-exactness against the CPU mirror of its arithmetic: tests/test_gpu_mirror.py.
Replaces Spectrogram.cpp:50-119 + :137-145 of the reference for those launches, as the three-stage plan does."""
import numpy as np
import pytest

from parity_util import assert_power_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _rand(torch, shape, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    return torch.empty(shape, device="cuda").uniform_(-1.0, 1.0, generator=g)


def test_which_launches_take_the_two_stage_plan(jsg, oracle, torch_cuda):
    """plan_select 2 pins it (sum-type and one-channel mixes, float columns), 1 pins the three-stage plan; automatically it is taken from
    FOUR channels mixed per column on when the launch fills its rounds of one 32-column workgroup per CU (k1024B_min_channels: measured
    +3.3 % at four channels, +4.1..+4.8 % at eight, -3..-7 % at one).  Max / Min and the display launches never."""
    torch = torch_cuda
    n, hop = 1024, 512
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    cap = jsg.capi
    F = 8192                                             # 256 workgroup steps of 32 columns: one full round on 256 CUs
    out = torch.empty((F, 544), device="cuda")

    def name(C, frames=F, **kw):
        d_x = torch.zeros((C, (frames - 1) * hop + n), device="cuda")
        return jsg.stft_kernel_name(plan, d_x, hop, frames, out[:frames], **kw)
    assert name(1) == "Cfg1024" and name(2) == "Cfg1024" and name(3) == "Cfg1024"
    assert name(4) == "Cfg1024B" and name(8) == "Cfg1024B" and name(5, mix_mode=cap.MIX_SUM) == "Cfg1024B"
    assert name(8, frames=1024) == "Cfg1024"            # 32 steps on 256 CUs: the small workgroups fill the chip, the big ones do not
    assert name(8, plan_select=1) == "Cfg1024" and name(1, plan_select=2) == "Cfg1024B" and name(1, frames=48, plan_select=2) == "Cfg1024B"
    assert name(8, mix_mode=cap.MIX_MAX) == "Cfg1024" and name(8, mix_mode=cap.MIX_MIN, plan_select=2) == "Cfg1024"
    assert name(8, mix_mode=cap.MIX_LEFT) == "Cfg1024" and name(8, mix_mode=cap.MIX_LEFT, plan_select=2) == "Cfg1024B"
    per = torch.empty((8, F, 544), device="cuda")
    d_x = torch.zeros((8, (F - 1) * hop + n), device="cuda")
    assert jsg.stft_kernel_name(plan, d_x, hop, F, per, mix_mode=cap.MIX_PER_CHANNEL) == "Cfg1024"     # one channel per column
    # a strided launch is judged by the frames of all its rows
    d_b = torch.zeros((8, 4, (1024 - 1) * hop + n), device="cuda")
    o_b = torch.empty((8, 1024, 544), device="cuda")
    assert jsg.stft_db_strided_kernel_name(plan, d_b, hop, 1024, o_b) == "Cfg1024B"
    assert jsg.stft_db_strided_kernel_name(plan, d_b[:2], hop, 1024, o_b[:2]) == "Cfg1024"


@pytest.mark.parametrize("C,hop,fb,mix,win_kind", [(1, 512, 2, "absmean", 1), (2, 256, 4, "absmean", 3), (8, 512, 2, "absmean", 1), (3, 512, 2, "sum", 4),
                                                    (5, 102, 10, "absmean", 2), (2, 512, 2, "right", 0), (6, 1024, 1, "absmean", 5)])
def test_two_stage_plan_against_the_float64_oracle(jsg, oracle, torch_cuda, C, hop, fb, mix, win_kind):
    """Linear power of Cfg1024B against the float64 DFT of the float32 windowed frames -> the reference's float32 channel mix
    (Spectrogram.cpp:64-76): inside the bound of tests/parity_util.py (1e-5 relative + the per-size floor; 5e-6 within 20 dB of the peak)."""
    torch = torch_cuda
    cap = jsg.capi
    n, F = 1024, 200                                     # 6 whole steps of 32 columns + a ragged one
    m = {"absmean": cap.MIX_ABSMEAN, "sum": cap.MIX_SUM, "right": cap.MIX_RIGHT}[mix]
    last = F - 1
    n_samples = (last // fb) * n + (last % fb) * hop + n
    x = oracle.synth_audio(C, n_samples, seed=600 + C + hop)
    win = oracle.window(win_kind, n)
    plan = jsg.Plan(n, win)
    d_x = torch.from_numpy(x).cuda()
    d_p = torch.zeros((F, 544), device="cuda")
    kw = dict(feedblocks=fb, mix_mode=m, plan_select=2)
    assert jsg.stft_kernel_name(plan, d_x, hop, F, d_p, **kw) == "Cfg1024B"
    jsg.stft_db(plan, d_x, hop, F, d_p, linear_out=True, **kw)
    torch.cuda.synchronize()
    starts = np.array([(j // fb) * n + (j % fb) * hop for j in range(F)])
    frames = (x[:, starts[:, None] + np.arange(n)[None, :]] * win[None, None, :]).astype(np.float32)
    p32 = oracle.power_spectrum_f64(frames).astype(np.float32)
    if mix == "sum":
        acc = np.zeros(p32.shape[1:], np.float32)
        for c in range(C):
            acc = (acc + p32[c]).astype(np.float32)
        ref = acc
    else:
        ref = oracle.mix_channels(p32, {"absmean": oracle.MIX_ABSMEAN, "right": oracle.MIX_RIGHT}[mix])
    assert_power_close(d_p[:, :513].cpu().numpy(), ref.astype(np.float64), f"Cfg1024B C={C} hop={hop} {mix}")


@pytest.mark.parametrize("C,F,K,hop,fb,mix,W,ring_pos,tail,linear,exact", [
    (1, 4096, 3, 512, 2, "absmean", 4096, 0, False, False, False),       # the C2 shape, three batches
    (1, 700, 5, 512, 2, "absmean", 800, 500, True, False, False),        # ring wrap + tail plane, ragged last step
    (8, 333, 4, 512, 2, "per_channel", 400, 390, False, False, False),   # per-channel planes (the C4 shard's form)
    (8, 333, 4, 512, 2, "per_channel", 400, 17, True, False, True),      # ... with the tail plane and the exact logarithm
    (4, 1031, 8, 256, 4, "absmean", 1031, 0, False, False, False),       # the automatic rule (4 channels, 8248 columns)
    (3, 129, 2, 205, 1, "sum", 200, 199, False, True, False),            # odd hop (4-byte aligned pair loads), linear power
    (6, 257, 3, 102, 10, "absmean", 300, 0, True, False, False),         # perc10's irregular hop, IEEE division by 6
])
def test_two_stage_strided_rows_equal_single_pinned_launches(jsg, oracle, torch_cuda, C, F, K, hop, fb, mix, W, ring_pos, tail, linear, exact):
    torch = torch_cuda
    cap = jsg.capi
    n, H = 1024, 513
    m = {"absmean": cap.MIX_ABSMEAN, "sum": cap.MIX_SUM, "per_channel": cap.MIX_PER_CHANNEL}[mix]
    per_ch = mix == "per_channel"
    last = F - 1
    n_samples = ((last // fb) * n + (last % fb) * hop + n + 3) // 4 * 4
    d_in = _rand(torch, (K, C, n_samples), seed=C * 1000 + F)
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HANN, n))
    pitch = 512 if tail else 544
    shape = (K, C, W, pitch) if per_ch else (K, W, pitch)
    rows = C if per_ch else 1
    got, ref = torch.full(shape, -7.0, device="cuda"), torch.full(shape, -7.0, device="cuda")
    t_got = torch.full((K, rows, W), -7.0, device="cuda") if tail else None
    t_ref = torch.full((K, rows, W), -7.0, device="cuda") if tail else None
    kw = dict(feedblocks=fb, mix_mode=m, ring_pos=ring_pos, linear_out=linear, exact_log=exact, plan_select=2)
    assert jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, got, d_tail=t_got, **kw) == "Cfg1024B"
    jsg.stft_db_strided(plan, d_in, hop, F, got, d_tail=t_got, **kw)
    for b in range(K):
        jsg.stft_db(plan, d_in[b], hop, F, ref[b], d_tail=(t_ref[b] if tail else None), **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, ref), "strided rows differ from single launches (or something outside the columns was written)"
    if tail:
        assert torch.equal(t_got, t_ref)
    # ... and the columns are those of the THREE-stage plan inside the float32 bound (the two plans round differently in the last bits)
    three = torch.full(shape, -7.0, device="cuda")
    t3 = torch.full((K, rows, W), -7.0, device="cuda") if tail else None
    kw3 = dict(kw, plan_select=1)
    jsg.stft_db_strided(plan, d_in, hop, F, three, d_tail=t3, **kw3)
    torch.cuda.synchronize()
    cols = [(ring_pos + i) % W for i in range(F)]
    a, b3 = got[..., cols, :H - (1 if tail else 0)], three[..., cols, :H - (1 if tail else 0)]
    if not linear:                                      # compare as power: a bin 60 dB below its column's peak may differ in the second decimal of its dB value
        a, b3 = torch.pow(10.0, a.double() / 10.0), torch.pow(10.0, b3.double() / 10.0)
    assert float(((a - b3).abs() / (b3.abs().amax(dim=-1, keepdim=True) + 1e-30)).max()) < 2e-6
    untouched = [c for c in range(W) if c not in set(cols)]
    if untouched:
        assert bool((got[..., untouched, :] == -7.0).all())


def test_two_stage_plan_in_a_hip_graph_and_with_a_pitched_input(jsg, oracle, torch_cuda):
    """Capturable like every other launch (no host synchronisation, no allocation), and the channel rows may sit at a pitch."""
    torch = torch_cuda
    n, hop, F, C = 1024, 512, 600, 4
    plan = jsg.Plan(n, oracle.window(oracle.WIN_BLACKMANHARRIS, n))
    wide = _rand(torch, (C, (F - 1) * hop + n + 64), seed=3)
    d_x = wide[:, 4:4 + (F - 1) * hop + n]              # rows at a pitch, starting 16 bytes into the allocation
    eager = torch.full((F, 544), -7.0, device="cuda")
    jsg.stft_db(plan, d_x, hop, F, eager, plan_select=2)
    st = torch.cuda.Stream()
    cap = torch.full((F, 544), -7.0, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        jsg.stft_db(plan, d_x, hop, F, cap, plan_select=2, stream=st.cuda_stream)
        torch.cuda.synchronize()
        cap.fill_(-7.0)
        with torch.cuda.graph(g, stream=st):
            jsg.stft_db(plan, d_x, hop, F, cap, plan_select=2, stream=st.cuda_stream)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap, eager)


# ---- round 6: "runs" (stft_db_kernel STREAM == 2) -- strided one-channel 1024-point launches at exactly 50 % overlap: a wavefront transforms
# kRunLen consecutive columns and keeps the overlapped half of the raw frame in registers (only the new hop is loaded inside a run).  Same
# arithmetic: the columns must be those of single launches (which take the plain kernel) bit for bit -- rows shorter than a run, rows that end
# inside a run (the surplus columns are transformed on clamped loads and NOT stored), ring wrap, per-channel planes, Left / Right, a first
# frame, unaligned inputs, the tail plane, grids that do not divide the step count, the exact logarithm, linear output. ----
@pytest.mark.parametrize("C,F,K,mix,W,pos,first,lead,bpc,linear", [
    (1, 1, 3, "absmean", 4, 3, 0, 0, 0, False), (1, 2, 5, "absmean", 2, 0, 0, 0, 0, False), (1, 3, 4, "absmean", 9, 8, 2, 0, 0, False),
    (1, 7, 6, "absmean", 7, 0, 0, 1, 0, False), (1, 8, 2, "absmean", 8, 5, 0, 0, 1, False), (1, 9, 7, "absmean", 40, 36, 5, 0, 0, True),
    (1, 15, 9, "absmean", 15, 14, 0, 3, 3, False), (1, 17, 3, "absmean", 20, 0, 1, 0, 7, False), (1, 701, 6, "absmean", 800, 500, 0, 0, 0, False),
    (1, 4096, 4, "absmean", 4096, 0, 0, 0, 0, False), (2, 333, 5, "right", 400, 390, 0, 0, 0, False), (3, 129, 4, "left", 129, 0, 4, 0, 1, False),
    (8, 333, 4, "per_channel", 400, 390, 0, 0, 0, False), (5, 11, 6, "per_channel", 11, 10, 3, 2, 3, True), (8, 4096, 2, "per_channel", 4096, 100, 0, 0, 0, False),
])
def test_runs_kernel_equals_single_launches(jsg, oracle, torch_cuda, C, F, K, mix, W, pos, first, lead, bpc, linear):
    import test_gpu_strided as ts
    cap = jsg.capi
    m = {"absmean": cap.MIX_ABSMEAN, "left": cap.MIX_LEFT, "right": cap.MIX_RIGHT, "per_channel": cap.MIX_PER_CHANNEL}[mix]
    name = ts._run_case(jsg, oracle, torch_cuda, 1024, C, F, K, 512, fb=2, mix=m, W=W, ring_pos=pos, first_frame=first, lead=lead, linear=linear,
                        blocks_per_cu=bpc, plan_select=1)
    assert name == "Cfg1024"


def test_runs_kernel_with_the_tail_plane_and_the_exact_logarithm(jsg, oracle, torch_cuda):
    torch = torch_cuda
    n, hop, F, K, W, pos = 1024, 512, 203, 5, 256, 250
    plan = jsg.Plan(n, oracle.window(oracle.WIN_HAMMING, n))
    d_in = _rand(torch, (K, 1, (F - 1) * hop + n), seed=12)
    for exact in (False, True):
        got, ref = torch.full((K, W, 512), -7.0, device="cuda"), torch.full((K, W, 512), -7.0, device="cuda")
        t_got, t_ref = torch.full((K, 1, W), -7.0, device="cuda"), torch.full((K, 1, W), -7.0, device="cuda")
        kw = dict(feedblocks=2, ring_pos=pos, exact_log=exact)
        jsg.stft_db_strided(plan, d_in, hop, F, got, d_tail=t_got, **kw)
        for b in range(K):
            jsg.stft_db(plan, d_in[b], hop, F, ref[b], d_tail=t_ref[b], **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, ref) and torch.equal(t_got, t_ref)
