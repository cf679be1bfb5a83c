"""GPU colour loop: palette indices and ARGB pixels must be bit-exact given the same dB input."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


def _adversarial_db(oracle, W, H, lo, hi, n_colors, seed):
    """dB values sitting on and next to every quantisation boundary, clamps, and random fill."""
    rng = np.random.default_rng(seed)
    lo32, hi32 = np.float32(min(lo, hi)), np.float32(max(lo, hi))
    v = rng.uniform(float(lo32) - 8, float(hi32) + 8, (W, H)).astype(np.float32)
    step = (float(hi32) - float(lo32)) / n_colors
    b = (float(lo32) + step * rng.integers(0, n_colors + 1, (W, H // 2))).astype(np.float32)
    jitter = rng.integers(-2, 3, b.shape)
    for k in (-2, -1, 1, 2):
        sel = jitter == k
        tgt = np.float32(-np.inf) if k < 0 else np.float32(np.inf)
        for _ in range(abs(k)):
            b[sel] = np.nextafter(b[sel], tgt)
    v[:, :H // 2] = b
    v[0, :8] = [lo32, hi32, -200, -120, -110, 1e30, -1e30, 0]
    return v


@pytest.mark.parametrize("scheme", range(7))
@pytest.mark.parametrize("lo,hi,n_colors", [(-50.0, 50.0, 256), (-110.0, 0.0, 256), (20.0, -80.0, 64), (10.0, 10.0, 256)])
def test_colormap_bit_exact(jsg, oracle, torch_cuda, scheme, lo, hi, n_colors):
    torch = torch_cuda
    W, H = 150, 513
    db = _adversarial_db(oracle, W, H, lo, hi, n_colors, seed=scheme)
    pal = oracle.OracleColorPalette(n_colors, scheme)
    pal.set_value_range(lo, hi)
    d_db = torch.from_numpy(db).cuda()
    d_lut = torch.from_numpy(jsg.colormap_lut(n_colors, scheme)).cuda()
    d_img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    d_idx = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    pos = 37
    jsg.colormap(d_db, d_lut, lo, hi, d_argb=d_img, d_index=d_idx, x_first=(W - pos) % W)
    torch.cuda.synchronize()
    img = d_img.cpu().numpy().view(np.uint32)
    idx = d_idx.cpu().numpy()
    ref_img = oracle.render_all(db, pos, pal, running=True)
    ref_idx = np.zeros((H, W), np.uint8)
    xs = (np.arange(W) + (W - pos)) % W
    ref_idx[::-1, :][:, xs] = pal.index(db).T.astype(np.uint8)
    assert (idx == ref_idx).all(), f"{(idx != ref_idx).sum()} palette indices differ"
    assert (img == ref_img).all()


def test_colormap_against_reference_golden_values(jsg, oracle, torch_cuda):
    """The reference's own getRGBColor answers (golden vectors from its CColorpalette.cpp) through the kernel."""
    torch = torch_cuda
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "colormap_ref.json")))
    for case in g["map_cases"]:
        v = np.array(case["values_hex"], dtype=np.uint32).view(np.float32)
        H = v.size
        d_db = torch.from_numpy(v[None, :].copy()).cuda()          # one column
        d_lut = torch.from_numpy(jsg.colormap_lut(case["n_colors"], case["scheme"])).cuda()
        d_img = torch.zeros((H, 1), dtype=torch.int32, device="cuda")
        d_idx = torch.zeros((H, 1), dtype=torch.uint8, device="cuda")
        jsg.colormap(d_db, d_lut, case["lo"], case["hi"], d_argb=d_img, d_index=d_idx)
        torch.cuda.synchronize()
        rgb = d_img.cpu().numpy()[::-1, 0].view(np.uint32)
        idx = d_idx.cpu().numpy()[::-1, 0]
        assert (idx == np.array(case["idx"], dtype=np.uint8)).all()
        assert (rgb == (np.array(case["rgb"], dtype=np.int64) | 0xFF000000).astype(np.uint32)).all()


@pytest.mark.parametrize("running", [True, False])
def test_display_ticks_match_oracle(jsg, oracle, running):
    """timerCallback sequence: full recolour, incremental ticks, slider change, ring wrap."""
    C, n = 2, 1024
    s = jsg.Spectrogram(C); o = oracle.OracleSpectrogram(C)
    s.setSamplerate(48000.0); s.setmemoryTime_s(0.4); s.setFFTSize(n); s.setfeed_percent(1)
    o.set_samplerate(48000.0); o.set_memory_time_s(0.4); o.set_fft_size(n); o.set_feed_percent(1)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    d = jsg.SpectrogramDisplay(s); od = oracle.OracleDisplay(o)
    d.setRunning(running); od.running = running
    img = np.zeros((H, W), np.uint32)
    x = oracle.synth_audio(C, 60 * n, seed=21)
    lo, hi = -50.0, 50.0
    blocks = 0
    for tick, nb in enumerate([3, 2, 0, 5, 1, 30, 2, 2]):
        for _ in range(nb):
            blk = x[:, blocks * n:(blocks + 1) * n]
            s.processSynchronBlock(blk); o.process_synchron_block(blk)
            blocks += 1
        if tick == 4:
            lo, hi = -30.0, 40.0
            d.invalidate(); od.recompute_all = True
        # the colour stage is exact given the same dB: drive the oracle display with the GPU's dB ring
        mem = np.zeros((W, H), np.float32)
        ring_before = o.mem.copy()
        nv, pos = d.timerCallback(img, lo, hi)
        onv, opos = od.timer_callback(lo, hi)
        assert (nv, pos) == (onv, opos), tick
        # end-to-end: count pixels that differ and require each to be a boundary case of the oracle's dB
        diff = img != od.img
        if diff.any():
            ys, xs = np.nonzero(diff)
            cols = (xs + pos) % W if running else xs
            bins = H - 1 - ys
            vals = od.displaymem[cols, bins].astype(np.float64)
            is_cursor = (img[ys, xs] == 0xFFFF0000) | (od.img[ys, xs] == 0xFFFF0000)
            assert not is_cursor.any(), "cursor columns differ"
            pal = od.palette
            frac = (vals - float(pal.vmin)) * float(pal.mult)
            dist = np.abs(frac - np.round(frac)) / float(pal.mult)
            assert (dist < 1e-3).all(), f"tick {tick}: pixel differs away from a quantisation boundary (dist {dist.max()})"
            assert diff.sum() <= max(4, int(2e-3 * img.size)), f"tick {tick}: {diff.sum()} pixels differ"
    s.close()


def test_display_exact_given_gpu_db(jsg, oracle):
    """Full image, bit-exact: recolour of the GPU's own dB ring vs the oracle palette applied to that same ring."""
    C, n = 1, 2048
    s = jsg.Spectrogram(C)
    s.setSamplerate(96000.0); s.setmemoryTime_s(0.5); s.setFFTSize(n); s.setfeed_percent_ext(12.5)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    x = oracle.synth_audio(C, 30 * n, fs=96000.0, seed=2)
    s.processBlocks(x)
    mem = np.zeros((W, H), np.float32)
    nv, pos = s.getMem(mem)
    for scheme, lo, hi in ((6, -50.0, 50.0), (4, -90.0, 10.0), (0, 0.0, 30.0)):
        d = jsg.SpectrogramDisplay(s, 256, scheme)
        img = np.zeros((H, W), np.uint32)
        d.timerCallback(img, lo, hi)
        pal = oracle.OracleColorPalette(256, scheme); pal.set_value_range(lo, hi)
        assert (img == oracle.render_all(mem, pos, pal, running=True)).all()
    s.close()


def test_display_tile_updates_match_full_image(jsg, oracle):
    """Incremental tile ticks (host scrolls its own image like the reference) stay equal to a full recolour."""
    C, n = 1, 1024
    s = jsg.Spectrogram(C)
    s.setSamplerate(48000.0); s.setmemoryTime_s(0.4); s.setFFTSize(n); s.setfeed_percent(1)
    W, H = s.getMemorySize(), s.getSpectrumSize()
    d = jsg.SpectrogramDisplay(s)
    x = oracle.synth_audio(C, 70 * n, seed=17)
    img = np.zeros((H, W), np.uint32)
    tile = np.zeros((H, 16), np.uint32)
    need_full, nv, pos = d.timerCallbackTile(tile)
    assert need_full                                             # first tick: full recolour pending
    d.timerCallback(img)
    blocks = 0
    for nb in (2, 3, 0, 5, 1, 7):
        for _ in range(nb):
            s.processSynchronBlock(x[:, blocks * n:(blocks + 1) * n]); blocks += 1
        need_full, nv, pos = d.timerCallbackTile(tile)
        assert not need_full and nv == 2 * nb
        if nv:                                                   # reference Spectrogram.cpp:665-682
            img[:, :W - nv] = img[:, nv:].copy()
            img[:, W - nv:] = tile[:, :nv]
        full = np.zeros((H, W), np.uint32)
        d.invalidate(); d.timerCallback(full)                    # ground truth: recolour everything
        assert (img == full).all()
    for _ in range(12):                                          # 24 new columns > 16-column tile
        s.processSynchronBlock(x[:, blocks * n:(blocks + 1) * n]); blocks += 1
    need_full, nv, pos = d.timerCallbackTile(tile)
    assert need_full and nv == 24
    s.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("JSG_FUZZ_SCENARIOS", "8")))))
def test_seeded_random_display_walks(jsg, oracle, seed):
    """Random GUI sessions: bursts of audio blocks (also more than a ring's worth), ticks, running <-> fixed display, colour
    range / scheme changes, pause.  After every tick the incrementally maintained image must equal a full recolour of the
    ring, bit for bit (reference Spectrogram.cpp:623-724: scroll + new columns, or fixed mode with its cursor; the
    reference's incremental fixed mode paints the cursor 1 / 2 / 4 columns wide, its full recolour one column: mirrored)."""
    rng = np.random.default_rng(500 + seed)
    C = int(rng.integers(1, 3)); n = int(rng.choice([512, 1024, 2048]))
    s = jsg.Spectrogram(C)
    s.setSamplerate(48000.0); s.setmemoryTime_s(float(rng.choice([0.15, 0.3]))); s.setFFTSize(n); s.setfeed_percent(int(rng.integers(0, 3)))
    W, H = s.getMemorySize(), s.getSpectrumSize()
    d = jsg.SpectrogramDisplay(s)
    x = oracle.synth_audio(C, 256 * 1024, seed=seed)
    img = np.zeros((H, W), np.uint32); full = np.zeros((H, W), np.uint32)
    lo, hi, at, running = -50.0, 50.0, 0, True
    cursor_w = 1 + (H < 2048) + 2 * (H < 1024)      # incremental fixed mode draws a wider cursor than a full recolour (:703-720 vs :650-656)
    for tick in range(25):
        ev = rng.choice(["tick"] * 6 + ["running", "range", "scheme", "pause"])
        if ev == "running":
            running = bool(rng.integers(0, 2))
            d.setRunning(running)
        elif ev == "range":
            lo, hi = float(rng.choice([-90.0, -50.0, -20.0, 10.0])), float(rng.choice([-30.0, 0.0, 50.0]))
            d.invalidate()
        elif ev == "scheme":
            d.setColorSceme(int(rng.integers(0, 7)))
        elif ev == "pause":
            s.setPauseMode(bool(rng.integers(0, 2)))
        for _ in range(int(rng.choice([0, 1, 2, 3, 7, 40]))):
            if at + n > x.shape[1]:
                at = 0
            s.processSynchronBlock(x[:, at:at + n]); at += n
        nv, pos = d.timerCallback(img, lo, hi)
        d.invalidate()
        nv2, pos2 = d.timerCallback(full, lo, hi)
        assert nv2 == 0 and pos2 == pos, (seed, tick)
        same = img == full
        if not running:      # the extra cursor columns pos+1 .. pos+cursor_w-1 are red in the incremental image only
            for dd in range(1, cursor_w):
                c = (pos + dd) % W
                assert (same[:, c] | (img[:, c] == 0xFFFF0000)).all(), (seed, tick, c)
                same[:, c] = True
        assert same.all(), (seed, tick, ev, int((~same).sum()))
    s.close()


def test_colormap_non_finite_values_stay_in_range(jsg, oracle, torch_cuda):
    """NaN / +-Inf dB values (the reference's int(NaN) is undefined behaviour) must map to a valid palette entry."""
    torch = torch_cuda
    v = np.array([[np.nan, np.inf, -np.inf, 1e38, -1e38, 0.0, -0.0, 49.999996]], dtype=np.float32)
    d_db = torch.from_numpy(v.copy()).cuda()
    d_lut = torch.from_numpy(jsg.colormap_lut(256, 6)).cuda()
    d_img = torch.zeros((8, 1), dtype=torch.int32, device="cuda")
    d_idx = torch.full((8, 1), 77, dtype=torch.uint8, device="cuda")
    jsg.colormap(d_db, d_lut, -50.0, 50.0, d_argb=d_img, d_index=d_idx)
    torch.cuda.synchronize()
    idx = d_idx.cpu().numpy()[::-1, 0]
    pal = oracle.OracleColorPalette(256, 6); pal.set_value_range(-50.0, 50.0)
    assert idx[0] == 0                                            # NaN: every comparison false, cvt(NaN) = 0
    assert (idx[1:] == pal.index(v[0, 1:])).all()                 # infinities and huge values clamp like finite ones
    lut = jsg.colormap_lut(256, 6)
    assert (d_img.cpu().numpy()[::-1, 0].view(np.uint32) == (lut[idx].astype(np.int64) | 0xFF000000).astype(np.uint32)).all()
