"""The oracle (CPU restatement) against the reference's own outputs and known answers.  CPU only."""
import numpy as np
import pytest


def test_colormap_luts_match_reference_build(oracle, golden):
    g = golden["colormap"]
    for scheme in range(7):
        ref = np.array(g["lut256"][str(scheme)], dtype=np.int32)
        assert (oracle.compute_colors(256, scheme) == ref).all()
        for n in (2, 7, 64, 100, 1000):
            ref = np.array(g["lut_other"][f"{scheme}:{n}"], dtype=np.int32)
            assert (oracle.compute_colors(n, scheme) == ref).all(), (scheme, n)


def test_colormap_fnv_and_spots_match_survey(oracle, golden):
    k = golden["kats"]
    for scheme, h in k["colormap_fnv1a32_lut256"].items():
        lut = oracle.compute_colors(256, int(scheme))
        assert "%08x" % oracle.fnv1a32(lut.astype("<i4").tobytes()) == h
    jade = oracle.compute_colors(256, oracle.CM_JADE)
    for i, v in k["jade_spots"].items():
        assert "%06x" % jade[int(i)] == v


def test_get_rgb_color_matches_reference_build(oracle, golden):
    for case in golden["colormap"]["map_cases"]:
        p = oracle.OracleColorPalette(case["n_colors"], case["scheme"])
        p.set_value_range(case["lo"], case["hi"])
        st = np.array(case["state_hex"], dtype=np.uint32)
        mine = np.array([p.vmin, p.vmax, p.mult], dtype=np.float32).view(np.uint32)
        assert (mine == st).all()
        v = np.array(case["values_hex"], dtype=np.uint32).view(np.float32)
        assert (p.index(v) == np.array(case["idx"])).all()
        assert (p.get_rgb_color(v) == np.array(case["rgb"])).all()


def test_color_index_kats_default_range(oracle, golden):
    k = golden["kats"]["colormap_range_m50_p50"]
    p = oracle.OracleColorPalette(256, oracle.CM_JADE)
    p.set_value_range(-50.0, 50.0)
    assert abs(float(p.mult) - k["access_mult"]) < 1e-7
    for v, idx in k["index_cases"]:
        assert int(p.index(np.array([v], dtype=np.float32))[0]) == idx, v


def test_switch_scheme_matches_reference(oracle, golden):
    p = oracle.OracleColorPalette(256, 6)
    p.set_color_scheme(4)
    assert (p.lut == np.array(golden["colormap"]["switch_scheme"]["6->4"], dtype=np.int32)).all()


def test_window_kats(oracle, golden):
    k = golden["kats"]
    for wid, d in k["windows_n1024"].items():
        w = oracle.window(int(wid), 1024)
        for key, val in d.items():
            if key.isdigit():
                assert abs(float(w[int(key)]) - val) <= abs(val) * 2e-7, (d["name"], key)
        if d.get("upper_half_zero"):
            assert (w[513:] == 0).all()
    for wid in range(6):
        for n in (1024, 2048, 4096):
            w = oracle.window(wid, n).astype(np.float64)
            assert abs(np.sum(w * w) / n - 1.0) <= k["window_rms_tolerance"] * 1.05  # SURVEY quotes 1.5e-6 to two digits (max seen 1.534e-6)


def test_geometry_kats(oracle, golden):
    k = golden["kats"]
    for n, d in k["geometry_fs48000_mem10"].items():
        for pct, (hop, fb, W) in d.items():
            h = oracle.feed_samples(float(pct), int(n))
            assert h == hop
            assert oracle.memsize_blocks(10.0, 48000.0, h) == W
    for n, last in k["perc10_last_hop"].items():
        h = oracle.feed_samples(10.0, int(n))
        assert int(n) - 9 * h == last
    assert oracle.NEW_ENTRY_SENTINEL == k["new_entry_sentinel"] == 100000000000 % (1 << 32)
    assert abs(float(oracle.G_MIN_VAL_FOR_LOG) - k["log_floor"]) < 1e-19


def test_engine_smoke_kat(oracle, golden):
    k = golden["kats"]["smoke_mono_1khz"]
    s = oracle.OracleSpectrogram(1)
    s.set_samplerate(k["fs"]); s.set_fft_size(k["n"]); s.set_feed_percent(oracle.FEED_50)
    t = np.arange(k["blocks"] * k["n"])
    x = (k["amplitude"] * np.sin(2 * np.pi * 1000.0 * t / k["fs"])).astype(np.float32)
    for b in range(k["blocks"]):
        s.process_synchron_block(x[None, b * k["n"]:(b + 1) * k["n"]])
    mem = np.zeros((s.get_memory_size(), s.get_spectrum_size()), dtype=np.float32)
    nv, pos = s.get_mem(mem)
    assert pos == k["columns"] and nv == oracle.NEW_ENTRY_SENTINEL + k["columns"]
    assert (mem[0] == np.float32(k["first_column_db"])).all()
    assert abs(float(mem[4, 21]) - k["steady_bin21_db"]) < 1e-3
    assert abs(float(mem[4, 22]) - k["steady_bin22_db"]) < 1e-3
    assert (mem[pos:] == np.float32(golden["kats"]["ring_fill_db"])).all()


@pytest.mark.parametrize("feed", [0, 1, 2, 3])
def test_engine_stream_equals_closed_form(oracle, feed):
    """processSynchronBlock block by block == STFT of the stream with N zeros prepended (incl. perc10's hop)."""
    C, n, K = 2, 1024, 5
    s = oracle.OracleSpectrogram(C)
    s.set_samplerate(48000); s.set_memory_time_s(2.0); s.set_fft_size(n); s.set_feed_percent(feed)
    x = oracle.synth_audio(C, K * n)
    for b in range(K):
        s.process_synchron_block(x[:, b * n:(b + 1) * n])
    mem = np.zeros((s.get_memory_size(), s.get_spectrum_size()), dtype=np.float32)
    s.get_mem(mem)
    ref = oracle.stft_db_reference(x, n, s.hop, s.feedblocks, s.window)
    assert ref.shape[0] == K * s.feedblocks
    assert (mem[:ref.shape[0]].view(np.uint32) == ref.view(np.uint32)).all()


def test_get_mem_incremental_and_wrap(oracle):
    s = oracle.OracleSpectrogram(1)
    s.set_samplerate(48000); s.set_memory_time_s(0.1); s.set_fft_size(1024); s.set_feed_percent(oracle.FEED_50)
    W = s.get_memory_size()
    assert W == 9
    x = oracle.synth_audio(1, 1024 * 12, kind="noise")
    mem = np.zeros((W, 513), dtype=np.float32)
    assert s.get_mem(np.zeros((W + 1, 513), dtype=np.float32))[0] == -1
    nv, pos = s.get_mem(mem)
    assert nv == oracle.NEW_ENTRY_SENTINEL and pos == 0 and (mem == -120).all()
    total = 0
    for b in range(12):
        s.process_synchron_block(x[:, b * 1024:(b + 1) * 1024])
        total += 2
        if b % 3 == 2:
            before = mem.copy()
            nv, pos = s.get_mem(mem)
            assert nv == 6 and pos == total % W
            new_cols = [(pos - 1 - i) % W for i in range(6)]
            for c in range(W):
                if c in new_cols:
                    assert (mem[c] == s.mem[c]).all()
                else:
                    assert (mem[c] == before[c]).all()


def test_display_oracle_recompute_matches_closed_form(oracle):
    s = oracle.OracleSpectrogram(1)
    s.set_samplerate(48000); s.set_memory_time_s(0.2); s.set_fft_size(1024); s.set_feed_percent(oracle.FEED_50)
    d = oracle.OracleDisplay(s)
    x = oracle.synth_audio(1, 1024 * 30)
    for b in range(13):
        s.process_synchron_block(x[:, b * 1024:(b + 1) * 1024])
    nv, pos = d.timer_callback()
    img = oracle.render_all(s.mem, pos, d.palette, running=True)
    assert (d.img == img).all()
    # incremental ticks must stay equal to a full recolour
    for b in range(13, 20):
        s.process_synchron_block(x[:, b * 1024:(b + 1) * 1024])
        nv, pos = d.timer_callback()
        assert nv == 2
        assert (d.img == oracle.render_all(s.mem, pos, d.palette, running=True)).all()


def test_c_port_agrees_with_numpy_oracle(oracle):
    """oracle/jsg_oracle_c.c (float32 scalar port, the cpu_baseline leg) against the float64 numpy restatement."""
    from oracle import oracle_c
    port = oracle_c.load()
    for n, hop, fb, C, mix in ((1024, 512, 2, 1, 0), (2048, 512, 4, 3, 0), (1024, 102, 10, 2, 1), (512, 256, 2, 2, 4)):
        K = 4
        x = oracle.synth_audio(C, K * n, seed=4, kind="noise")
        win = oracle.window(oracle.WIN_HANN, n)
        xp = np.concatenate([np.zeros((C, n), np.float32), x], axis=1)
        got = port.stft_db(xp, n, hop, K * fb, win, feedblocks=fb, mix=mix)
        ref = oracle.stft_db_reference(x, n, hop, fb, win, mode=mix)
        assert got.shape == ref.shape
        assert (got[0] == np.float32(-110.0)).all()
        assert np.abs(got - ref).max() < 2e-3 and np.median(np.abs(got - ref)) < 1e-5


# ---- the kernel mirror (oracle/jsg_mirror.c): the bit-exactness checker of the GPU path is itself held to the float64 oracle ----
def test_mirror_of_the_gpu_arithmetic_is_within_the_parity_bound_of_the_float64_dft(oracle):
    from oracle import mirror as mirror_mod
    from parity_util import assert_power_close
    m = mirror_mod.load()
    for n, plans in mirror_mod.PLANS.items():
        for hop, C, win_kind in ((n // 2, 1, oracle.WIN_HANN), (n // 4, 3, oracle.WIN_BLACKMANHARRIS), (n // 4, 2, oracle.WIN_HANN), (n // 2, 8, oracle.WIN_FLATTOP)):
            F = 12
            x = oracle.synth_audio(C, (F - 1) * hop + n, seed=n + C)
            win = oracle.window(win_kind, n)
            idx = (np.arange(F) * hop)[:, None] + np.arange(n)[None, :]
            frames = (x[:, idx] * win[None, None, :]).astype(np.float32)
            ref = oracle.mix_channels(oracle.power_spectrum_f64(frames).astype(np.float32), oracle.MIX_ABSMEAN).astype(np.float64)
            for plan in plans:
                if plan.endswith("P") and C % 2:      # the pair plan transforms channel PAIRS: even channel counts only
                    continue
                if not plan.endswith("P") and C in (2, 8) and n != 2048:   # (the extra even-count cases are there for the pair plan)
                    continue
                got = m.columns(plan, x, hop, F, win, feedblocks=n // hop, mix=oracle.MIX_ABSMEAN)
                assert_power_close(got, ref, f"mirror {plan} C={C}")


def test_shared_exact_logarithm_is_within_two_ulp_of_the_reference_expression():
    """jsg_exact_db (csrc/jsg_exact_math.h, compiled into the mirror and into the GPU's exact_log pass) against the reference's
    float(10.0 * log10(double(p + 1e-11f))) (Spectrogram.cpp:107): at most 2 ulp of the dB value, 4.1e-6 dB absolute."""
    from oracle import mirror as mirror_mod
    m = mirror_mod.load()
    rng = np.random.default_rng(7)
    p = np.concatenate([10.0 ** rng.uniform(-14, 12, 400000), rng.uniform(0, 2, 100000), [0.0, 1e-11, 1.0, 0.99999994, 2.0, 1e-30]]).astype(np.float32)
    got = m.exact_db(p)
    y = (p + np.float32(1e-11)).astype(np.float32)
    ref64 = 10.0 * np.log10(y.astype(np.float64))
    ref = ref64.astype(np.float32)
    ulp = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / np.spacing(np.abs(ref)).astype(np.float64)
    assert ulp.max() <= 2.0 and np.abs(got.astype(np.float64) - ref64).max() < 5e-6
    assert got[p == 0.0][0] == np.float32(10.0 * np.log10(np.float64(np.float32(1e-11))))      # the -110 dB of an all-zero frame
