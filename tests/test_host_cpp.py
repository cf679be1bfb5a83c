"""The C++ drop-in (jadespectrogram_amd/host/*.h: Spectrogram, SynchronBlockProcessor stand-in, CColorPalette) --
compiled without JUCE against libjsg.so; on the GPU box it replays the plugin's call sequence."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_driver(jsg):
    exe = os.path.join(tempfile.gettempdir(), "jsg_host_dropin_test")
    src = os.path.join(ROOT, "tests", "cpp", "host_dropin_test.cpp")
    libdir = os.path.dirname(jsg.capi.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
           "-L", libdir, "-ljsg", f"-Wl,-rpath,{libdir}"]
    subprocess.check_call(cmd)
    return exe


def test_dropin_headers_compile_and_link_without_juce(jsg):
    exe = _build_driver(jsg)
    assert os.path.exists(exe)
    # without arguments the driver exits with its usage code (no GPU touched)
    assert subprocess.call([exe]) == 2


@pytest.mark.gpu
def test_dropin_plugin_call_sequence(jsg, oracle):
    exe = _build_driver(jsg)
    C, n, K = 2, 1024, 9
    x = oracle.synth_audio(C, K * n + 300, seed=31)          # 300 extra samples stay in the re-blocker's FIFO
    with tempfile.TemporaryDirectory() as d:
        fin, fmem, fimg = (os.path.join(d, f) for f in ("in.f32", "mem.f32", "img.u32"))
        x.tofile(fin)
        env = dict(os.environ)
        out = subprocess.check_output([exe, fin, str(C), str(x.shape[1]), str(n), fmem, fimg], env=env).decode()
        info = json.loads(out.strip().splitlines()[-1])
        W, H = info["W"], info["H"]
        mem = np.fromfile(fmem, dtype=np.float32).reshape(W, H)
        img = np.fromfile(fimg, dtype=np.uint32).reshape(H, W)
    o = oracle.OracleSpectrogram(C)
    o.set_samplerate(48000.0); o.set_memory_time_s(1.0); o.set_fft_size(n); o.set_feed_percent(oracle.FEED_50)
    for b in range(K):
        o.process_synchron_block(x[:, b * n:(b + 1) * n])
    assert (W, H) == (o.memsize_blocks, o.freqsize)
    assert info["pos"] == o.mem_counter == info["pos2"]
    assert info["newVals"] == oracle.NEW_ENTRY_SENTINEL + 2 * K
    d = np.abs(mem.astype(np.float64) - o.mem.astype(np.float64))
    assert d.max() < 2e-3, d.max()
    pal = oracle.OracleColorPalette(256, oracle.CM_JADE)
    pal.set_value_range(-50.0, 50.0)
    assert (img == oracle.render_all(mem, info["pos"], pal, running=True)).all()
    assert info["rgb0"] == int(pal.get_rgb_color(np.float32([-200.0]))[0])
    assert info["rgb_mid"] == int(pal.get_rgb_color(np.float32([0.0]))[0])


def test_reblocker_stand_in_cpu():
    """SynchronBlockProcessor stand-in: host blocks of any size -> fixed fft-size blocks, sample order preserved."""
    exe = os.path.join(tempfile.gettempdir(), "jsg_reblocker_test")
    src = os.path.join(ROOT, "tests", "cpp", "reblocker_test.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", src, "-o", exe])
    assert subprocess.call([exe]) == 0
