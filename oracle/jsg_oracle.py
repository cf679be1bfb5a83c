"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference's STFT-spectrogram hot path (SURVEY.md section 8a rows
a1..a11).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (jadespectrogram_amd/) never does and fails loudly when its HIP
library is missing.

Pinning status (DESIGN.md "Oracle"):
  * colour map (a9, a10)        : PINNED against the reference's own CColorpalette.cpp compiled as-is
                                   (oracle/_ref, recipe oracle/Makefile) and the golden vectors it produced
                                   (tests/golden/colormap_ref.json).
  * windows / geometry / framing : pinned by the known-answer values recorded in SURVEY.md section 4
                                   (tests/golden/survey_kats.json).  Spectrogram.cpp itself is unbuildable
                                   here (needs JUCE + the author's TGM library; no stand-ins are written).
  * FFT power scale (a3)         : PARITY UNPINNED -- `spectrum::power` lives in an un-vendored external
                                   library with no pinned version; we define power[k] = power_scale*|X[k]|^2
                                   with X the un-normalised DFT, power_scale = 1, evaluated here in float64.

Every function cites the reference file:line (under /root/reference) it follows.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

f32 = np.float32

# ----------------------------------------------------------------------------------------------
# enums (Spectrogram.h:84-107, CColorpalette.h:9-18) -- same enumerator order as the reference
# ----------------------------------------------------------------------------------------------
MIX_ABSMEAN, MIX_MAX, MIX_MIN, MIX_LEFT, MIX_RIGHT = range(5)          # Spectrogram.h:84-91
WIN_RECT, WIN_HANN, WIN_HAMMING, WIN_BLACKMANHARRIS, WIN_FLATTOP, WIN_HANNPOISSON = range(6)  # :92-100
FEED_100, FEED_50, FEED_25, FEED_10 = range(4)                          # Spectrogram.h:101-107
CM_MONO, CM_BW, CM_HOT, CM_RAINBOW, CM_VIRIDIS, CM_PLASMA, CM_JADE = range(7)  # CColorpalette.h:9-18

FEED_TABLE = {FEED_100: (100.0, 1), FEED_50: (50.0, 2), FEED_25: (25.0, 4), FEED_10: (10.0, 10)}  # Spectrogram.cpp:189-209

G_MIN_VAL_FOR_LOG = f32(0.00000000001)     # Spectrogram.cpp:36  (1e-11f)
NEW_ENTRY_SENTINEL = 1215752192            # int(100000000000) narrowed, Spectrogram.cpp:18,168,236
RING_FILL_DB = f32(-120.0)                 # Spectrogram.cpp:223
G_MAX_COLOR_VAL = f32(50.0)                # PlugInGUISettings.h:37
G_MIN_COLOR_VAL = f32(-50.0)               # PlugInGUISettings.h:38
JUCE_RED_ARGB = 0xFFFF0000                 # juce::Colours::red, used at Spectrogram.cpp:654,717


# ----------------------------------------------------------------------------------------------
# a6 geometry: Spectrogram::buildmem, Spectrogram.cpp:213-218
# ----------------------------------------------------------------------------------------------
def feed_samples(feed_percent: float, fftsize: int) -> int:
    """m_feed_samples = int(m_feed_percent*0.01*m_fftsize+0.5)  (Spectrogram.cpp:216); float*double."""
    return int(float(f32(feed_percent)) * 0.01 * fftsize + 0.5)


def memsize_blocks(memsize_s: float, fs: float, hop: int) -> int:
    """m_memsize_blocks = int(m_memsize_s*m_fs/m_feed_samples + 0.5) (Spectrogram.cpp:217): float*float/float."""
    v = f32(f32(memsize_s) * f32(fs)) / f32(hop)
    return int(float(v) + 0.5)


def next_power_of_2(fftsize_ms: float, fs: float) -> int:
    """Spectrogram::getnextpowerof2, Spectrogram.cpp:171-176."""
    first = f32(float(f32(fftsize_ms)) * 0.001 * float(f32(fs)))          # double product stored to float
    n = int(math.log(float(first)) / float(f32(math.log(2.0)))) + 1        # log(float)->double / log(2.f)->float
    return int(f32(2.0) ** n)


# ----------------------------------------------------------------------------------------------
# a7 windows: Spectrogram::setWindowFkt, Spectrogram.cpp:239-293
# ----------------------------------------------------------------------------------------------
def window(kind: int, n: int) -> np.ndarray:
    """Periodic windows evaluated in double, rounded to float, then RMS-normalised with a float
    accumulator in ascending k (Spectrogram.cpp:283-291).  HannPoisson's size_t wrap
    (Spectrogram.cpp:280) is reproduced: fabs(N-2k) with N-2k unsigned => upper half exactly 0."""
    w = np.zeros(n, dtype=f32)
    nf = f32(0.0)
    two_pi = 2.0 * math.pi
    for k in range(n):
        c1 = math.cos(two_pi * k / n)                       # cos(2.0*M_PI*kk/m_fftsize)
        if kind == WIN_RECT:
            v = f32(1.0)
        elif kind == WIN_HANN:
            v = f32(0.5 * (1.0 - c1))                       # :256
        elif kind == WIN_HAMMING:
            v = f32(25.0 / 46.0 - (1.0 - 25.0 / 46.0) * c1)  # :259
        elif kind == WIN_BLACKMANHARRIS:                    # :262-266, float coefficients
            a0, a1, a2, a3 = (float(f32(x)) for x in (0.35875, 0.48829, 0.14128, 0.01168))
            v = f32(a0 - a1 * c1 + a2 * math.cos(4.0 * math.pi * k / n) - a3 * math.cos(6.0 * math.pi * k / n))
        elif kind == WIN_FLATTOP:                           # :269-275
            a0, a1, a2, a3, a4 = (float(f32(x)) for x in (0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368))
            v = f32(a0 - a1 * c1 + a2 * math.cos(4.0 * math.pi * k / n)
                    - a3 * math.cos(6.0 * math.pi * k / n) + a4 * math.cos(8.0 * math.pi * k / n))
        elif kind == WIN_HANNPOISSON:                       # :278-280
            alpha = 2.0
            d = (n - 2 * k) % (1 << 64)                     # size_t arithmetic: wraps for 2k > N
            arg = -alpha * float(d) / n
            e = math.exp(arg) if arg > -745.0 else 0.0
            v = f32(0.5 * (1.0 - c1) * e)
        else:
            raise ValueError("unknown window")
        w[k] = v
        nf = f32(nf + f32(v * v))                           # :283 float accumulate
    nf = f32(nf / f32(n))                                   # :285
    nf = f32(math.sqrt(float(nf)))                          # :286 (double sqrt of a float, rounded: == sqrtf)
    return (w / nf).astype(f32)                             # :289 float/float


# ----------------------------------------------------------------------------------------------
# a3 FFT power (external in the reference; see module docstring) -- float64 evaluation
# ----------------------------------------------------------------------------------------------
def power_spectrum(frame_f32: np.ndarray, power_scale: float = 1.0) -> np.ndarray:
    """spectrum::power(float* in, vector<float>& out) as used at Spectrogram.cpp:144:
    N real floats in, N/2+1 non-negative floats out.  Evaluated in float64, rounded once."""
    X = np.fft.rfft(frame_f32.astype(np.float64), axis=-1)
    return (power_scale * (X.real * X.real + X.imag * X.imag)).astype(f32)


def power_spectrum_f64(frame_f32: np.ndarray, power_scale: float = 1.0) -> np.ndarray:
    X = np.fft.rfft(frame_f32.astype(np.float64), axis=-1)
    return power_scale * (X.real * X.real + X.imag * X.imag)


def mix_channels(power: np.ndarray, mode: int) -> np.ndarray:
    """Spectrogram.cpp:64-106; power is [C][H] float32 (or [C][F][H]); float accumulation in channel order."""
    C = power.shape[0]
    if mode == MIX_ABSMEAN:
        acc = np.zeros(power.shape[1:], dtype=f32)
        for c in range(C):
            acc = (acc + power[c]).astype(f32)              # :72
        return (acc / f32(C)).astype(f32)                   # :74  float /= size_t
    if mode == MIX_MAX:
        acc = np.zeros(power.shape[1:], dtype=f32)          # :78
        for c in range(C):
            acc = np.where(power[c] > acc, power[c], acc)   # :81
        return acc.astype(f32)
    if mode == MIX_MIN:
        acc = np.full(power.shape[1:], 1000000.0, dtype=f32)  # :86
        for c in range(C):
            acc = np.where(power[c] < acc, power[c], acc)
        return acc.astype(f32)
    if mode == MIX_LEFT:
        return power[0].astype(f32)                         # :94
    if mode == MIX_RIGHT:
        # :98 guards with m_channels>0 (always true) and reads m_power[1]; with one channel that is an
        # out-of-bounds read in the reference.  The only defined behaviour is C>=2 -> channel 1.
        if C < 2:
            raise ValueError("MIX_RIGHT with a single channel is undefined behaviour in the reference")
        return power[1].astype(f32)
    raise ValueError("unknown mix mode")


def to_db(p: np.ndarray) -> np.ndarray:
    """m_powerfinal = 10.0*log10(m_powerfinal + 1e-11f)  (Spectrogram.cpp:107): float add, double log10
    (canonical overload choice, SURVEY section 7 hard part 4), double multiply, rounded to float on store."""
    s = (p.astype(f32) + G_MIN_VAL_FOR_LOG).astype(f32)
    return (10.0 * np.log10(s.astype(np.float64))).astype(f32)


# ----------------------------------------------------------------------------------------------
# a1/a2/a4/a5/a6/a8: the engine, Spectrogram.cpp:16-331
# ----------------------------------------------------------------------------------------------
class OracleSpectrogram:
    """State-for-state restatement of class Spectrogram (engine half).  The channel count is explicit
    (the reference never calls setchannels)."""

    def __init__(self, channels: int = 2):
        # ctor defaults, Spectrogram.cpp:16-24
        self.fs = f32(48000.0)
        self.channels = int(channels)
        self.feed_percent = f32(100.0)
        self.feedblocks = 1
        self.memsize_s = f32(1.0)
        self.fftsize = 1024
        self.mode = MIX_ABSMEAN
        self.window_choice = WIN_HANN
        self.pause = False
        self.power_scale = 1.0
        self._buildmem()
        # NOTE: the ctor does not call setWindowFkt(); m_window is empty until setFFTSize/setWindow.
        self.window = None

    # --- setters (Spectrogram.cpp:148-211, Spectrogram.h:122-123) ---
    def set_samplerate(self, fs):
        self.fs = f32(fs); self._buildmem()

    def set_channels(self, c):
        self.channels = int(c); self._buildmem()

    def set_fft_size(self, n):
        self.fftsize = int(n); self._buildmem(); self._set_window_fkt()
        self.new_entry_counter = NEW_ENTRY_SENTINEL

    def set_closest_fft_size_ms(self, ms):
        self.fftsize = next_power_of_2(ms, self.fs); self._buildmem(); self._set_window_fkt()

    def set_memory_time_s(self, s):
        self.memsize_s = f32(s); self._buildmem()

    def set_feed_percent(self, feed):
        self.feed_percent, self.feedblocks = f32(FEED_TABLE[feed][0]), FEED_TABLE[feed][1]
        self._buildmem()

    def set_feed_percent_ext(self, percent: float):
        """Extension (not offered by the reference): any percentage with
        hop = int(pct*0.01*N+0.5), feedblocks = N // hop."""
        self.feed_percent = f32(percent)
        self.feedblocks = self.fftsize // feed_samples(percent, self.fftsize)
        self._buildmem()

    def set_pause_mode(self, p):
        self.pause = bool(p)

    def set_window(self, w):
        self.window_choice = int(w); self._set_window_fkt()

    def get_spectrum_size(self):
        return self.freqsize

    def get_memory_size(self):
        return self.memsize_blocks

    # --- buildmem, Spectrogram.cpp:213-238 ---
    def _buildmem(self):
        n = self.fftsize
        self.hop = feed_samples(self.feed_percent, n)
        self.memsize_blocks = memsize_blocks(self.memsize_s, self.fs, self.hop)
        self.freqsize = n // 2 + 1
        self.mem = np.full((self.memsize_blocks, self.freqsize), RING_FILL_DB, dtype=f32)
        self.indatamem = np.zeros((self.channels, 2 * n), dtype=f32)
        self.in_counter = n
        self.new_entry_counter = NEW_ENTRY_SENTINEL
        self.mem_counter = 0

    def _set_window_fkt(self):
        self.window = window(self.window_choice, self.fftsize)

    # --- processSynchronBlock, Spectrogram.cpp:37-135 ---
    def process_synchron_block(self, data: np.ndarray) -> int:
        n = self.fftsize
        data = np.asarray(data, dtype=f32)
        assert data.shape == (self.channels, n)
        self.indatamem[:, self.in_counter:self.in_counter + n] = data            # :41-48
        self.in_counter += n
        for bb in range(self.feedblocks):                                        # :50
            off = self.hop * bb
            frames = self.indatamem[:, off:off + n] * self.window[None, :]       # :54-55, :140-141 (float mul)
            power = power_spectrum(frames.astype(f32), self.power_scale)         # :144
            col = to_db(mix_channels(power, self.mode))                          # :64-107
            if not self.pause:                                                   # :111-118
                self.new_entry_counter += 1
                self.mem[self.mem_counter] = col
                self.mem_counter += 1
                if self.mem_counter == self.memsize_blocks:
                    self.mem_counter = 0
        if self.in_counter == 2 * n:                                             # :121-131
            self.in_counter = n
            self.indatamem[:, :n] = self.indatamem[:, n:].copy()
        return 0

    # --- getMem, Spectrogram.cpp:295-331 ---
    def get_mem(self, mem: np.ndarray):
        """mem is the caller's [W][H] float32 buffer, updated in place; returns (newVals, pos)."""
        W = self.memsize_blocks
        if mem.shape[0] != W:
            return -1, None                                                      # :297-298
        nec = self.new_entry_counter
        if nec >= W:                                                             # :300-304
            mem[:, :] = self.mem
        else:
            start = self.mem_counter - nec                                       # :307
            if start >= 0:
                mem[start:self.mem_counter] = self.mem[start:self.mem_counter]   # :310-311
            else:
                mem[:self.mem_counter] = self.mem[:self.mem_counter]             # :315-316
                mem[W + start:] = self.mem[W + start:]                           # :318-319
        self.new_entry_counter = 0                                               # :328
        return nec, self.mem_counter                                             # :327-330


def stft_db_reference(samples: np.ndarray, n: int, hop: int, feedblocks: int, win: np.ndarray,
                      mode: int = MIX_ABSMEAN, power_scale: float = 1.0, per_channel: bool = False,
                      return_power: bool = False) -> np.ndarray:
    """Closed form of what K calls of processSynchronBlock emit for a fresh engine (SURVEY section 3.1 'frame
    timeline'): frame j = k*feedblocks+bb starts at stream position (k-1)*N + bb*hop, positions < 0 read 0.
    samples: [C][K*N] float32.  Returns [K*feedblocks][N/2+1] dB (or [C][...] when per_channel)."""
    samples = np.asarray(samples, dtype=f32)
    C, total = samples.shape
    K = total // n
    padded = np.concatenate([np.zeros((C, n), dtype=f32), samples[:, :K * n]], axis=1)
    starts = np.array([k * n + bb * hop for k in range(K) for bb in range(feedblocks)], dtype=np.int64)
    idx = starts[:, None] + np.arange(n)[None, :]
    frames = (padded[:, idx] * win[None, None, :]).astype(f32)                  # [C][F][N]
    pw64 = power_spectrum_f64(frames, power_scale)
    if return_power:
        return pw64
    power = pw64.astype(f32)
    if per_channel:
        return to_db(power)
    return to_db(mix_channels(power, mode))


# ----------------------------------------------------------------------------------------------
# a9 / a10 colour map: CColorpalette.h:32-47, CColorpalette.cpp:39-54, :106-339
# ----------------------------------------------------------------------------------------------
_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "colormap_ref.json")
_base_tables = {}


def _quantised_base_table(scheme: int) -> np.ndarray:
    """int(cm_xxx[i][c]*255) packed 0xRRGGBB for i in 0..255 (CColorpalette.cpp:261-264, :279-282).
    ColormapData.h's double tables are reference source and are NOT copied into this repo; the 256 packed
    8-bit entries below are golden OUTPUT of the reference build (tests/golden/colormap_ref.json, made by
    oracle/gen_golden.py) -- at NrOfColors == 256 the table index equals kk, so the LUT is the base table."""
    if scheme not in _base_tables:
        with open(_GOLDEN) as fh:
            g = json.load(fh)
        _base_tables[scheme] = np.array(g["lut256"][str(scheme)], dtype=np.int64)
    return _base_tables[scheme]


def _itrunc(x) -> int:
    return int(x)  # C++ float/double -> int truncates toward zero, like Python's int()


def compute_colors(n_colors: int, scheme: int) -> np.ndarray:
    """CColorPalette::ComputeColors (CColorpalette.cpp:106-339), m_InvertScheme == 0 (never set)."""
    N = int(n_colors)
    lut = np.zeros(N, dtype=np.int64)
    half = N // 2
    for kk in range(N):
        fk = f32(kk)
        if scheme == CM_MONO:                                                   # :112-132
            col = 0 if kk <= half else 0xFFFFFF
        elif scheme == CM_BW:                                                   # :134-147
            v = _itrunc(f32(f32(255.0) * fk) / f32(N))
            col = (v << 16) | (v << 8) | v
        elif scheme == CM_RAINBOW:                                              # :148-215
            slope = f32(4.0) / f32(N)
            if kk < N // 8:
                b = _itrunc(float(f32(255.0)) * (float(f32(fk * slope)) + 0.5)); g = 0; r = 0
            elif kk < 3 * N // 8:                                               # two identical branches :164-178
                b = 255; g = _itrunc(f32(f32(f32(255.0) * f32(kk - N // 8)) * slope)); r = 0
            elif kk < 5 * N // 8:                                               # two identical branches :179-190
                t = f32(f32(kk - 3 * N // 8) * slope)
                b = _itrunc(f32(f32(255.0) * f32(f32(1.0) - t))); g = 255
                r = _itrunc(f32(f32(f32(255.0) * f32(kk - 3 * N // 8)) * slope))
            elif kk < 7 * N // 8:                                               # :191-204
                b = 0; g = _itrunc(f32(f32(255.0) * f32(f32(1.0) - f32(f32(kk - 5 * N // 8) * slope)))); r = 255
            else:                                                               # :205-211
                b = 0; g = 0; r = _itrunc(f32(f32(255.0) * f32(f32(1.0) - f32(f32(kk - 7 * N // 8) * slope))))
            col = (r << 16) | (g << 8) | b
        elif scheme == CM_HOT:                                                  # :216-252
            s1 = f32(8.0) / f32(3 * N)
            s2 = f32(8.0) / f32(2 * N)
            if kk < 3 * N // 8:
                b = 0; g = 0; r = _itrunc(f32(f32(255.0) * f32(fk * s1)))
            elif kk < 6 * N // 8:
                b = 0; g = _itrunc(f32(f32(f32(255.0) * f32(kk - 3 * N // 8)) * s1)); r = 255
            else:
                b = _itrunc(f32(f32(f32(255.0) * f32(kk - 6 * N // 8)) * s2)); g = 255; r = 255
            col = (r << 16) | (g << 8) | b
        elif scheme in (CM_VIRIDIS, CM_PLASMA):                                 # :254-289
            index = _itrunc(f32(f32(fk / f32(N)) * f32(256)))
            col = int(_quantised_base_table(scheme)[index])
        elif scheme == CM_JADE:                                                 # :290-335
            rs, rm, re = f32(0.3529), f32(0.89019), f32(0.95)
            gs, gm, ge = f32(0.372549), f32(0.023529), f32(0.95)
            bs, bm, be = f32(0.33725), f32(0.074509), f32(0.95)
            mix = 2 * N // 4
            if kk < mix:
                t = f32(fk / f32(mix))
                b = _itrunc(f32(f32(255) * f32(f32(t * f32(bm - bs)) + bs)))
                g = _itrunc(f32(f32(255) * f32(f32(t * f32(gm - gs)) + gs)))
                r = _itrunc(f32(f32(255) * f32(f32(t * f32(rm - rs)) + rs)))
            else:
                t = f32(f32(kk - mix) / f32(mix))
                b = _itrunc(f32(f32(255) * f32(f32(t * f32(be - bm)) + bm)))
                g = _itrunc(f32(f32(255) * f32(f32(t * f32(ge - gm)) + gm)))
                r = _itrunc(f32(f32(255) * f32(f32(t * f32(re - rm)) + rm)))
            col = (r << 16) | (g << 8) | b
        else:
            raise ValueError("unknown colour scheme")
        lut[kk] = col
    return lut.astype(np.int32)


class OracleColorPalette:
    """CColorPalette (CColorpalette.h:6-61)."""

    def __init__(self, n_colors: int = 2, scheme: int = CM_MONO):
        self.n = int(n_colors)
        self.scheme = int(scheme)
        self.vmin = f32(0.0)
        self.vmax = f32(1.0)
        self.mult = f32(self.n) / f32(self.vmax - self.vmin)                    # CColorpalette.cpp:10
        self.lut = compute_colors(self.n, self.scheme)

    def set_value_range(self, lo, hi):                                          # CColorpalette.cpp:39-54
        lo, hi = f32(lo), f32(hi)
        if hi >= lo:
            self.vmin, self.vmax = lo, hi
        else:
            self.vmin, self.vmax = hi, lo
        if self.vmax == self.vmin:
            self.vmin = f32(0.99 * float(self.vmax))                            # double product stored to float
        self.mult = f32(self.n) / f32(self.vmax - self.vmin)

    def set_color_scheme(self, scheme):                                         # CColorpalette.cpp:62-66
        self.scheme = int(scheme)
        self.lut = compute_colors(self.n, self.scheme)

    def index(self, values: np.ndarray) -> np.ndarray:
        """The index getRGBColor computes (CColorpalette.h:34-45), vectorised, float32 arithmetic."""
        v = np.asarray(values, dtype=f32).copy()
        top = f32(self.vmax * f32(0.9999))
        v = np.where(v >= self.vmax, top, v)                                    # :34-35
        v = np.where(v < self.vmin, self.vmin, v)                               # :37-38
        idx = ((v - self.vmin).astype(f32) * self.mult).astype(f32).astype(np.int64)  # :40 truncation
        return np.where(idx < self.n, idx, self.n - 1).astype(np.int32)         # :42-45

    def get_rgb_color(self, values: np.ndarray) -> np.ndarray:
        return self.lut[self.index(values)]


# ----------------------------------------------------------------------------------------------
# a11 colour loop: SpectrogramComponent::timerCallback, Spectrogram.cpp:590-731
# ----------------------------------------------------------------------------------------------
class OracleDisplay:
    """The image-producing half of timerCallback.  Image is uint32 ARGB [H][W] (juce::Colour(uint32))."""

    def __init__(self, spectrogram: OracleSpectrogram, n_colors: int = 256, scheme: int = CM_JADE):
        self.spec = spectrogram
        self.palette = OracleColorPalette(n_colors, scheme)                     # Spectrogram.cpp:337
        self.palette.set_value_range(G_MIN_COLOR_VAL, G_MAX_COLOR_VAL)          # :342
        self.W = 1; self.H = 1
        self.img = np.zeros((1, 1), dtype=np.uint32)
        self.displaymem = np.zeros((1, 1), dtype=f32)
        self.recompute_all = True
        self.running = True                                                     # m_isRunningDisplay

    def timer_callback(self, min_color=G_MIN_COLOR_VAL, max_color=G_MAX_COLOR_VAL):
        W = self.spec.get_memory_size(); H = self.spec.get_spectrum_size()
        if W != self.W or H != self.H:                                          # :595-605
            self.W, self.H = W, H
            self.img = np.zeros((H, W), dtype=np.uint32)    # rescaled() content is overwritten below
            self.displaymem = np.zeros((W, H), dtype=f32)
        new_vals, pos = self.spec.get_mem(self.displaymem)                      # :607-608
        if new_vals > self.W:
            self.recompute_all = True                                           # :610-613
        self.palette.set_value_range(min_color, max_color)                      # :617
        colours = lambda col: (self.palette.get_rgb_color(self.displaymem[col]).astype(np.int64)
                               | 0xFF000000).astype(np.uint32)                  # :636-637
        if self.recompute_all:                                                  # :623-657
            self.recompute_all = False
            newwstart = W - pos
            for ww in range(W):
                neww = ww + newwstart
                if neww >= W:
                    neww -= W
                x = neww if self.running else ww
                self.img[::-1, x] = colours(ww)                                 # y = H-1-hh
            if not self.running:
                self.img[:, pos] = JUCE_RED_ARGB                                # :650-656
        else:
            startread = pos - new_vals                                          # :661
            if self.running:                                                    # :663-683
                if new_vals > 0:
                    self.img[:, :W - new_vals] = self.img[:, new_vals:].copy()  # moveImageSection :665
                for ww in range(W - new_vals, W):
                    readpos = W + startread if startread < 0 else startread
                    self.img[::-1, ww] = colours(readpos)
                    startread += 1
            else:                                                               # :684-721
                for _ in range(new_vals):
                    readpos = W + startread if startread < 0 else startread
                    self.img[::-1, readpos] = colours(readpos)
                    startread += 1
                drawwidth = 1 + (1 if H < 2048 else 0) + (2 if H < 1024 else 0)
                for dd in range(drawwidth):
                    drawpos = pos + dd
                    if drawpos == W:
                        drawpos -= W
                    if drawpos < W:   # the reference would write out of bounds for pos+dd > W; never reached with W>=4
                        self.img[:, drawpos] = JUCE_RED_ARGB
        return new_vals, pos


def display_freq_rows(fs, height: int, min_freq, max_freq):
    """The sub-rectangle selection of SpectrogramComponent::paint (Spectrogram.cpp:441-459)."""
    fs, mn, mx = f32(fs), f32(min_freq), f32(max_freq)
    if float(mn) >= float(fs) * 0.5:
        mn = f32(0.9 * float(fs) * 0.5)                                          # :444-445
    if float(mx) >= float(fs) * 0.5:
        mx = f32(float(fs) * 0.5)                                                # :446-447
    if mn >= mx:
        mn = f32(0.9 * float(mx))                                                # :451
    lo = 2.0 * float(mn) / float(fs) * height
    hi = 2.0 * float(mx) / float(fs) * height
    start, end = int(lo + 0.5), int(hi + 0.5)                                    # :455-456
    return start, end, int(hi - lo + 0.5), height - end                          # :457, :459


def render_all(db_ring: np.ndarray, pos: int, palette: OracleColorPalette, running: bool = True) -> np.ndarray:
    """Closed form of the recompute-all branch (Spectrogram.cpp:623-657) for a [W][H] dB ring."""
    W, H = db_ring.shape
    rgb = (palette.get_rgb_color(db_ring).astype(np.int64) | 0xFF000000).astype(np.uint32)   # [W][H]
    img = np.zeros((H, W), dtype=np.uint32)
    xs = (np.arange(W) + (W - pos)) % W if running else np.arange(W)
    img[::-1, :][:, xs] = rgb.T
    if not running:
        img[:, pos] = JUCE_RED_ARGB
    return img


# ----------------------------------------------------------------------------------------------
# synthetic inputs, SURVEY section 8d
# ----------------------------------------------------------------------------------------------
def synth_audio(channels: int, n_samples: int, fs: float = 48000.0, seed: int = 1234, kind: str = "mix") -> np.ndarray:
    """x_c[n] = 0.5 sin(2 pi f_c n / fs) + 0.1 u[n], f_c = 220*2^(c/12), u ~ U(-1,1) from default_rng(seed+c)."""
    out = np.zeros((channels, n_samples), dtype=f32)
    t = np.arange(n_samples, dtype=np.float64)
    for c in range(channels):
        rng = np.random.default_rng(seed + c)
        u = rng.uniform(-1.0, 1.0, n_samples)
        if kind == "mix":
            fc = 220.0 * 2.0 ** (c / 12.0)
            out[c] = (0.5 * np.sin(2.0 * np.pi * fc * t / fs) + 0.1 * u).astype(f32)
        elif kind == "noise":
            out[c] = u.astype(f32)
        elif kind == "silence":
            pass
        else:
            raise ValueError(kind)
    return out


def fnv1a32(data: bytes) -> int:
    h = 0x811C9DC5
    for b in data:
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h
