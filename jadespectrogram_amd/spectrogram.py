"""Python mirror of the reference's host interface for the hot path, on top of the C-ABI (include/jsg.h).

    Spectrogram          <-> class Spectrogram          (reference Spectrogram.h:81-169)   method names kept
    CColorPalette        <-> class CColorPalette        (reference CColorpalette.h:6-61)
    SpectrogramDisplay   <-> the colour half of SpectrogramComponent::timerCallback (Spectrogram.cpp:590-731)
    Plan / stft_db / colormap : the stateless device ops on torch tensors that live in HBM (used by bench.py)

Everything computes on the GPU through libjsg.so; numpy/torch are used for buffers only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import JsgError, check, lib


# --------------------------------------------------------------------------------------------------
# host-side precompute (same arithmetic as the reference, done in the library)
# --------------------------------------------------------------------------------------------------
def feed_samples(feed_percent: float, fftsize: int) -> int:
    return check(lib().jsg_feed_samples(feed_percent, fftsize))


def memsize_blocks(memsize_s: float, fs: float, hop: int) -> int:
    return check(lib().jsg_memsize_blocks(memsize_s, fs, hop))


def next_power_of_2(ms: float, fs: float) -> int:
    return check(lib().jsg_next_power_of_2(ms, fs))


def window(kind: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    check(lib().jsg_window_build(kind, n, out.ctypes.data))
    return out


def colormap_lut(n_colors: int, scheme: int) -> np.ndarray:
    out = np.zeros(n_colors, dtype=np.int32)
    check(lib().jsg_colormap_build(n_colors, scheme, out.ctypes.data))
    return out


def colormap_range(n_colors: int, lo: float, hi: float):
    a, b, m = C.c_float(), C.c_float(), C.c_float()
    check(lib().jsg_colormap_range(n_colors, lo, hi, C.byref(a), C.byref(b), C.byref(m)))
    return np.float32(a.value), np.float32(b.value), np.float32(m.value)


# --------------------------------------------------------------------------------------------------
# class Spectrogram
# --------------------------------------------------------------------------------------------------
class Spectrogram:
    """Drop-in mirror of the reference's `Spectrogram` (engine half).  The channel count is explicit."""

    ChannelMixMode = type("ChannelMixMode", (), dict(AbsMean=0, Max=1, Min=2, Left=3, Right=4, PerChannel=100))
    Windows = type("Windows", (), dict(Rect=0, Hann=1, Hamming=2, BlackmanHarris=3, FlatTop=4, HannPoisson=5))
    FeedPercentage = type("FeedPercentage", (), dict(perc100=0, perc50=1, perc25=2, perc10=3))

    def __init__(self, channels: int = 2):
        self._h = C.c_void_p()
        check(lib().jsg_create(C.byref(self._h), int(channels)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().jsg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _c(self, rc):
        return check(rc, self._h)

    # setters (Spectrogram.h:116-123)
    def setSamplerate(self, fs): self._c(lib().jsg_set_samplerate(self._h, fs))
    def setchannels(self, c): self._c(lib().jsg_set_channels(self._h, int(c)))
    def setFFTSize(self, n): self._c(lib().jsg_set_fft_size(self._h, int(n)))
    def setclosestFFTSize_ms(self, ms): self._c(lib().jsg_set_closest_fft_size_ms(self._h, ms))
    def setmemoryTime_s(self, s): self._c(lib().jsg_set_memory_time_s(self._h, s))
    def setfeed_percent(self, feed): self._c(lib().jsg_set_feed_percent(self._h, int(feed)))
    def setfeed_percent_ext(self, pct): self._c(lib().jsg_set_feed_percent_ext(self._h, pct))
    def setPauseMode(self, mode): self._c(lib().jsg_set_pause_mode(self._h, int(bool(mode))))
    def setWindow(self, win): self._c(lib().jsg_set_window(self._h, int(win)))
    def setMixMode(self, mode): self._c(lib().jsg_set_mix_mode(self._h, int(mode)))
    def setPowerScale(self, s): self._c(lib().jsg_set_power_scale(self._h, s))
    def setExactLog(self, on): self._c(lib().jsg_set_exact_log(self._h, int(bool(on))))

    def setWindowTable(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32)
        self._c(lib().jsg_set_window_table(self._h, w.ctypes.data, w.size))

    def getnextpowerof2(self, ms): return next_power_of_2(ms, self.getSamplerate())
    def getSpectrumSize(self): return lib().jsg_get_spectrum_size(self._h)
    def getMemorySize(self): return lib().jsg_get_memory_size(self._h)
    def getSamplerate(self): return lib().jsg_get_samplerate(self._h)
    def getFFTSize(self): return lib().jsg_get_fft_size(self._h)
    def getFeedSamples(self): return lib().jsg_get_feed_samples(self._h)
    def getFeedBlocks(self): return lib().jsg_get_feedblocks(self._h)
    def getChannels(self): return lib().jsg_get_channels(self._h)

    def getWindow(self):
        w = np.zeros(self.getFFTSize(), dtype=np.float32)
        self._c(lib().jsg_get_window(self._h, w.ctypes.data, w.size))
        return w

    def processSynchronBlock(self, data, midi=None, realtime: bool = False, timeout_ms: int = 10000) -> int:
        """data: [channels][fft size] float32 (reference: vector<vector<float>>&, Spectrogram.cpp:37).

        A Python caller is not an audio thread, so the default is the LOSSLESS entry point (jsg_process_block_wait): like the reference,
        no block is ever lost -- when the engine's 64-slot ring is full the call waits for the worker.  A block that still cannot be
        queued (geometry change in progress, timeout) raises instead of vanishing.  realtime=True: the wait-free call of a live host
        (jsg_process_block), which DROPS when the ring is full and returns 1; droppedBlocks() counts them."""
        data = np.ascontiguousarray(data, dtype=np.float32)
        n, ch = self.getFFTSize(), self.getChannels()
        if data.shape != (ch, n):
            raise JsgError(capi.JSG_ERR_SIZE_MISMATCH, f"block must be [{ch}][{n}], got {data.shape}")
        ptrs = (C.c_void_p * ch)(*[data[c].ctypes.data for c in range(ch)])
        if realtime:
            return self._c(lib().jsg_process_block(self._h, ptrs))
        rc = self._c(lib().jsg_process_block_wait(self._h, ptrs, ch, n, int(timeout_ms)))
        if rc == 1:
            raise JsgError(1, "processSynchronBlock: the block was dropped (geometry change in progress, or the ring stayed full "
                              f"for {timeout_ms} ms); pass realtime=True to get the wait-free, lossy call of a live host")
        return rc

    def droppedBlocks(self) -> int:
        return int(lib().jsg_get_dropped_blocks(self._h))

    def processBlocks(self, samples) -> int:
        """samples: [channels][K * fft size]; the same as K processSynchronBlock calls, one kernel launch."""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        n, ch = self.getFFTSize(), self.getChannels()
        if samples.ndim != 2 or samples.shape[0] != ch or samples.shape[1] % n:
            raise JsgError(capi.JSG_ERR_SIZE_MISMATCH, "samples must be [channels][K*fftsize]")
        return self._c(lib().jsg_process_blocks(self._h, samples.ctypes.data, samples.shape[1], samples.shape[1] // n))

    def processBlocksDevice(self, d_samples) -> int:
        """d_samples: torch float32 CUDA tensor [channels][K * fft size] already resident in HBM."""
        n, ch = self.getFFTSize(), self.getChannels()
        assert d_samples.is_cuda and d_samples.dtype.is_floating_point and d_samples.element_size() == 4
        assert d_samples.dim() == 2 and d_samples.shape[0] == ch and d_samples.shape[1] % n == 0
        assert d_samples.stride(1) == 1
        return self._c(lib().jsg_process_blocks_device(self._h, d_samples.data_ptr(), d_samples.stride(0),
                                                       d_samples.shape[1] // n))

    def getMem(self, mem: np.ndarray):
        """mem: caller's [W][H] float32 array, updated in place.  Returns (newVals, pos); newVals == -1 on a
        size mismatch like the reference (Spectrogram.cpp:297-298)."""
        if mem.dtype != np.float32 or not mem.flags.c_contiguous or mem.ndim != 2 or mem.shape[1] != self.getSpectrumSize():
            return -1, None
        pos = C.c_int(0)
        rc = lib().jsg_get_mem(self._h, mem.ctypes.data, mem.shape[0], C.byref(pos))
        if rc == capi.JSG_ERR_SIZE_MISMATCH:
            return -1, None
        self._c(rc)
        return rc, pos.value

    def ring_device(self):
        p, pitch, w, pos = C.c_void_p(), C.c_int64(), C.c_int(), C.c_int()
        self._c(lib().jsg_ring_device(self._h, C.byref(p), C.byref(pitch), C.byref(w), C.byref(pos)))
        return p.value, pitch.value, w.value, pos.value

    def sync(self): self._c(lib().jsg_sync(self._h))


class SpectrogramDisplay:
    """timerCallback's colour loop on the engine's device-resident ring (Spectrogram.cpp:590-731)."""

    def __init__(self, spectrogram: Spectrogram, n_colors: int = 256, scheme: int = capi.CM_JADE):
        self.spec = spectrogram
        spectrogram._c(lib().jsg_display_set_colormap(spectrogram._h, n_colors, scheme))

    def setColorSceme(self, scheme, n_colors: int = 256):
        self.spec._c(lib().jsg_display_set_colormap(self.spec._h, n_colors, int(scheme)))

    def setRunning(self, running: bool):
        self.spec._c(lib().jsg_display_set_running(self.spec._h, int(bool(running))))

    def invalidate(self):
        self.spec._c(lib().jsg_display_invalidate(self.spec._h))

    def timerCallback(self, img: np.ndarray, min_color=-50.0, max_color=50.0):
        """img: [H][W] uint32 ARGB, updated in place.  Returns (newVals, pos)."""
        assert img.dtype == np.uint32 and img.ndim == 2 and img.strides[1] == 4
        nv, pos = C.c_int(), C.c_int()
        self.spec._c(lib().jsg_display_update(self.spec._h, min_color, max_color, img.ctypes.data, img.strides[0] // 4,
                                              C.byref(nv), C.byref(pos)))
        return nv.value, pos.value


    def timerCallbackTile(self, tile: np.ndarray, min_color=-50.0, max_color=50.0):
        """Incremental tick: only the new columns, as a [H][max_cols] tile (oldest first).  Returns
        (need_full, newVals, pos); need_full=True means call timerCallback() for the whole image instead."""
        assert tile.dtype == np.uint32 and tile.ndim == 2 and tile.strides[1] == 4
        nv, pos = C.c_int(), C.c_int()
        rc = self.spec._c(lib().jsg_display_update_tile(self.spec._h, min_color, max_color, tile.ctypes.data,
                                                        tile.strides[0] // 4, tile.shape[1], C.byref(nv), C.byref(pos)))
        return rc == 1, nv.value, pos.value


def display_freq_rows(fs: float, height: int, min_freq: float, max_freq: float):
    """paint()'s frequency window (reference Spectrogram.cpp:441-459): (startPixel, endPixel, heightInterval, hStart)."""
    a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib().jsg_display_freq_rows(fs, height, min_freq, max_freq, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value


class CColorPalette:
    """Host mirror of the reference's CColorPalette: table + range on the host, bulk mapping on the GPU."""
    kMono, kBW, kHot, kRainbow, kViridis, kPlasma, kJade = range(7)

    def __init__(self, NrOfColors: int = 2, ColorScheme: int = 0):
        self.m_NrOfColors, self.m_ColorScheme = int(NrOfColors), int(ColorScheme)
        self.m_Color = colormap_lut(self.m_NrOfColors, self.m_ColorScheme)
        self.m_Min, self.m_Max, self.m_AccessMult = colormap_range(self.m_NrOfColors, 0.0, 1.0)
        self.m_Min, self.m_Max = np.float32(0.0), np.float32(1.0)

    def setValueRange(self, Min, Max):
        self.m_Min, self.m_Max, self.m_AccessMult = colormap_range(self.m_NrOfColors, Min, Max)

    def setNrOfColors(self, n):
        self.m_NrOfColors = int(n)
        self.m_AccessMult = np.float32(self.m_NrOfColors) / np.float32(self.m_Max - self.m_Min)
        self.m_Color = colormap_lut(self.m_NrOfColors, self.m_ColorScheme)

    def setColorSceme(self, scheme):
        self.m_ColorScheme = int(scheme)
        self.m_Color = colormap_lut(self.m_NrOfColors, self.m_ColorScheme)


# --------------------------------------------------------------------------------------------------
# stateless device ops on torch tensors
# --------------------------------------------------------------------------------------------------
class Plan:
    def __init__(self, n: int, window_table: np.ndarray, power_scale: float = 1.0):
        w = np.ascontiguousarray(window_table, dtype=np.float32)
        assert w.size == n
        self._p = C.c_void_p()
        check(lib().jsg_plan_create(C.byref(self._p), int(n), w.ctypes.data, power_scale))
        self.n = int(n)

    def close(self):
        if getattr(self, "_p", None) is not None and self._p:
            lib().jsg_plan_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StftLaunch:
    """A prepared launch: the argument block is filled once, `launch(stream)` is then a single FFI call
    (the per-call Python work of stft_db() is ~2 us, comparable to a short kernel)."""

    def __init__(self, plan: Plan, d_in, hop: int, n_frames: int, d_out, **kw):
        self._plan = plan
        self._keep = (d_in, d_out)
        self._args = _stft_args(plan, d_in, hop, n_frames, d_out, **kw)
        self._ref = C.byref(self._args)
        self._fn = lib().jsg_stft_db_launch

    def launch(self, stream: int):
        rc = self._fn(self._plan._p, self._ref, stream)
        if rc < 0:
            check(rc)


def stft_db(plan: Plan, d_in, hop: int, n_frames: int, d_out, *, feedblocks: int | None = None, mix_mode: int = 0,
            first_frame: int = 0, ring_pos: int = 0, linear_out: bool = False, blocks_per_cu: int = 0,
            stream: int | None = None, plan_select: int = 0, exact_log: bool = False, d_tail=None):
    """Enqueue one fused STFT->dB launch.  d_in: torch CUDA float32 [C][samples]; d_out: [W][pitch] (or
    [C][W][pitch] with mix_mode PER_CHANNEL).  Frame j starts at sample (j//feedblocks)*n + (j%feedblocks)*hop.
    d_tail: see _stft_args (jsg_stft_args.out_tail: bin n/2 in a dense plane, columns of n/2 floats)."""
    import torch
    a = _stft_args(plan, d_in, hop, n_frames, d_out, feedblocks=feedblocks, mix_mode=mix_mode, first_frame=first_frame,
                   ring_pos=ring_pos, linear_out=linear_out, blocks_per_cu=blocks_per_cu, plan_select=plan_select, exact_log=exact_log, d_tail=d_tail)
    if stream is None:
        stream = torch.cuda.current_stream(d_in.device).cuda_stream
    check(lib().jsg_stft_db_launch(plan._p, C.byref(a), C.c_void_p(stream)))


def columns_from_tail_layout(d_db, d_tail, d_dst, *, n: int | None = None, stream: int | None = None):
    """jsg_columns_from_tail_layout_launch: d_db [W][pitch >= n/2] + d_tail [W] (the tail-plane layout of stft_db(..., d_tail=)) -> d_dst [W][>= n/2+1],
    the reference's dense column shape (what getMem hands out).  `n` = the FFT size (a Plan's .n); without it the rows of d_db must be
    exactly n/2 floats wide (the shape stft_db's tail layout is normally given) -- the height is never guessed from padded rows."""
    import torch
    assert d_db.is_cuda and d_tail.is_cuda and d_dst.is_cuda and d_db.dim() == 2 and d_dst.dim() == 2 and d_db.stride(1) == 1 and d_dst.stride(1) == 1
    W = d_db.shape[0]
    assert d_tail.numel() == W and d_tail.is_contiguous() and d_dst.shape[0] == W
    if n is None:
        n = 2 * d_db.shape[1]
        if n < 2 or n & (n - 1):
            raise JsgError(capi.JSG_ERR_INVALID, f"columns_from_tail_layout: rows of {d_db.shape[1]} floats are not n/2 of a power-of-two FFT size; pass n=")
    H = n // 2 + 1
    if d_db.shape[1] < H - 1 or d_dst.shape[1] < H:
        raise JsgError(capi.JSG_ERR_INVALID, f"columns_from_tail_layout: n = {n} needs source rows of >= {H - 1} and destination rows of >= {H} floats")
    if stream is None:
        stream = torch.cuda.current_stream(d_db.device).cuda_stream
    check(lib().jsg_columns_from_tail_layout_launch(d_db.data_ptr(), d_db.stride(0), d_tail.data_ptr(), W, H, d_dst.data_ptr(), d_dst.stride(0), C.c_void_p(stream)))


def stft_db_batches(plan: Plan, batches, hop: int, n_frames: int, *, stream: int | None = None, **kw):
    """jsg_stft_db_launch_batches: `batches` = [(d_in, d_out), ...] independent launches of the same geometry, stream-ordered with
    respect to `stream` like one launch but overlapped on the library's own working streams (include/jsg.h)."""
    import torch
    arr = (capi.StftArgs * len(batches))()
    for i, (d_in, d_out) in enumerate(batches):
        a = _stft_args(plan, d_in, hop, n_frames, d_out, **kw)
        C.memmove(C.byref(arr, i * C.sizeof(capi.StftArgs)), C.byref(a), C.sizeof(capi.StftArgs))
    if stream is None:
        stream = torch.cuda.current_stream(batches[0][0].device).cuda_stream
    check(lib().jsg_stft_db_launch_batches(plan._p, arr, len(batches), C.c_void_p(stream)))


def _strided_args(plan: Plan, d_in, hop: int, n_frames: int, d_out, **kw):
    """d_in float32 [k][channels][samples], d_out [k][W][pitch] (per-channel mode: [k][channels][W][pitch]): batch b = d_in[b] -> d_out[b]."""
    assert d_in.dim() == 3 and d_in.stride(2) == 1 and d_out.shape[0] == d_in.shape[0] and d_out.stride(-1) == 1
    a = _stft_args(plan, d_in[0], hop, n_frames, d_out[0], **kw)
    return a, int(d_in.shape[0]), int(d_in.stride(0)), int(d_out.stride(0))


def stft_db_strided(plan: Plan, d_in, hop: int, n_frames: int, d_out, *, stream: int | None = None, **kw):
    """jsg_stft_db_launch_strided: k independent batches of one geometry in ONE kernel launch on one stream (include/jsg.h)."""
    import torch
    a, k, s_in, s_out = _strided_args(plan, d_in, hop, n_frames, d_out, **kw)
    if stream is None:
        stream = torch.cuda.current_stream(d_in.device).cuda_stream
    check(lib().jsg_stft_db_launch_strided(plan._p, C.byref(a), k, s_in, s_out, C.c_void_p(stream)))


def stft_db_strided_kernel_name(plan: Plan, d_in, hop: int, n_frames: int, d_out, **kw) -> str:
    """The kernel a strided launch takes: as stft_kernel_name, judged by the frames of the whole launch."""
    a, k, s_in, _ = _strided_args(plan, d_in, hop, n_frames, d_out, **kw)
    buf = C.create_string_buffer(32)
    check(lib().jsg_stft_db_strided_kernel_name(plan._p, C.byref(a), k, s_in, buf, 32))
    return buf.value.decode()


def stft_kernel_name(plan: Plan, d_in, hop: int, n_frames: int, d_out, **kw) -> str:
    """The kernel configuration jsg_stft_db_launch picks for this launch ("Cfg1024", "Cfg2048B", ...)."""
    a = _stft_args(plan, d_in, hop, n_frames, d_out, **kw)
    buf = C.create_string_buffer(32)
    check(lib().jsg_stft_kernel_name(plan._p, C.byref(a), buf, 32))
    return buf.value.decode()


def _stft_args(plan: Plan, d_in, hop: int, n_frames: int, d_out, *, feedblocks: int | None = None, mix_mode: int = 0,
               first_frame: int = 0, ring_pos: int = 0, linear_out: bool = False, blocks_per_cu: int = 0, plan_select: int = 0,
               exact_log: bool = False, d_tail=None):
    """d_tail (jsg_stft_args.out_tail): float32 CUDA tensor of rows x W floats (rows = 1, or channels in per-channel mode, times the batches
    of a strided launch), contiguous -- bin n/2 of every column goes there and a column of d_out is then n/2 floats."""
    import torch
    assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.dim() == 2 and d_in.stride(1) == 1
    assert d_out.is_cuda and d_out.dtype == torch.float32 and d_out.stride(-1) == 1
    H = plan.n // 2 + 1 - (1 if d_tail is not None else 0)
    if d_out.shape[-1] < H:
        raise JsgError(capi.JSG_ERR_INVALID, f"output rows hold {d_out.shape[-1]} floats, a column needs {H}")
    if mix_mode == capi.MIX_PER_CHANNEL and (d_out.dim() != 3 or d_out.shape[0] != d_in.shape[0]):
        raise JsgError(capi.JSG_ERR_INVALID, "per-channel mode needs an output of [channels][W][pitch]")
    a = capi.StftArgs()
    a.in_ = d_in.data_ptr()
    a.in_pitch = d_in.stride(0) if d_in.shape[0] > 1 else d_in.shape[1]
    a.in_samples = d_in.shape[1]      # the launcher refuses frames that would read past the rows
    a.channels = d_in.shape[0]
    a.hop = hop
    a.feedblocks = feedblocks if feedblocks is not None else max(1, plan.n // hop)
    a.mix_mode = mix_mode
    a.first_frame = first_frame
    a.n_frames = n_frames
    a.out_db = d_out.data_ptr()
    a.out_pitch = d_out.stride(-2)
    a.out_channel_pitch = d_out.stride(0) if d_out.dim() == 3 else 0
    a.ring_width = d_out.shape[-2]
    a.ring_pos = ring_pos
    a.linear_out = int(bool(linear_out))
    a.blocks_per_cu = int(blocks_per_cu)
    a.exact_log = int(bool(exact_log))     # dB by the shared float32 routine (bit-reproducible on a CPU) instead of v_log_f32
    a.plan_select = int(plan_select)       # 0 automatic, 1 small-workgroup kernel, 2 "B" kernel (2048 / 4096 points), 3 pair plan (2048 points, even channel
                                           # counts, sum-type mixes); 1024 points: 2 = the two-stage kernel Cfg1024B (include/jsg.h)
    if d_tail is not None:
        assert d_tail.is_cuda and d_tail.dtype == torch.float32 and d_tail.is_contiguous() and d_tail.shape[-1] == d_out.shape[-2]
        a.out_tail = d_tail.data_ptr()
    return a


def colormap(d_db, d_lut, lo: float, hi: float, *, d_argb=None, d_index=None, col_first: int = 0, n_cols: int | None = None,
             x_first: int = 0, height: int | None = None, stream: int | None = None):
    """Enqueue the colour loop: d_db [W][pitch] float32 CUDA -> d_argb [H][Wimg] int32/uint32 and/or d_index uint8."""
    import torch
    a = capi.ColormapArgs()
    a.db = d_db.data_ptr()
    a.db_pitch = d_db.stride(0)
    a.ring_width = d_db.shape[0]
    a.height = height if height is not None else d_db.shape[1]
    a.col_first = col_first
    a.n_cols = n_cols if n_cols is not None else d_db.shape[0]
    a.x_first = x_first
    a.lut = d_lut.data_ptr()
    a.n_colors = d_lut.numel()
    a.vmin, a.vmax, a.access_mult = (float(v) for v in colormap_range(a.n_colors, lo, hi))
    if d_argb is not None:
        a.argb_out = d_argb.data_ptr()
        a.argb_pitch = d_argb.stride(0)
        a.x_wrap = d_argb.shape[1]
    if d_index is not None:
        a.index_out = d_index.data_ptr()
        a.index_pitch = d_index.stride(0)
        a.x_wrap = d_index.shape[1]
    if stream is None:
        stream = torch.cuda.current_stream(d_db.device).cuda_stream
    check(lib().jsg_colormap_launch(C.byref(a), C.c_void_p(stream)))


def _stft_image_args(plan: Plan, d_in, hop: int, n_frames: int, d_lut, lo: float, hi: float, d_argb, d_index_scratch, *,
                     feedblocks: int | None = None, mix_mode: int = 0, first_frame: int = 0, ring_pos: int = 0,
                     ring_width: int | None = None, x_first: int | None = None, plan_select: int = 0, exact_log: bool = False,
                     blocks_per_cu: int = 0):
    import torch
    assert d_argb.is_cuda and d_argb.element_size() == 4 and d_argb.dim() == 2 and d_argb.stride(1) == 1
    if d_index_scratch is not None:
        assert d_index_scratch.is_cuda and d_index_scratch.dtype == torch.uint8 and d_index_scratch.dim() == 2 and d_index_scratch.stride(1) == 1
    W = ring_width if ring_width is not None else (d_index_scratch.shape[0] if d_index_scratch is not None else n_frames)
    a = capi.StftImageArgs()
    H = plan.n // 2 + 1
    st = capi.StftArgs()
    st.in_ = d_in.data_ptr()
    st.in_pitch = d_in.stride(0) if d_in.shape[0] > 1 else d_in.shape[1]
    st.in_samples = d_in.shape[1]
    st.channels = d_in.shape[0]
    st.hop = hop
    st.feedblocks = feedblocks if feedblocks is not None else max(1, plan.n // hop)
    st.mix_mode = mix_mode
    st.first_frame = first_frame
    st.n_frames = n_frames
    st.ring_width = W
    st.ring_pos = ring_pos
    st.plan_select = int(plan_select)
    st.exact_log = int(bool(exact_log))
    st.blocks_per_cu = int(blocks_per_cu)
    a.stft = st
    c = capi.ColormapArgs()
    c.ring_width = W
    c.height = H
    c.col_first = ring_pos
    c.n_cols = n_frames
    c.x_first = ring_pos if x_first is None else x_first
    c.x_wrap = d_argb.shape[1]
    c.lut = d_lut.data_ptr()
    c.n_colors = d_lut.numel()
    c.vmin, c.vmax, c.access_mult = (float(v) for v in colormap_range(c.n_colors, lo, hi))
    c.argb_out = d_argb.data_ptr()
    c.argb_pitch = d_argb.stride(0)
    a.colour = c
    if d_index_scratch is not None:
        a.index_scratch = d_index_scratch.data_ptr()
        a.index_scratch_pitch = d_index_scratch.stride(0)
    return a


def stft_image(plan: Plan, d_in, hop: int, n_frames: int, d_lut, lo: float, hi: float, d_argb, d_index_scratch=None, *,
               stream: int | None = None, **kw):
    """Fused display path (jsg_stft_image_launch): STFT -> palette index -> ARGB rows of d_argb [n/2+1][Wimg]; no dB column is
    written.  One kernel where the plan's workgroups hold eight whole columns (1024 points; 4096 points when the launch takes
    the "B" kernel), else two kernels through d_index_scratch (uint8 [ring_width][pitch >= n/2+1]; stft_image_needs_scratch()).
    The image equals stft_db() + colormap() bit for bit."""
    import torch
    a = _stft_image_args(plan, d_in, hop, n_frames, d_lut, lo, hi, d_argb, d_index_scratch, **kw)
    if stream is None:
        stream = torch.cuda.current_stream(d_in.device).cuda_stream
    check(lib().jsg_stft_image_launch(plan._p, C.byref(a), C.c_void_p(stream)))


def stft_image_needs_scratch(plan: Plan, d_in, hop: int, n_frames: int, d_lut, lo: float, hi: float, d_argb, d_index_scratch=None, **kw) -> bool:
    """True when jsg_stft_image_launch runs as two kernels for this launch and therefore needs the index scratch."""
    a = _stft_image_args(plan, d_in, hop, n_frames, d_lut, lo, hi, d_argb, d_index_scratch, **kw)
    return bool(check(lib().jsg_stft_image_needs_scratch(plan._p, C.byref(a))))


def stft_image_strided(plan: Plan, d_in, hop: int, n_frames: int, d_lut, lo: float, hi: float, d_argb, d_index_scratch=None, *,
                       stream: int | None = None, **kw):
    """`k` images of one geometry from one call (jsg_stft_image_launch_strided): d_in float32 [k][channels][samples], d_argb
    int32 / uint32 [k][n/2+1][Wimg].  One kernel launch for all of them where the single-kernel form applies to the total size
    (stft_image_strided_needs_scratch() tells), else k launches in stream order."""
    import torch
    assert d_in.dim() == 3 and d_argb.dim() == 3 and d_in.shape[0] == d_argb.shape[0] and d_in.stride(2) == 1 and d_argb.stride(2) == 1
    assert d_argb.shape[1] >= plan.n // 2 + 1, "image rows: one per bin"
    assert d_in.shape[0] == 1 or d_in.stride(0) == 0 or d_in.stride(0) >= (d_in.shape[1] - 1) * d_in.stride(1) + d_in.shape[2], "images overlap in the input"
    a = _stft_image_args(plan, d_in[0], hop, n_frames, d_lut, lo, hi, d_argb[0], d_index_scratch, **kw)
    if stream is None:
        stream = torch.cuda.current_stream(d_in.device).cuda_stream
    check(lib().jsg_stft_image_launch_strided(plan._p, C.byref(a), int(d_in.shape[0]), int(d_in.stride(0)), int(d_argb.stride(0)),
                                              C.c_void_p(stream)))


def stft_image_strided_needs_scratch(plan: Plan, d_in, hop: int, n_frames: int, d_lut, lo: float, hi: float, d_argb, d_index_scratch=None,
                                     **kw) -> bool:
    a = _stft_image_args(plan, d_in[0], hop, n_frames, d_lut, lo, hi, d_argb[0], d_index_scratch, **kw)
    return bool(check(lib().jsg_stft_image_strided_needs_scratch(plan._p, C.byref(a), int(d_in.shape[0]))))


def db_from_power(d_power, d_out, divisor: float = 1.0, stream: int | None = None):
    """out = 10*log10(power/divisor + 1e-11f) elementwise on the GPU (finishes a cross-GPU AbsMean)."""
    import torch
    assert d_power.is_cuda and d_power.dtype == torch.float32 and d_power.is_contiguous()
    assert d_out.is_cuda and d_out.dtype == torch.float32 and d_out.is_contiguous() and d_out.numel() == d_power.numel()
    if stream is None:
        stream = torch.cuda.current_stream(d_power.device).cuda_stream
    check(lib().jsg_db_from_power_launch(d_power.data_ptr(), d_out.data_ptr(), d_power.numel(), divisor, C.c_void_p(stream)))
