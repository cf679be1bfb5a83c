"""ctypes binding of the C-ABI declared in include/jsg.h (libjsg.so: hand-written HIP for gfx950).

This is plumbing only.  There is NO CPU fallback: if the shared library is missing, loading fails loudly, and on a
machine without a usable MI355X the compute entry points return JSG_ERR_NO_DEVICE / JSG_ERR_HIP.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libjsg.so")

JSG_OK = 0
JSG_ERR_SIZE_MISMATCH = -1
JSG_ERR_INVALID = -2
JSG_ERR_UNSUPPORTED = -3
JSG_ERR_HIP = -4
JSG_ERR_NO_DEVICE = -5
JSG_ERR_NOMEM = -6

MIX_ABSMEAN, MIX_MAX, MIX_MIN, MIX_LEFT, MIX_RIGHT = range(5)
MIX_PER_CHANNEL = 100
MIX_SUM = 101
WIN_RECT, WIN_HANN, WIN_HAMMING, WIN_BLACKMANHARRIS, WIN_FLATTOP, WIN_HANNPOISSON = range(6)
FEED_100, FEED_50, FEED_25, FEED_10 = range(4)
CM_MONO, CM_BW, CM_HOT, CM_RAINBOW, CM_VIRIDIS, CM_PLASMA, CM_JADE = range(7)


class JsgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"jsg error {code}: {msg}")
        self.code = code


class StftArgs(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_pitch", C.c_int64), ("channels", C.c_int32), ("hop", C.c_int32),
                ("feedblocks", C.c_int32), ("mix_mode", C.c_int32), ("first_frame", C.c_int64),
                ("n_frames", C.c_int64), ("out_db", C.c_void_p), ("out_pitch", C.c_int64),
                ("out_channel_pitch", C.c_int64), ("ring_width", C.c_int32), ("ring_pos", C.c_int32),
                ("linear_out", C.c_int32), ("blocks_per_cu", C.c_int32), ("in_samples", C.c_int64),
                ("plan_select", C.c_int32), ("exact_log", C.c_int32), ("reserved0", C.c_int32), ("out_tail", C.c_void_p)]


class ColormapArgs(C.Structure):
    _fields_ = [("db", C.c_void_p), ("db_pitch", C.c_int64), ("ring_width", C.c_int32), ("height", C.c_int32),
                ("col_first", C.c_int32), ("n_cols", C.c_int32), ("x_first", C.c_int32), ("x_wrap", C.c_int32),
                ("lut", C.c_void_p), ("n_colors", C.c_int32), ("vmin", C.c_float), ("vmax", C.c_float),
                ("access_mult", C.c_float), ("argb_out", C.c_void_p), ("argb_pitch", C.c_int64),
                ("index_out", C.c_void_p), ("index_pitch", C.c_int64)]


class StftImageArgs(C.Structure):
    _fields_ = [("stft", StftArgs), ("colour", ColormapArgs), ("index_scratch", C.c_void_p), ("index_scratch_pitch", C.c_int64)]


# every symbol include/jsg.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "jsg_abi_version": (C.c_int, []),
    "jsg_device_count": (C.c_int, []),
    "jsg_feed_samples": (C.c_int, [C.c_float, C.c_int]),
    "jsg_memsize_blocks": (C.c_int, [C.c_float, C.c_float, C.c_int]),
    "jsg_next_power_of_2": (C.c_int, [C.c_float, C.c_float]),
    "jsg_window_build": (C.c_int, [C.c_int, C.c_int, _P]),
    "jsg_colormap_build": (C.c_int, [C.c_int, C.c_int, _P]),
    "jsg_colormap_range": (C.c_int, [C.c_int, C.c_float, C.c_float, _P, _P, _P]),
    "jsg_plan_create": (C.c_int, [C.POINTER(_P), C.c_int, _P, C.c_float]),
    "jsg_plan_destroy": (C.c_int, [_P]),
    "jsg_plan_fft_size": (C.c_int, [_P]),
    "jsg_stft_db_launch": (C.c_int, [_P, C.POINTER(StftArgs), _P]),
    "jsg_stft_db_launch_many": (C.c_int, [_P, C.POINTER(StftArgs), C.c_int, C.POINTER(_P), C.c_int]),
    "jsg_stft_db_launch_many_threads": (C.c_int, [_P, C.POINTER(StftArgs), C.c_int, C.POINTER(_P), C.c_int, C.c_int]),
    "jsg_stft_db_launch_batches": (C.c_int, [_P, C.POINTER(StftArgs), C.c_int, _P]),
    "jsg_stft_kernel_name": (C.c_int, [_P, C.POINTER(StftArgs), C.c_char_p, C.c_int]),
    "jsg_stft_db_launch_strided": (C.c_int, [_P, C.POINTER(StftArgs), C.c_int, C.c_int64, C.c_int64, _P]),
    "jsg_stft_db_strided_kernel_name": (C.c_int, [_P, C.POINTER(StftArgs), C.c_int, C.c_int64, C.c_char_p, C.c_int]),
    "jsg_calib_copy_launch": (C.c_int, [_P, _P, C.c_int64, _P]),
    "jsg_columns_from_tail_layout_launch": (C.c_int, [_P, C.c_int64, _P, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "jsg_colormap_launch": (C.c_int, [C.POINTER(ColormapArgs), _P]),
    "jsg_stft_image_launch": (C.c_int, [_P, C.POINTER(StftImageArgs), _P]),
    "jsg_stft_image_needs_scratch": (C.c_int, [_P, C.POINTER(StftImageArgs)]),
    "jsg_stft_image_launch_strided": (C.c_int, [_P, C.POINTER(StftImageArgs), C.c_int, C.c_int64, C.c_int64, _P]),
    "jsg_stft_image_strided_needs_scratch": (C.c_int, [_P, C.POINTER(StftImageArgs), C.c_int]),
    "jsg_db_from_power_launch": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "jsg_db_from_power_launch_ex": (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int, _P]),
    "jsg_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "jsg_create_on_device": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int]),
    "jsg_get_device": (C.c_int, [_P]),
    "jsg_create_sharded": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int]),
    "jsg_process_block_sharded": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "jsg_destroy_sharded": (C.c_int, [C.POINTER(_P), C.c_int]),
    "jsg_destroy": (C.c_int, [_P]),
    "jsg_last_error": (C.c_char_p, [_P]),
    "jsg_set_samplerate": (C.c_int, [_P, C.c_float]),
    "jsg_set_channels": (C.c_int, [_P, C.c_int]),
    "jsg_set_fft_size": (C.c_int, [_P, C.c_int]),
    "jsg_set_closest_fft_size_ms": (C.c_int, [_P, C.c_float]),
    "jsg_set_memory_time_s": (C.c_int, [_P, C.c_float]),
    "jsg_set_feed_percent": (C.c_int, [_P, C.c_int]),
    "jsg_set_feed_percent_ext": (C.c_int, [_P, C.c_float]),
    "jsg_set_pause_mode": (C.c_int, [_P, C.c_int]),
    "jsg_set_window": (C.c_int, [_P, C.c_int]),
    "jsg_set_window_table": (C.c_int, [_P, _P, C.c_int]),
    "jsg_set_mix_mode": (C.c_int, [_P, C.c_int]),
    "jsg_set_power_scale": (C.c_int, [_P, C.c_float]),
    "jsg_set_exact_log": (C.c_int, [_P, C.c_int]),
    "jsg_get_spectrum_size": (C.c_int, [_P]),
    "jsg_get_memory_size": (C.c_int, [_P]),
    "jsg_get_samplerate": (C.c_float, [_P]),
    "jsg_get_fft_size": (C.c_int, [_P]),
    "jsg_get_feed_samples": (C.c_int, [_P]),
    "jsg_get_feedblocks": (C.c_int, [_P]),
    "jsg_get_channels": (C.c_int, [_P]),
    "jsg_get_window": (C.c_int, [_P, _P, C.c_int]),
    "jsg_process_block": (C.c_int, [_P, C.POINTER(_P)]),
    "jsg_process_block_n": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int]),
    "jsg_process_block_wait": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.c_int]),
    "jsg_get_dropped_blocks": (C.c_longlong, [_P]),
    "jsg_process_blocks": (C.c_int, [_P, _P, C.c_int64, C.c_int]),
    "jsg_process_blocks_device": (C.c_int, [_P, _P, C.c_int64, C.c_int]),
    "jsg_get_mem": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "jsg_get_mem_rows": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "jsg_peek_mem": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int)]),
    "jsg_ring_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "jsg_sync": (C.c_int, [_P]),
    "jsg_stream": (_P, [_P]),
    "jsg_display_set_colormap": (C.c_int, [_P, C.c_int, C.c_int]),
    "jsg_display_set_running": (C.c_int, [_P, C.c_int]),
    "jsg_display_invalidate": (C.c_int, [_P]),
    "jsg_display_update": (C.c_int, [_P, C.c_float, C.c_float, _P, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "jsg_display_update_tile": (C.c_int, [_P, C.c_float, C.c_float, _P, C.c_int64, C.c_int, C.POINTER(C.c_int),
                                          C.POINTER(C.c_int)]),
    "jsg_display_freq_rows": (C.c_int, [C.c_float, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libjsg.so (once).  Raises if the HIP extension has not been built -- no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  jadespectrogram_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  If libjsg.so is
        # loaded first it pulls in /opt/rocm's copies, and the second runtime to initialise then sees no device.
        # Importing torch first makes the dynamic loader resolve libjsg.so's libamdhip64.so.7 to the copy that is
        # already mapped.  (A C++/JUCE host without torch simply uses /opt/rocm's runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, engine=None) -> int:
    if rc < 0:
        msg = lib().jsg_last_error(engine)
        raise JsgError(rc, msg.decode() if msg else "")
    return rc
