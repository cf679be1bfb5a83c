"""ctypes loader of the plain-C oracle (oracle/jsg_oracle_c.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libjsg_oracle_c.so")


class _Port:
    def __init__(self, lib):
        self.lib = lib
        lib.jsg_oracle_stft_db.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                           ctypes.c_void_p, ctypes.c_int]
        lib.jsg_oracle_stft_db.restype = ctypes.c_int
        lib.jsg_oracle_colour_columns.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_long,
                                                  ctypes.c_int]
        lib.jsg_oracle_colour_columns.restype = ctypes.c_int

    def stft_db(self, x, n, hop, n_frames, win, feedblocks=None, mix=0, power_scale=1.0, threads=1, out=None):
        """x: [C][samples] float32 (already carrying any pre-roll); returns [n_frames][n/2+1] dB (written into `out`
        when given: timing loops reuse one buffer instead of page-faulting a fresh one per call)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        win = np.ascontiguousarray(win, dtype=np.float32)
        fb = feedblocks if feedblocks is not None else max(1, n // hop)
        if out is None:
            out = np.empty((n_frames, n // 2 + 1), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (n_frames, n // 2 + 1)
        rc = self.lib.jsg_oracle_stft_db(x.ctypes.data, x.shape[0], x.shape[1], n, hop, fb, n_frames, win.ctypes.data,
                                         mix, power_scale, out.ctypes.data, threads)
        assert rc == 0
        return out

    def colour_columns(self, db, lut, vmin, vmax, mult, threads=1):
        """db: [W][H] float32 -> ARGB image [H][W] uint32 (x = column, low frequencies at the bottom)."""
        db = np.ascontiguousarray(db, dtype=np.float32)
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        W, H = db.shape
        img = np.empty((H, W), dtype=np.uint32)
        rc = self.lib.jsg_oracle_colour_columns(db.ctypes.data, W, H, lut.ctypes.data, lut.size, vmin, vmax, mult,
                                                img.ctypes.data, W, threads)
        assert rc == 0
        return img


def load(build=True):
    if build and (not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "jsg_oracle_c.c"))):
        subprocess.check_call(["make", "-s", "-C", HERE, "port"])
    return _Port(ctypes.CDLL(SO))
