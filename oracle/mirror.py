"""ctypes loader of the kernel mirror (oracle/jsg_mirror.c).  TEST INFRASTRUCTURE ONLY: the bit-exactness checker of the GPU path."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libjsg_mirror.so")
PLANS = {512: ("Cfg512",), 1024: ("Cfg1024", "Cfg1024B"), 2048: ("Cfg2048", "Cfg2048B", "Cfg2048P"), 4096: ("Cfg4096", "Cfg4096B"), 8192: ("Cfg8192",)}


class _Mirror:
    def __init__(self, lib):
        self.lib = lib
        lib.jsg_mirror_columns.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p]
        lib.jsg_mirror_columns.restype = ctypes.c_int
        lib.jsg_mirror_exact_db.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
        lib.jsg_mirror_exact_db.restype = None

    def columns(self, plan, x, hop, n_frames, win, feedblocks=None, mix=0, power_scale=1.0, first_frame=0, exact_db=False):
        """plan: kernel name ("Cfg1024", "Cfg2048B", ...; what jsg_stft_kernel_name reports); x: [C][samples] float32.  Returns
        [n_frames][n/2+1] float32: mixed linear power in the GPU kernel's own operation order, or dB through the shared logarithm."""
        n = int(plan[3:].rstrip("BP"))
        x = np.ascontiguousarray(x, dtype=np.float32)
        win = np.ascontiguousarray(win, dtype=np.float32)
        assert win.size == n
        fb = feedblocks if feedblocks is not None else max(1, n // hop)
        out = np.empty((n_frames, n // 2 + 1), dtype=np.float32)
        rc = self.lib.jsg_mirror_columns(plan.encode(), x.ctypes.data, x.shape[1], x.shape[0], hop, fb, first_frame, n_frames, win.ctypes.data,
                                         power_scale, mix, int(bool(exact_db)), out.ctypes.data)
        assert rc == 0, rc
        return out

    def exact_db(self, p):
        p = np.ascontiguousarray(p, dtype=np.float32)
        out = np.empty_like(p)
        self.lib.jsg_mirror_exact_db(p.ctypes.data, out.ctypes.data, p.size)
        return out


def load(build=True):
    deps = [os.path.join(HERE, "jsg_mirror.c"), os.path.join(HERE, "..", "jadespectrogram_amd", "csrc", "jsg_exact_math.h")]
    if build and (not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(d) for d in deps)):
        subprocess.check_call(["make", "-s", "-C", HERE, "port"])
    return _Mirror(ctypes.CDLL(SO))
