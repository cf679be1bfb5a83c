#!/usr/bin/env python3
"""DEV TOOL: resident workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor) of the default kernels."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jadespectrogram_amd import capi
torch.cuda.init()
lib = capi.lib()
lib.jsg_dev_occupancy.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
for n in (512, 1024, 2048, 4096, 8192):
    lds, thr = C.c_int(), C.c_int()
    nb = lib.jsg_dev_occupancy(n, C.byref(lds), C.byref(thr))
    print(f"N={n}: {nb} workgroups/CU x {thr.value} threads = {nb * thr.value // 64} waves/CU, LDS {lds.value} B per workgroup")
