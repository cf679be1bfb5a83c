#!/usr/bin/env python3
"""DEV TOOL: does the access width matter?  134 MB copies: 16-byte loads/stores vs 8-byte loads + 4-byte stores."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
lib = jsg.capi.lib()
lib.jsg_dev_copy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
nbytes = 134_217_728
src = [torch.rand(nbytes // 4, device="cuda") for _ in range(3)]; dst = [torch.empty(nbytes // 4, device="cuda") for _ in range(3)]
st = torch.cuda.current_stream().cuda_stream
for mode, name in ((1, "16B loads, 16B nt stores"), (2, "8B loads, 4B nt stores")):
    for blocks in (512, 1024, 2048, 4096):
        for i in range(5): lib.jsg_dev_copy_launch(src[i % 3].data_ptr(), dst[i % 3].data_ptr(), nbytes, blocks, mode, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): lib.jsg_dev_copy_launch(src[i % 3].data_ptr(), dst[i % 3].data_ptr(), nbytes, blocks, mode, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(json.dumps(dict(kernel=name, blocks=blocks, us=round(us, 1), TBs=round(2 * nbytes / us / 1e6, 2))), flush=True)
