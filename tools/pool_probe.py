#!/usr/bin/env python3
"""DEV TOOL: does the 0.58-or-0.61 of the C2 strided dispatch depend on WHICH memory the process was given?  (Run on the GPU box.)

The same strided dispatch (64 batches of 4096 frames, N=1024, hop 512) is timed over PP_POOLS separately allocated pools, all alive at
once, in PP_ROUNDS interleaved rounds; the float4 calibration copy is timed over the same pools.  If the pools of ONE process differ,
the spread of bench.py between processes is physical placement (fragment size / channel mapping of the pages a process happens to
get); if they agree and only processes differ, it is per-process state of the device.

    python tools/pool_probe.py   [PP_POOLS=8] [PP_POOL_GIB=2.5] [PP_ROUNDS=3] [PP_PITCH=544]
"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("PP_LIB"):          # a variant build (python -m jadespectrogram_amd._build --variant NAME ...)
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["PP_LIB"])

n, hop, F, K = 1024, 512, 4096, 64
H, pitch = 513, int(os.environ.get("PP_PITCH", 544))
ns = (F * hop + n - hop + 3) // 4 * 4
in_floats, out_floats = K * ns, K * F * pitch
algo = 4100 * F * K
pools = int(os.environ.get("PP_POOLS", 8))
gib = float(os.environ.get("PP_POOL_GIB", 2.5))
rounds = int(os.environ.get("PP_ROUNDS", 3))
words = int(gib * (1 << 30)) // 4
assert words >= in_floats + out_floats + 64
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
lib = jsg.capi.lib()
st = torch.cuda.Stream()
held = []
for p in range(pools):
    t = torch.empty(words, dtype=torch.float32, device="cuda")
    t[:in_floats].uniform_(-0.5, 0.5)
    held.append(t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t_stft(pool, reps=10):
    i_ = pool[:in_floats].view(K, 1, ns)
    off = (in_floats + 31) // 32 * 32
    o_ = pool[off:off + out_floats].view(K, F, pitch)
    with torch.cuda.stream(st):
        jsg.stft_db_strided(plan, i_, hop, F, o_, stream=st.cuda_stream)
        e0.record(st)
        for _ in range(reps):
            jsg.stft_db_strided(plan, i_, hop, F, o_, stream=st.cuda_stream)
        e1.record(st)
    torch.cuda.synchronize()
    return algo / (e0.elapsed_time(e1) * 1e3 / reps) / 8e6


def t_copy(pool, reps=10):
    half_b = (words * 4 // 2) // 4096 * 4096
    src, dst = pool.data_ptr(), pool.data_ptr() + half_b
    with torch.cuda.stream(st):
        lib.jsg_calib_copy_launch(ctypes.c_void_p(src), ctypes.c_void_p(dst), half_b, ctypes.c_void_p(st.cuda_stream))
        e0.record(st)
        for _ in range(reps):
            lib.jsg_calib_copy_launch(ctypes.c_void_p(src), ctypes.c_void_p(dst), half_b, ctypes.c_void_p(st.cuda_stream))
        e1.record(st)
    torch.cuda.synchronize()
    return 2 * half_b / (e0.elapsed_time(e1) * 1e3 / reps) / 8e6


stft = {p: [] for p in range(pools)}
copy = {p: [] for p in range(pools)}
for r in range(rounds):
    for p, pool in enumerate(held):
        stft[p].append(round(t_stft(pool), 4))
        copy[p].append(round(t_copy(pool), 4))
print(json.dumps({"lib": os.environ.get("PP_LIB", "product"), "pid": os.getpid(), "pools": pools, "pool_gib": gib, "stft_frac_per_pool": stft, "copy_frac_per_pool": copy,
                  "addr_gib": [round(h.data_ptr() / (1 << 30), 2) for h in held]}))
