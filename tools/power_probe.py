#!/usr/bin/env python3
"""DEV TOOL (GPU box): core clock and socket power while each workload of the path keeps the card busy for PW_SECONDS.

Why: under the C2 dispatch rocm-smi shows the socket at its 1400 W cap with the core clock pulled down from 2.4 to ~1.9 GHz, i.e. the
kernel is POWER-bound before it is HBM- or issue-bound.  This tool puts numbers on it: for the calibration copy, the C2 / C3 / C5
dispatches (and C2 with linear-power output, i.e. without the logarithm) it reports the achieved rate together with the median core
clock and power sampled from the card's hwmon files in sysfs (read-only; tools/hwmon.py).

    python tools/power_probe.py      [PW_SECONDS=6] [PW_ONLY=copy,c2,...]
"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):          # a variant build (python -m jadespectrogram_amd._build --variant NAME ...; round-4-tree variants: ABI 5)
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])
    if os.environ.get("SP_ABI5"):     # a library built from the round-4 tree: it lacks the entry points of ABI 6 (the argument structs only grew at the end)
        jsg.capi.SIGNATURES.pop("jsg_process_block_wait", None)
TAG = os.environ.get("PW_TAG", os.path.basename(os.environ.get("SP_LIB", "product")))

SECONDS = float(os.environ.get("PW_SECONDS", 6))
ONLY = [s for s in os.environ.get("PW_ONLY", "").split(",") if s]


from hwmon import Hwmon, Watch   # (tools/hwmon.py)

HW = Hwmon(0)


def watch(run_once, bytes_per_call, units_per_call, label):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            run_once(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    calls, t0 = 0, time.time()
    with Watch(HW, settle_s=min(1.5, SECONDS / 3)) as w:
        e0.record(st)
        while time.time() - t0 < SECONDS:
            with torch.cuda.stream(st):
                for _ in range(20):
                    run_once(st)
            calls += 20
            if calls % 200 == 0:
                st.synchronize()
        e1.record(st)
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    summ = w.summary()
    ups = units_per_call * calls / ms * 1e3
    watts = summ.get("socket_W_median")
    print(json.dumps({"build": TAG, "workload": label, "seconds": round(ms / 1e3, 2), "GBps_algorithmic": round(bytes_per_call * calls / ms / 1e6, 1),
                      "frac_of_8TBps": round(bytes_per_call * calls / ms / 1e6 / 8000, 4), "units_per_s": round(ups, 1),
                      "units_per_joule": (round(ups / watts, 1) if watts and ups else None),
                      "GB_algorithmic_per_joule": (round(bytes_per_call * calls / ms / 1e6 / watts, 3) if watts else None),
                      **summ, "cap_W": HW.cap_W()}), flush=True)


lib = jsg.capi.lib()


def want(k):
    return not ONLY or k in ONLY


if want("idle"):
    time.sleep(1.0)
    print(json.dumps({"workload": "idle", "sample": HW.sample(), "cap_W": HW.cap_W(), "hwmon": HW.dir}), flush=True)

if want("copy"):
    half_b = 1 << 30
    pool = torch.empty(2 * half_b // 4, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    src, dst = pool.data_ptr(), pool.data_ptr() + half_b
    watch(lambda st: lib.jsg_calib_copy_launch(ctypes.c_void_p(src), ctypes.c_void_p(dst), half_b, ctypes.c_void_p(st.cuda_stream)),
          2 * half_b, 0, "calibration copy, 1 GiB each way")
    del pool


def stft_case(label, n, hop, C, F, K, mix=None, linear=False, bpc=0, plan_select=0, tail=False):
    H = n // 2 + 1
    pitch = n // 2 if tail else (H + 31) // 32 * 32
    ns = (F * hop + n - hop + 3) // 4 * 4
    plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
    d_in = torch.empty((K, C, ns), dtype=torch.float32, device="cuda").uniform_(-0.5, 0.5)
    d_out = torch.empty((K, F, pitch), dtype=torch.float32, device="cuda")
    kw = {}
    if mix is not None:
        kw["mix_mode"] = mix
    if linear:
        kw["linear_out"] = True
    if bpc:
        kw["blocks_per_cu"] = bpc
    if plan_select:
        kw["plan_select"] = plan_select
    if tail:
        kw["d_tail"] = torch.empty((K, 1, F), dtype=torch.float32, device="cuda")
    watch(lambda st: jsg.stft_db_strided(plan, d_in, hop, F, d_out, stream=st.cuda_stream, **kw),
          K * (C * F * hop * 4 + F * H * 4), K * F * C, label)


if want("c2"):
    stft_case("C2: N=1024 hop 512 mono, 64 x 4096 frames per dispatch", 1024, 512, 1, 4096, 64)
if want("c2lin"):
    stft_case("C2 geometry, linear power out (no logarithm)", 1024, 512, 1, 4096, 64, linear=True)
if want("c2tail"):
    stft_case("C2, tail-plane column layout", 1024, 512, 1, 4096, 64, tail=True)
if want("c3a"):
    stft_case("C3 geometry, three-stage plan Cfg2048 (plan_select 1)", 2048, 512, 8, 4096, 12, mix=jsg.capi.MIX_ABSMEAN, plan_select=1)
if want("c3p"):
    stft_case("C3 geometry, pair plan Cfg2048P (plan_select 3)", 2048, 512, 8, 4096, 12, mix=jsg.capi.MIX_ABSMEAN, plan_select=3)
if want("c3"):
    stft_case("C3: N=2048 hop 512, 8 channels AbsMean, 12 x 4096 columns per dispatch", 2048, 512, 8, 4096, 12, mix=jsg.capi.MIX_ABSMEAN)
if want("c512"):
    stft_case("N=512 hop 512 mono, 32 x 8192 frames", 512, 512, 1, 8192, 32)
if want("c4096"):
    stft_case("N=4096 hop 512 mono, 32 x 2048 frames", 4096, 512, 1, 2048, 32)
if want("c2cached"):
    stft_case("C2 geometry, ONE batch of 4096 frames per dispatch, the same 16.8 MB every time (served by L2 / Infinity Cache: no HBM)", 1024, 512, 1, 4096, 1)
if want("c2cached8"):
    stft_case("C2 geometry, 8 x 4096 frames per dispatch, the same 134 MB every time (Infinity Cache)", 1024, 512, 1, 4096, 8)
if want("c2bpc3"):
    stft_case("C2, 3 workgroups per CU in the grid", 1024, 512, 1, 4096, 64, bpc=3)
if want("c2048a"):
    stft_case("N=2048 hop 512 mono, 32 x 4096, three-stage plan", 2048, 512, 1, 4096, 32, plan_select=1)
if want("c2048b"):
    stft_case("N=2048 hop 512 mono, 32 x 4096, two-stage plan", 2048, 512, 1, 4096, 32, plan_select=2)
if want("c1024x8"):
    stft_case("N=1024 hop 512, 8 channels AbsMean, 8 x 4096 columns", 1024, 512, 8, 4096, 8, mix=jsg.capi.MIX_ABSMEAN)
if want("c5"):
    n, hop, C, F, K = 4096, 512, 2, 1875, 44
    H = n // 2 + 1
    img_pitch = (F + 31) // 32 * 32
    ns = (F * hop + n - hop + 3) // 4 * 4
    plan5 = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
    d_in5 = torch.empty((K, C, ns), dtype=torch.float32, device="cuda").uniform_(-0.5, 0.5)
    d_img5 = torch.zeros((K, H, img_pitch), dtype=torch.int32, device="cuda")
    d_lut5 = torch.from_numpy(jsg.colormap_lut(256, jsg.capi.CM_JADE)).cuda()
    watch(lambda st: jsg.stft_image_strided(plan5, d_in5, hop, F, d_lut5, -50.0, 50.0, d_img5[:, :, :F], None, stream=st.cuda_stream,
                                            feedblocks=n // hop, mix_mode=jsg.capi.MIX_ABSMEAN),
          K * F * (C * hop * 4 + H * 4), K * F, "C5: N=4096 hop 512 stereo -> ARGB, 44 images of 1875 columns per dispatch")
