#!/usr/bin/env python3
"""DEV TOOL (round 5): the C2 strided dispatch with the reference column layout (pitch 544 floats, bin n/2 inline: a 4-byte piece in a
17th 128-byte line per column) against the tail-plane layout (jsg_stft_args.out_tail: columns of exactly 512 floats + a dense plane of
bin n/2), interleaved rounds in one process, HIP events on the launch stream.  TP_N / TP_HOP / TP_CHANNELS / TP_FRAMES / TP_BATCHES."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):          # a variant build (python -m jadespectrogram_amd._build --variant NAME ...)
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])

n = int(os.environ.get("TP_N", "1024")); hop = int(os.environ.get("TP_HOP", "512")); C = int(os.environ.get("TP_CHANNELS", "1"))
F = int(os.environ.get("TP_FRAMES", "4096")); K = int(os.environ.get("TP_BATCHES", "64"))
reps = int(os.environ.get("TP_REPS", "12")); rounds = int(os.environ.get("TP_ROUNDS", "9"))
M, H = n // 2, n // 2 + 1
pitch = (H + 31) // 32 * 32
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
ns = (F * hop + n - hop + 3) // 4 * 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = torch.rand((K, C, ns), device="cuda", generator=g) - 0.5
ref = torch.full((K, F, pitch), -7.0, device="cuda")
dense = torch.full((K, F, M), -7.0, device="cuda")
tail = torch.full((K, 1, F), -7.0, device="cuda")
st = torch.cuda.Stream()
kw = dict(feedblocks=n // hop, mix_mode=jsg.capi.MIX_ABSMEAN)
algo = (4 * hop * C + 4 * H) * F
cfgs = [("reference layout (pitch %d, bin n/2 inline)" % pitch, lambda: jsg.stft_db_strided(plan, d_in, hop, F, ref, stream=st.cuda_stream, **kw)),
        ("tail plane (pitch %d + dense plane)" % M, lambda: jsg.stft_db_strided(plan, d_in, hop, F, dense, d_tail=tail, stream=st.cuda_stream, **kw))]
for bpc in [int(v) for v in os.environ.get("TP_BPC", "").split(",") if v]:      # workgroups per CU of the grid, tail-plane layout
    cfgs.append(("tail plane, %d workgroups per CU" % bpc,
                 (lambda bpc: lambda: jsg.stft_db_strided(plan, d_in, hop, F, dense, d_tail=tail, blocks_per_cu=bpc, stream=st.cuda_stream, **kw))(bpc)))
for bpc in [int(v) for v in os.environ.get("TP_BPC_REF", "").split(",") if v]:  # ... and the reference layout
    cfgs.append(("reference layout, %d workgroups per CU" % bpc,
                 (lambda bpc: lambda: jsg.stft_db_strided(plan, d_in, hop, F, ref, blocks_per_cu=bpc, stream=st.cuda_stream, **kw))(bpc)))
with torch.cuda.stream(st):
    for _, fn in cfgs:
        fn()
torch.cuda.synchronize()
same = bool(torch.equal(dense, ref[..., :M]) and torch.equal(tail[:, 0, :], ref[..., M]))
times = {label: [] for label, _ in cfgs}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    for label, fn in cfgs:
        with torch.cuda.stream(st):
            fn()
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
        torch.cuda.synchronize()
        times[label].append(e0.elapsed_time(e1) * 1e3 / reps)
for label, _ in cfgs:
    t = sorted(times[label]); med = t[len(t) // 2]
    print(json.dumps({"layout": label, "us_per_dispatch_median": round(med, 2), "best": round(t[0], 2), "frac_of_8_median": round(K * algo / med / 8e6, 4),
                      "frac_of_8_best": round(K * algo / t[0] / 8e6, 4), "rounds": [round(x, 1) for x in times[label]]}))
print(json.dumps({"kernel": jsg.stft_db_strided_kernel_name(plan, d_in, hop, F, ref, **kw), "n": n, "hop": hop, "channels": C, "frames": F, "batches": K,
                  "values_identical": same, "rotation_MB": round(K * (algo + (pitch - H) * 4 * F) / 1e6)}))
