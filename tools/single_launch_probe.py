import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jadespectrogram_amd as jsg
if os.environ.get("SP_LIB"):
    jsg.capi.LIB_PATH = os.path.abspath(os.environ["SP_LIB"])
n, hop, F, K = 1024, 512, 4096, 64
plan = jsg.Plan(n, jsg.window(jsg.capi.WIN_HANN, n))
ns = F * hop + n - hop
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_in = torch.rand((K, 1, ns), device="cuda", generator=g) - 0.5
out = torch.empty((K, F, 544), device="cuda")
dense = torch.empty((K, F, 512), device="cuda"); tail = torch.empty((K, 1, F), device="cuda")
st = torch.cuda.Stream()
algo = (4 * hop + 4 * 513) * F
res = {}
for tailmode in (False, True):
    for bpc in (0, 1, 2, 3, 4, 6):
        gph = torch.cuda.CUDAGraph()
        def run():
            for b in range(K):
                if tailmode: jsg.stft_db(plan, d_in[b], hop, F, dense[b], d_tail=tail[b], blocks_per_cu=bpc, stream=st.cuda_stream)
                else: jsg.stft_db(plan, d_in[b], hop, F, out[b], blocks_per_cu=bpc, stream=st.cuda_stream)
        with torch.cuda.stream(st):
            run(); torch.cuda.synchronize()
            with torch.cuda.graph(gph, stream=st):
                run()
            gph.replay(); gph.replay(); torch.cuda.synchronize()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(8): gph.replay()
                e1.record(st); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / (8 * K))
        ts.sort()
        print(json.dumps({"tail": tailmode, "blocks_per_cu": bpc, "us_per_launch_median": round(ts[2], 3), "frac_of_8": round(algo / ts[2] / 8e6, 4)}), flush=True)
