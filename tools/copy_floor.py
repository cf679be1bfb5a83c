import torch, json
def t(nbytes, steps=200, nbuf=None):
    n = nbytes // 4
    nbuf = nbuf or max(2, int(600e6 // (2 * nbytes)) + 1)
    src = [torch.rand(n, device="cuda") for _ in range(nbuf)]; dst = [torch.empty(n, device="cuda") for _ in range(nbuf)]
    for i in range(20): dst[i % nbuf].copy_(src[i % nbuf])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): dst[i % nbuf].copy_(src[i % nbuf])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    return dict(copy_bytes_each_way=nbytes, us=round(us, 2), TBs=round(2 * nbytes / us / 1e6, 3))
for nb in (8_400_000, 33_600_000, 134_000_000, 537_000_000):
    print(json.dumps(t(nb)), flush=True)
