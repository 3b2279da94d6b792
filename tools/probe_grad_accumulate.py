"""How many ATen kernels does autograd spend accumulating into pre-existing .grad tensors (the data-parallel buckets) per step?
(run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from uncrtaints_amd.src import losses
dev = torch.device("cuda", 0)
model = bench.build_model(dev, seed=1)
crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
x, y, dates = bench.synthetic(4, 3, 256, 256, seed=1, device=dev)
params = [p for p in model.parameters() if p.requires_grad]
flat = torch.zeros(sum(p.numel() for p in params), device=dev)
def step(views):
    if views:
        off = 0
        flat.zero_()
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
    else:
        for p in params: p.grad = None
    out = model(x, batch_positions=dates)
    m, v = losses.split_prediction(out, 13, 26)
    loss, _ = crit(m, y, v)
    loss.backward()
for views in (False, True):
    for _ in range(2): step(views)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(views); torch.cuda.synchronize()
    rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
    n = sum(e.count for e in rows); t = sum(e.device_time_total for e in rows)
    print(f"grads as bucket views={views}: {len(params)} parameters, ATen device ops {n}, {t:.0f} us:", sorted(((e.count, e.key) for e in rows), reverse=True)[:5])
