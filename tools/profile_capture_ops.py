"""Which torch (ATen) ops are recorded into the captured training step, and from where (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from uncrtaints_amd.src import losses
dev = torch.device("cuda", 0)
model = bench.build_model(dev, seed=1)
model.temporal_aggregator.set_seed(1)
crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True)
x, y, dates = bench.synthetic(4, 3, 256, 256, seed=1, device=dev)
ctr = torch.zeros(1, dtype=torch.int64, device=dev)
model.temporal_aggregator.step_counter = ctr
def step():
    ctr.add_(1)
    opt.zero_grad(set_to_none=True)
    out = model(x, batch_positions=dates)
    m, v = losses.split_prediction(out, 13, 26)
    loss, _ = crit(m, y, v)
    loss.backward()
    opt.step()
    return loss
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    with torch.cuda.graph(g):
        step()
rows = [e for e in prof.key_averages(group_by_stack_n=14) if e.key.startswith("aten::") and any(k in e.key for k in
        ("fill", "zero", "ones", "copy", "clone", "contiguous", "cat", "add", "to", "_to_copy"))]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [s for s in e.stack if "uncrtaints_amd" in s or "bench" in s or "losses" in s or "optim" in s or "profile_capture" in s][:3]
    print(f"{e.key:24s} n={e.count:3d}  {' <- '.join(s.split('/')[-1] for s in st)}")
