"""Which torch (ATen) kernels run inside one eager training step, and from where (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from uncrtaints_amd.src import losses
dev = torch.device("cuda", 0)
model = bench.build_model(dev, seed=1)
crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
x, y, dates = bench.synthetic(4, 3, 256, 256, seed=1, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(x, batch_positions=dates)
    m, v = losses.split_prediction(out, 13, 26)
    loss, _ = crit(m, y, v)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [s for s in e.stack if "uncrtaints_amd" in s or "bench" in s or "losses" in s or "optim" in s][:3]
    print(f"{e.key:28s} n={e.count:3d} cuda={e.device_time_total:8.1f}us  {' <- '.join(s.split('/')[-1] for s in st)}")
