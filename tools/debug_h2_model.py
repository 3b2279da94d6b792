"""Model-level A/B of the forward fp16 two-part GEMMs against the exact split (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from uncrtaints_amd import engine as E
import test_gpu_ddp as T

res = {}
for h2 in (True, False):
    E._H2_FWD = h2
    m = T._model()
    xs, ys, ds = zip(*(T._shard(r) for r in range(2)))
    from uncrtaints_amd.src import losses
    out = m(torch.cat(xs), batch_positions=torch.cat(ds))
    crit = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode="diag")
    loss = sum(crit(out[r:r + 1, :, :13], ys[r], out[r:r + 1, :, 13:26])[0] for r in range(2)) / 2
    loss.backward()
    res[h2] = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    res[h2].update({"buf/" + n: b.detach().cpu() for n, b in m.named_buffers() if "running" in n})
    res[h2]["out"] = out.detach().cpu()
    print("h2", h2, "loss", float(loss))
rows = []
for n in res[True]:
    a, b = res[True][n], res[False][n]
    s = b.abs().max().item()
    if s > 0:
        rows.append(((a - b).abs().max().item() / s, n))
rows.sort(reverse=True)
for r in rows[:12]:
    print(f"{r[0]:.3e} {r[1]}")
