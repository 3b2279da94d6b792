"""Losses and metrics at odd sizes against the oracle (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.learning import metrics
def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
g = torch.Generator().manual_seed(0)
bad = 0
for (B, H, W) in [(1, 1, 1), (1, 3, 5), (2, 7, 9), (3, 33, 17), (1, 64, 65), (2, 31, 128), (5, 16, 16), (1, 255, 3)]:
    for mode, kv in (("diag", 13), ("iso", 1)):
        mu = torch.rand(B, 1, 13, H, W, generator=g); t = torch.rand(B, 1, 13, H, W, generator=g)
        var = torch.rand(B, 1, kv, H, W, generator=g) * 0.5 + 1e-3
        for red in ("mean", "sum", "none"):
            mo, vo = mu.clone().requires_grad_(True), var.clone().requires_grad_(True)
            lo = orc.mgnll(mo, t, vo, mode=mode, reduction=red)[0]
            md, vd = mu.cuda().requires_grad_(True), var.cuda().requires_grad_(True)
            ld, _ = losses.MultiGaussianNLLLoss(reduction=red, full=True, mode=mode)(md, t.cuda(), vd)
            e = [rel(ld.detach(), lo.detach())]
            w = torch.rand(lo.shape, generator=g) if red == "none" else torch.tensor(1.0)
            (lo * w).sum().backward(); (ld * w.cuda()).sum().backward()
            e += [rel(md.grad, mo.grad), rel(vd.grad, vo.grad)]
            if max(e) > 2e-5: bad += 1; print("MGNLL", (B, H, W), mode, red, ["%.1e" % v for v in e])
    if H >= 1:
        targ = torch.rand(B, 13, H, W, generator=g); pred = (targ + 0.1 * torch.randn(B, 13, H, W, generator=g)).clamp(0, 1); v = torch.rand(B, 13, H, W, generator=g)
        do = orc.img_metrics(targ, pred, v); dd = metrics.img_metrics(targ.cuda(), pred.cuda(), v.cuda())
        for k in do:
            if not np.allclose(np.asarray(dd[k]), np.asarray(do[k]), rtol=1e-4, atol=5e-6, equal_nan=True):
                bad += 1; print("metrics", (B, H, W), k, np.asarray(dd[k]).ravel()[:3], np.asarray(do[k]).ravel()[:3])
print("bad:", bad)
