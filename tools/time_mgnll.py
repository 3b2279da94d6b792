"""Development: the MGNLL kernels alone at the bench shape (B = 4, 13 bands, 256 x 256), 200 launches each -- run under
`rocprofv3 --kernel-trace --stats` for their average durations (tools/profile_bench.sh pattern)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from uncrtaints_amd import engine as E

B, K, H, W = 4, 13, 256, 256
g = torch.Generator().manual_seed(0)
for Kv in (13, 1):
    out = torch.rand(B, 1, K + Kv, H, W, generator=g).cuda()
    pred, var = out[:, :, :K], out[:, :, K:] + 0.1
    targ = torch.rand(B, 1, K, H, W, generator=g).cuda()
    gout = torch.ones((), device="cuda")
    for _ in range(200):
        loss, _ = E.mgnll_forward(pred, targ, var, 1e-8, "mean", False, False)
        E.mgnll_backward(gout, pred, targ, var, 1e-8, "mean")
    torch.cuda.synchronize()
    print(Kv, float(loss))
