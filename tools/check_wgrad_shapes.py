"""Weight-gradient and data GEMMs over many (frames, pixels) combinations against an fp64 einsum (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import engine as E
torch.manual_seed(0)
dev = "cuda"
def rel(a, b): return ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
for (N, P) in [(6, 6144), (6, 4096), (12, 5120), (2, 8192), (3, 3072), (1, 1024), (5, 7168), (7, 2048), (12, 65536 // 4)]:
    for (Cd, Cx, pro_x) in [(256, 128, 1), (128, 256, 2)]:
        d = torch.randn(N, Cd, P, device=dev); d2 = torch.randn(N, Cd, P, device=dev); x = torch.randn(N, Cx, P, device=dev)
        dk = tuple(torch.randn(N * Cd, device=dev) for _ in range(3))
        xk = (torch.randn(N * Cx, device=dev), torch.randn(N * Cx, device=dev), torch.rand(N * Cx, device=dev) if pro_x == 2 else None)
        dn = (dk[0].view(N, Cd, 1) * d + dk[1].view(N, Cd, 1) * d2 + dk[2].view(N, Cd, 1)).double().cpu()
        xa = (xk[0].view(N, Cx, 1) * x + xk[1].view(N, Cx, 1)).double().cpu()
        if pro_x == 2:
            xa = torch.nn.functional.gelu(xa)      # the SE scale is not applied in the weight-gradient operand
        ref = torch.einsum("nap,nbp->nab", dn, xa)
        for per_frame in (False, True):
            dW, _ = E.pw_wgrad(d, x, N, Cd, Cx, P, pro_d=3, dk=dk, d2=d2, pro_x=pro_x, xk=(xk[0], xk[1], None), per_frame=per_frame)
            r = ref if per_frame else ref.sum(0, keepdim=True)
            print(f"wgrad N={N} P={P} {Cd}x{Cx} per_frame={per_frame}: {rel(dW, r):.2e}")
