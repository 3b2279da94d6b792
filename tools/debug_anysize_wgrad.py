"""Kernel-level check of the any-size tail corrections (GPU box): per-frame dW2 products, SE pooling, statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from uncrtaints_amd import engine as E, hip_backend as hb
torch.manual_seed(0)
dev = "cuda"
N, C, Ch, H, W = 2, 128, 256, 100, 100
geom = E.plan_geom(H, W); P, Pc = geom.P, geom.Pc
def padded(n, c):
    t = torch.zeros(n, c, 1, Pc, device=dev); t[..., :P] = torch.randn(n, c, 1, P, device=dev); return t
dy, h3, h2 = padded(N, C), padded(N, C), padded(N, Ch)
c1, c2, c3, mu = (torch.randn(N * C, device=dev) for _ in range(4))
A2, B2 = torch.randn(N * Ch, device=dev), torch.randn(N * Ch, device=dev)
with E.geom_scope(geom):
    G, _ = E.pw_wgrad(dy, h2, N, C, Ch, Pc, pro_d=E.PRO_NORMBWD, dk=(c1, c2, c3, mu), d2=h3, pro_x=E.PRO_AFFINE_GELU, xk=(A2, B2, None), per_frame=True)
    G0 = G.clone()
    hb.call("uncr_fix_wgrad_tail", G, N, C, Ch, c2, c3, mu, B2, geom.ntail, E._stream())
dh = (c1.view(N, C, 1) * dy.view(N, C, Pc)[..., :P] + c2.view(N, C, 1) * (h3.view(N, C, Pc)[..., :P] - mu.view(N, C, 1)) + c3.view(N, C, 1)).double()
z = torch.nn.functional.gelu((A2.view(N, Ch, 1) * h2.view(N, Ch, Pc)[..., :P] + B2.view(N, Ch, 1)).double())
ref = torch.einsum("nop,nip->noi", dh, z)
print("G uncorrected", float((G0.double() - ref).abs().max() / ref.abs().max()), "corrected", float((G.double() - ref).abs().max() / ref.abs().max()))
# SE pooling
with E.geom_scope(geom):
    pp = E.se_pool(h2, A2, B2, N * Ch, Pc)
    before = pp.buf.sum(1)[:, 0].clone()
    hb.call("uncr_fix_sepool_tail", pp.buf, pp.slots, B2, N * Ch, geom.ntail, E._stream())
after = pp.buf.sum(1)[:, 0]
refp = z.sum(-1).reshape(-1)
print("sepool uncorrected", float((before.double() - refp).abs().max() / refp.abs().max()), "corrected", float((after.double() - refp).abs().max() / refp.abs().max()))
# the same products on the row-scaled fp16 route (magnitude bounds given)
d_amax = dy.abs().amax(dim=(1, 2, 3)).view(N, 1).contiguous()
d2_amax = h3.abs().amax(dim=(1, 2, 3)).view(N, 1).contiguous()
x_ub = (A2.view(N, Ch, 1).abs() * h2.view(N, Ch, Pc).abs().amax(-1, keepdim=True) + B2.view(N, Ch, 1).abs()).reshape(-1).contiguous()
with E.geom_scope(geom):
    G2, _ = E.pw_wgrad(dy, h2, N, C, Ch, Pc, pro_d=E.PRO_NORMBWD, dk=(c1, c2, c3, mu), d2=h3, pro_x=E.PRO_AFFINE_GELU, xk=(A2, B2, None), per_frame=True,
                       d_amax=d_amax, d2_amax=d2_amax, x_ub=x_ub)
    hb.call("uncr_fix_wgrad_tail", G2, N, C, Ch, c2, c3, mu, B2, geom.ntail, E._stream())
print("G on the fp16x2 route, corrected", float((G2.double() - ref).abs().max() / ref.abs().max()))
