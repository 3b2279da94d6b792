"""Accuracy of the device erf / GELU / v_exp_f32 against fp64 (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.special import erf
from uncrtaints_amd import hip_backend as hb
g = torch.Generator().manual_seed(0)
x = torch.cat([torch.randn(2_000_000, generator=g) * 1.5, torch.linspace(-6, 6, 2_000_001)]).cuda()
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
xd = x.double().cpu().numpy()
phi = lambda u: 0.5 * (1 + erf(u / np.sqrt(2)))
refs = {0: erf(xd), 1: xd * phi(xd), 2: phi(xd) + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)}
for what, name in ((0, "erf_f"), (1, "gelu_f"), (2, "gelu_grad_f")):
    hb.call("uncr_debug_erf", x, y, x.numel(), what, st)
    d = y.double().cpu().numpy() - refs[what]
    print(f"{name}: max abs {np.abs(d).max():.3e}  mean {d.mean():+.3e}  rms {np.sqrt((d*d).mean()):.3e}")
a = -torch.rand(4_000_000, generator=g).cuda() * 20
hb.call("uncr_debug_erf", a, y[: a.numel()], a.numel(), 3, st)
ad = a.double().cpu().numpy()
rel = (y[: a.numel()].double().cpu().numpy() - np.exp2(ad)) / np.exp2(ad)
print(f"v_exp_f32 on [-20,0]: max rel {np.abs(rel).max():.3e}  mean rel {rel.mean():+.3e}  rms {np.sqrt((rel*rel).mean()):.3e}")
t32 = torch.erf(x)
d = t32.double().cpu().numpy() - refs[0]
print(f"torch.erf (device): max abs {np.abs(d).max():.3e}  mean {d.mean():+.3e}  rms {np.sqrt((d*d).mean()):.3e}")
