"""Systematic part of the GELU / GELU' error (GPU box): mean error and its correlation with u over u ~ N(0, s), for the shipped fit and for
torch's fp32 erf-based GELU.  A random rounding error of rms r averages to r / sqrt(n) over n samples; a fit's error does not average out."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.special import erf
from uncrtaints_amd import hip_backend as hb
n = 4_000_000
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
f = hb.dev_lib().cdll.uncr_debug_erf
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
phi = lambda u: 0.5 * (1 + erf(u / np.sqrt(2)))
for scale, shift in ((1.0, 0.0), (0.5, 0.3), (2.0, -0.5)):
    x = (torch.randn(n, generator=g) * scale + shift).cuda()
    xd = x.double().cpu().numpy()
    refs = {1: xd * phi(xd), 2: phi(xd) + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)}
    xg = x.clone().requires_grad_(True)
    yt = torch.nn.functional.gelu(xg)
    yt.sum().backward()
    cand = {}
    for what, fn in ((1, "gelu"), (2, "gelu'")):
        y = torch.empty_like(x)
        assert f(x.data_ptr(), y.data_ptr(), x.numel(), what, st) == 0
        cand[("fit", fn)] = y.double().cpu().numpy() - refs[what]
    cand[("torch", "gelu")] = yt.detach().double().cpu().numpy() - refs[1]
    cand[("torch", "gelu'")] = xg.grad.double().cpu().numpy() - refs[2]
    print(f"u ~ N({shift}, {scale}^2), n = {n}")
    for (who, fn), d in sorted(cand.items(), key=lambda kv: kv[0][1]):
        rms = np.sqrt((d * d).mean())
        print(f"   {who:6s} {fn:6s}: rms {rms:.2e}  mean {d.mean():+.2e} ({abs(d.mean()) / (rms / np.sqrt(n)):6.1f} sigma)  "
              f"mean(d*u) {np.mean(d * xd):+.2e} ({abs(np.mean(d * xd)) / (np.sqrt(np.mean((d * xd) ** 2)) / np.sqrt(n)):6.1f} sigma)")
