"""Accuracy of gelu_f / gelu_grad_f in the shipped form and in the re-associated form (libdev built with -DUNCR_GELU_DIET=1) vs fp64."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.special import erf
from uncrtaints_amd import hip_backend as hb
g = torch.Generator().manual_seed(0)
x = torch.cat([torch.randn(2_000_000, generator=g) * 1.5, torch.linspace(-6, 6, 2_000_001)]).cuda()
xd = x.double().cpu().numpy()
phi = lambda u: 0.5 * (1 + erf(u / np.sqrt(2)))
refs = {1: xd * phi(xd), 2: phi(xd) + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)}
st = torch.cuda.current_stream().cuda_stream
libs = {"shipped": hb.dev_lib().cdll}
alt = os.path.join(hb.HERE, "lib", "ablate", "libdev_diet.so")
if os.path.exists(alt):
    libs["diet"] = ctypes.CDLL(alt)
for name, lib in libs.items():
    f = lib.uncr_debug_erf
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for what, fn in ((1, "gelu_f"), (2, "gelu_grad_f")):
        y = torch.empty_like(x)
        assert f(x.data_ptr(), y.data_ptr(), x.numel(), what, st) == 0
        d = y.double().cpu().numpy() - refs[what]
        rel = np.abs(d) / np.maximum(np.abs(refs[what]), 1e-30)
        neg = xd < -2
        print(f"{name:8s} {fn:12s}: max abs {np.abs(d).max():.3e}  mean {d.mean():+.3e}  rms {np.sqrt((d*d).mean()):.3e}  max rel on x < -2: {rel[neg].max():.3e}")
