"""Launch time of the split weight-gradient kernels alone (UNCR_HIP_LIB selects a library variant): mean of 20 launches behind 5 warm-ups,
operands re-warmed by their producers' write pattern is NOT reproduced here (cold operands; N = 4 in the step is cache-assisted).
    python tools/time_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncrtaints_amd import engine as E  # noqa: E402

P, dev = 65536, "cuda"
torch.manual_seed(0)
t = lambda *s: torch.randn(*s, device=dev)


def run(N, shape):
    if shape == "256x128":
        d, d2, x = t(N, 256, P), t(N, 256, P), t(N, 128, P)
        dk = tuple(t(N * 256) for _ in range(3))
        xk = (t(N * 128), t(N * 128), None)
        fn = lambda: E.pw_wgrad(d, x, N, 256, 128, P, pro_d=3, dk=dk, d2=d2, pro_x=1, xk=xk, partials=True)
    else:
        d, d2, x = t(N, 128, P), t(N, 128, P), t(N, 256, P)
        dk = tuple(t(N * 128) for _ in range(3))
        k2 = tuple(torch.rand(N * 256, device=dev) for _ in range(2))
        ub = (k2[0].view(N, 256) * x.abs().amax(dim=2) + k2[1].view(N, 256)).reshape(-1).contiguous()
        a1, a2 = (v.abs().amax(dim=(1, 2)).view(N, 1).contiguous() for v in (d, d2))
        fn = lambda: E.pw_wgrad(d, x, N, 128, 256, P, pro_d=3, dk=dk, d2=d2, pro_x=2, xk=(k2[0], k2[1], None), partials=True,
                                d_amax=a1, d2_amax=a2, x_ub=ub)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 60, r[1]


for shape in ("256x128", "128x256"):
    print(shape, " ".join(f"N={N}: {run(N, shape)[0]:.1f} us" for N in (2, 4, 8, 12)), flush=True)
