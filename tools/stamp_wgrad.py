"""Per-block phase stamps of the split weight-gradient kernel (library built with -DWGS_STAMP, tools/build_file_variant.sh):
    UNCR_HIP_LIB=uncrtaints_amd/lib/ablate/lib_wgstamp.so python tools/stamp_wgrad.py
Prints, per shape and frame count, the kernel time and the mean / max of: prologue (coefficients, bounds), chunk loop, epilogue
(partial tile stores, acknowledged) in s_memtime ticks, and the spread of the blocks' start and end stamps."""
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
        am = lambda v: v.abs().amax(dim=(1, 2)).view(N, 1).contiguous()
        fn = lambda: E.pw_wgrad(d, x, N, 128, 256, P, pro_d=3, dk=dk, d2=d2, pro_x=2, xk=(k2[0], k2[1], None), partials=True,
                                d_amax=am(d), d2_amax=am(d2), x_ub=ub)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if shape != "256x128":      # (the bounds are computed outside the timed launch)
        a1, a2 = am(d), am(d2)
        fn = lambda: E.pw_wgrad(d, x, N, 128, 256, P, pro_d=3, dk=dk, d2=d2, pro_x=2, xk=(k2[0], k2[1], None), partials=True,
                                d_amax=a1, d2_amax=a2, x_ub=ub)
        fn()
    e0.record(); part, nbx, cop, cip = fn(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    raw = part.view(N * nbx, -1)[:, :8].cpu().double()
    rtot = raw[:, 0] + raw[:, 6] + raw[:, 1] + raw[:, 2]
    # the counter runs at the shader clock: every block lives for the whole kernel, so mean block total = the kernel's event time
    s = raw * (us / float(rtot.mean()))
    s[:, 3], s[:, 7], s[:, 4] = raw[:, 3] * 0.01, raw[:, 7] * 0.01, raw[:, 4] * 0.01
    tot = s[:, 0] + s[:, 6] + s[:, 1] + s[:, 2]
    f = lambda v: f"{v.mean():.1f}/{v.max():.1f}"
    xcc = (s[:, 7] * 100).round().long()
    spread = []
    for x in range(8):      # the counters of different XCDs need not agree: start spread inside each XCD
        st = (s[xcc == x, 4] * 100) % (1 << 24)
        if st.numel():
            spread.append(float(st.max() - st.min()) * 0.01)
    print(f"wgrad [{shape}] N={N}: {us:.1f} us, {N * nbx} blocks x {s[:, 3].mean() * 100:.0f} chunks; us mean/max: coefficients {f(s[:, 0])} "
          f"first chunk {f(s[:, 6])} loop {f(s[:, 1])} epilogue {f(s[:, 2])} block total {f(tot)}; start spread inside an XCD "
          f"{max(spread) if spread else -1:.1f} us; per chunk {float((s[:, 1] / s[:, 3]).mean()) * 10:.0f} ns", flush=True)


for shape in ("256x128", "128x256"):
    for N in (2, 4, 8, 12):
        run(N, shape)
