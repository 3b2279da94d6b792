"""The seven wide-GEMM kernels of the step in isolation at the bench shape (N = 4 frames, 256 x 256), HIP-event timed:
    [UNCR_HIP_LIB=...] python tools/bench_pw.py [--iters 30] [--only pw2,pw1,dz,dx,wg1,wg2] [--N 4]
Prints one line per kernel: microseconds per launch and algorithmic TB/s (the bytes SURVEY 8(d) / DESIGN 4 count)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncrtaints_amd import engine as E  # noqa: E402
from uncrtaints_amd import hip_backend as hb  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
    iters, N = int(arg("--iters", 30)), int(arg("--N", 4))
    only = arg("--only", "pw1,pw2,dz,dx,wg1,wg2").split(",")
    dev, P = "cuda", 65536
    torch.manual_seed(0)
    t = lambda *s: torch.randn(*s, device=dev)
    amax = lambda x: x.abs().amax(dim=(1, 2)).view(N, 1).contiguous()
    res = {}
    x, x2, h2 = t(N, 128, P), t(N, 128, P), t(N, 256, P)
    W1 = E.pack_wt(t(256, 128) * 0.05, transpose=True)
    k = tuple(t(N * 128) for _ in range(3))
    ek = tuple(torch.rand(N * 256, device=dev) for _ in range(4))
    ub1 = (k[0].abs().view(N, 128) * x.abs().amax(dim=2) + k[1].abs().view(N, 128)).reshape(-1).contiguous()
    o256, o128 = torch.empty(N, 256, P, device=dev), torch.empty(N, 128, P, device=dev)
    if "pw1" in only:
        ms = timeit(lambda: E.pw_gemm(x, W1, N, 128, 256, P, pro=1, k=k, epi=1, in_amax=ub1, out=o256), iters)
        res["pw1 fwd  [128->256, affine]"] = (ms, 4.0 * N * P * (128 + 256))
    if "dz" in only:
        a1, a2 = amax(x), amax(x2)
        ms = timeit(lambda: E.pw_gemm(x, W1, N, 128, 256, P, pro=3, k=k, x2=x2, epi=3, aux=h2, ek=ek, in_amax=a1, in2_amax=a2, out=o256), iters)
        res["dz + pass-B [128->256, normbwd]"] = (ms, 4.0 * N * P * (2 * 128 + 2 * 256))
    if "pw2" in only:
        W2 = E.pack_wt(t(128, 256) * 0.05, transpose=True)
        k2 = tuple(torch.rand(N * 256, device=dev) for _ in range(3))
        ub2 = (k2[0].view(N, 256) * h2.abs().amax(dim=2) + k2[1].view(N, 256)).reshape(-1).contiguous()
        ms = timeit(lambda: E.pw_gemm(h2, W2, N, 256, 128, P, pro=2, k=k2, epi=1, in_amax=ub2, out=o128), iters)
        res["pw2 fwd  [256->128, GELU+SE]"] = (ms, 4.0 * N * P * (256 + 128))
    d, d2, xx = t(N, 256, P), t(N, 256, P), t(N, 128, P)
    dk = tuple(t(N * 256) for _ in range(3))
    if "wg1" in only:
        xk = (t(N * 128), t(N * 128), None)
        ms = timeit(lambda: E.pw_wgrad(d, xx, N, 256, 128, P, pro_d=3, dk=dk, d2=d2, pro_x=1, xk=xk), iters)
        res["wgrad [256x128] (+reduce)"] = (ms, 4.0 * N * P * (2 * 256 + 128))
    if "wg2" in only:
        k2f = tuple(torch.rand(N * 256, device=dev) for _ in range(3))
        wb = dict(d_amax=amax(xx), d2_amax=amax(x2), x_ub=(k2f[0].view(N, 256) * d.abs().amax(dim=2) + k2f[1].view(N, 256)).reshape(-1).contiguous())
        dk1 = tuple(v[:N * 128] for v in dk)
        ms = timeit(lambda: E.pw_wgrad(xx, d, N, 128, 256, P, pro_d=3, dk=dk1, d2=x2, pro_x=2, xk=(k2f[0], k2f[1], None), **wb), iters)
        res["wgrad [128x256] (+reduce)"] = (ms, 4.0 * N * P * (2 * 128 + 256))
    if "dx" in only:
        W1k = E.pack_wt(t(256, 128) * 0.05, transpose=False)
        dy, xh3 = t(N, 128, P), t(N, 128, P)
        c = tuple(t(N * 128) for _ in range(3))
        slots = hb.query("uncr_pw_stat_slots", N, 128, P)
        part = torch.empty(N * 128, slots, 2, device=dev)
        ad, ad2 = amax(d), amax(d2)
        ms = timeit(lambda: hb.call("uncr_pw_gemm_dx", d, d2, W1k, o128, dk[0], dk[1], dk[2], None, dy, xx, xh3, c[0], c[1], c[2], None, None,
                                    None, None, part, N, 256, 128, P, 0, None, ad, 1, ad2, 1, P, E._stream()), iters)
        res["dx [256->128, normbwd + skip]"] = (ms, 4.0 * N * P * (2 * 256 + 4 * 128))
    tag = os.path.basename(os.environ.get("UNCR_HIP_LIB", "base"))
    for name, (ms, by) in res.items():
        print(f"{tag:28s} {name:34s} {ms * 1e3:8.1f} us  {by / ms / 1e9:6.2f} TB/s  ({by / ms / 1e9 / 8.0:.3f} of 8)", flush=True)


if __name__ == "__main__":
    main()
