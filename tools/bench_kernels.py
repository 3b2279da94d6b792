"""Microbenchmarks of single kernels at workload shapes (run on the GPU box).  Usage:
   python tools/bench_kernels.py [gemm|wgrad|dw|ew|agg] [--iters K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uncrtaints_amd import engine as E

def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    dev = "cuda"
    N, P = 4, 65536
    if what == "gemm":
        for (Cin, Cout, pro, epi) in [(128, 256, 0, 0), (128, 256, 1, 1), (128, 256, 3, 0), (256, 128, 0, 0), (256, 128, 2, 1), (256, 128, 3, 2), (128, 256, 3, 3)]:
            x = torch.randn(N, Cin, P, device=dev); x2 = torch.randn(N, Cin, P, device=dev)
            W = torch.randn(Cout, Cin, device=dev) * 0.05
            Wt = E.pack_wt(W, transpose=True)
            k = tuple(torch.randn(N * Cin, device=dev) for _ in range(3))
            aux = torch.randn(N, Cout, P, device=dev) if epi >= 2 else None
            ek = tuple(torch.rand(N * Cout, device=dev) for _ in range(4)) if epi == 3 else (None,) * 4
            out = torch.empty(N, Cout, P, device=dev)
            fn = lambda: E.pw_gemm(x, Wt, N, Cin, Cout, P, pro=pro, k=k, x2=x2 if pro == 3 else None, epi=epi, aux=aux, out=out, ek=ek)
            ms = timeit(fn, iters)
            fl = 2.0 * N * P * Cin * Cout
            by = 4.0 * N * P * (Cin * (2 if pro == 3 else 1) + Cout * (2 if epi >= 2 else 1))
            print(f"pw_gemm {Cin}->{Cout} pro{pro} epi{epi}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF  {by/ms/1e6:.0f} GB/s")
    elif what == "wgrad":
        for (Cd, Cx, pro_d, pro_x) in [(256, 128, 3, 1), (128, 256, 3, 2), (256, 128, 0, 0)]:
            d = torch.randn(N, Cd, P, device=dev); d2 = torch.randn(N, Cd, P, device=dev); x = torch.randn(N, Cx, P, device=dev)
            dk = tuple(torch.randn(N * Cd, device=dev) for _ in range(3)); xk = (torch.randn(N * Cx, device=dev), torch.randn(N * Cx, device=dev), None)
            fn = lambda: E.pw_wgrad(d, x, N, Cd, Cx, P, pro_d=pro_d, dk=dk, d2=d2 if pro_d == 3 else None, pro_x=pro_x, xk=xk)
            ms = timeit(fn, iters)
            print(f"pw_wgrad {Cd}x{Cx} pro_d{pro_d} pro_x{pro_x}: {ms*1e3:.1f} us  {2.0*N*P*Cd*Cx/ms/1e9:.1f} TF (incl. reduce)")
    elif what == "dw":
        from uncrtaints_amd import hip_backend as hb
        C, H, W = 256, 256, 256
        for Nf in (4, 12):
            t = lambda *s: torch.randn(*s, device=dev)
            h1, h2, du2, out = t(Nf, C, H, W), t(Nf, C, H, W), t(Nf, C, H, W), torch.empty(Nf, C, H, W, device=dev)
            cA, cB, k1, k2, k3 = (torch.randn(Nf * C, device=dev) for _ in range(5))
            w = t(C, 9)
            sf, sb = hb.query("uncr_dw_slots_fwd", H), hb.query("uncr_dw_slots_bwd", H)
            partf, partb, dwp = torch.empty(Nf * C, sf, 2, device=dev), torch.empty(Nf * C, sb, 2, device=dev), torch.empty(Nf * C, sb, 9, device=dev)
            ms = timeit(lambda: hb.call("uncr_dw_fwd", h1, cA, cB, w, out, partf, Nf, C, H, W, 0, 0, E._stream()), iters)
            print(f"dw_fwd N={Nf}: {ms*1e3:.1f} us  {8.0*Nf*C*H*W/ms/1e6:.0f} GB/s")
            ms = timeit(lambda: hb.call("uncr_dw_bwd", du2, h2, h1, k1, k2, k3, None, cA, cB, w, out, partb, dwp, None, 0, Nf, C, H, W, 0, 0, None, E._stream()), iters)
            print(f"dw_bwd N={Nf}: {ms*1e3:.1f} us  {16.0*Nf*C*H*W/ms/1e6:.0f} GB/s")
    elif what == "agg":
        # the L-TAE stage's full-resolution kernels at the bench shape
        from uncrtaints_amd import hip_backend as hb
        B, T, C, H, W, NH, AH = 4, 3, 128, 256, 256, 16, 32
        e = torch.randn(B, T, C, H, W, device=dev)
        att = torch.softmax(torch.randn(NH, B, T, AH, AH, device=dev), dim=2)
        dg = torch.randn(B, C, H, W, device=dev)
        for train in (False, True):
            fn = lambda: E.aggregate_forward(e, att, None, train, 0.1, 1234)
            ms = timeit(fn, iters)
            print(f"aggregate_fwd train={train}: {ms*1e3:.1f} us  {4.0*B*C*H*W*(T+1)/ms/1e6:.0f} GB/s")
            _, sv, _ = fn()
            ms = timeit(lambda: E.aggregate_backward(dg, sv), iters)
            print(f"aggregate_bwd (+bilinear adjoint) train={train}: {ms*1e3:.1f} us  {4.0*B*H*W*(C*(2*T+1)+NH*T)/ms/1e6:.0f} GB/s")
        ms = timeit(lambda: E.maxpool_forward(e, AH, AH), iters)
        print(f"maxpool_fwd: {ms*1e3:.1f} us  {4.0*B*T*C*H*W/ms/1e6:.0f} GB/s")
    elif what == "gemmscale":
        Cin, Cout = 128, 256
        for nb in (128, 256, 512, 1024, 2048):
            Pn = 128 * nb
            x = torch.randn(1, Cin, Pn, device=dev)
            Wt = E.pack_wt(torch.randn(Cout, Cin, device=dev) * 0.05, transpose=True)
            out = torch.empty(1, Cout, Pn, device=dev)
            fn = lambda: E.pw_gemm(x, Wt, 1, Cin, Cout, Pn, out=out)
            ms = timeit(fn, iters)
            print(f"blocks={nb}: {ms*1e3:.1f} us  {2.0*Pn*Cin*Cout/ms/1e9:.1f} TF")
    elif what == "stamps":
        Cin, Cout = 128, 256
        for nb in (256, 2048):
            Pn = 128 * nb
            x = torch.randn(1, Cin, Pn, device=dev)
            Wt = E.pack_wt(torch.randn(Cout, Cin, device=dev) * 0.05, transpose=True)
            out = torch.empty(1, Cout, Pn, device=dev)
            st = torch.zeros(4096 * 8, device=dev)
            fn = lambda: E.pw_gemm(x, Wt, 1, Cin, Cout, Pn, out=out, ek=(None, None, None, st))
            ms = timeit(fn, 10)
            s = st.view(-1, 8).cpu(); s = s[s[:, 3] > 0]
            tiles = s[:, 3].sum().item()
            print(f"tiles={nb} blocks={len(s)} wall {ms*1e3:.1f} us: prologue {s[:,0].mean():.0f}  loop+epilogues {s[:,1].mean():.0f} "
                  f"(per tile {(s[:,1] / s[:,3]).mean():.0f}, of which epilogue {(s[:,2] / s[:,3]).mean():.0f}) ticks; "
                  f"-> {(s[:,:2].sum(1)).mean()/ms/1e3:.0f} ticks/us if blocks span the kernel")
    elif what in ("traffic", "traffic_bf16"):
        # one launch sequence of the step's dominant kernels at the bench shapes, for rocprofv3 --pmc passes
        # (traffic_bf16: the same launches on bf16 activation storage, BASELINE config 3)
        import json
        from uncrtaints_amd import hip_backend as hb
        bf = what == "traffic_bf16"
        adt = torch.bfloat16 if bf else torch.float32
        act = 1 if bf else 0
        ta = lambda *s: torch.randn(*s, device=dev).to(adt)
        Cin, Cout = 128, 256
        x, x2, h2 = ta(N, Cin, P), ta(N, Cin, P), ta(N, Cout, P)
        Wt = E.pack_wt(torch.randn(Cout, Cin, device=dev) * 0.05, transpose=True)
        k = tuple(torch.randn(N * Cin, device=dev) for _ in range(3)); ek = tuple(torch.rand(N * Cout, device=dev) for _ in range(4))
        # magnitude bounds, so that the fp32 launches take the fp16 two-part kernels they take inside the step
        amax = lambda t: t.float().abs().amax(dim=(1, 2)).view(N, 1).contiguous()
        ub1 = (k[0].abs().view(N, Cin) * x.float().abs().amax(dim=2) + k[1].abs().view(N, Cin)).reshape(-1).contiguous()
        for _ in range(3):
            E.pw_gemm(x, Wt, N, Cin, Cout, P, pro=3, k=k, x2=x2, epi=3, aux=h2, ek=ek, in_amax=amax(x), in2_amax=amax(x2))   # fused dz + pass-B
            E.pw_gemm(x, Wt, N, Cin, Cout, P, pro=1, k=k, epi=1, in_amax=ub1)                    # pw1 forward
        Wt2 = E.pack_wt(torch.randn(Cin, Cout, device=dev) * 0.05, transpose=True)
        k2f = tuple(torch.rand(N * Cout, device=dev) for _ in range(3))
        ub2 = (k2f[0].view(N, Cout) * h2.float().abs().amax(dim=2) + k2f[1].view(N, Cout)).reshape(-1).contiguous()
        for _ in range(3):
            E.pw_gemm(h2, Wt2, N, Cout, Cin, P, pro=2, k=k2f, epi=1, in_amax=ub2)                # pw2 forward (SE scale + GELU prologue)
        # the two wide weight-gradient GEMMs of an MBConv backward
        d, d2, xx = ta(N, 256, P), ta(N, 256, P), ta(N, 128, P)
        dk = tuple(torch.randn(N * 256, device=dev) for _ in range(3)); xk = (torch.randn(N * 128, device=dev), torch.randn(N * 128, device=dev), None)
        for _ in range(3):
            E.pw_wgrad(d, xx, N, 256, 128, P, pro_d=3, dk=dk, d2=d2, pro_x=1, xk=xk)
            wb = {} if bf else dict(d_amax=amax(xx), d2_amax=amax(x2), x_ub=(k2f[0].view(N, 256) * d.float().abs().amax(dim=2)
                                                                          + k2f[1].view(N, 256)).reshape(-1).contiguous())
            E.pw_wgrad(xx, d, N, 128, 256, P, pro_d=3, dk=tuple(t[:N * 128] for t in dk), d2=x2, pro_x=2, xk=(k2f[0], k2f[1], None), **wb)
        # backward of pw1 with the PreNorm backward + skip epilogue (uncr_pw_gemm_dx, with the producer's statistics)
        W1k = E.pack_wt(torch.randn(256, 128, device=dev) * 0.05, transpose=False)
        dy, xh3 = ta(N, 128, P), ta(N, 128, P)
        dx = torch.empty(N, 128, P, device=dev, dtype=adt)
        c = tuple(torch.randn(N * 128, device=dev) for _ in range(3))
        slots = hb.query("uncr_pw_stat_slots", N, 128, P)
        part = torch.empty(N * 128, slots, 2, device=dev)
        for _ in range(3):
            hb.call("uncr_pw_gemm_dx", d, d2, W1k, dx, dk[0], dk[1], dk[2], None, dy, xx, xh3, c[0], c[1], c[2], None, None, None, None, part,
                    N, 256, 128, P, act, None, None if bf else amax(d), 0 if bf else 1, None if bf else amax(d2), 0 if bf else 1, P, E._stream())
        # the depthwise kernels
        C, H, W = 256, 256, 256
        t4 = lambda *s: torch.randn(*s, device=dev).to(adt)
        h1, hh2, du2, out = t4(N, C, H, W), t4(N, C, H, W), t4(N, C, H, W), torch.empty(N, C, H, W, device=dev, dtype=adt)
        cA, cB, k1, k2, k3 = (torch.randn(N * C, device=dev) for _ in range(5))
        w9 = torch.randn(C, 9, device=dev)
        sf, sb = hb.query("uncr_dw_slots_fwd", H), hb.query("uncr_dw_slots_bwd", H)
        partf, partb, dwp = torch.empty(N * C, sf, 2, device=dev), torch.empty(N * C, sb, 2, device=dev), torch.empty(N * C, sb, 9, device=dev)
        for _ in range(3):
            hb.call("uncr_dw_fwd", h1, cA, cB, w9, out, partf, N, C, H, W, act, 0, E._stream())
            hb.call("uncr_dw_bwd", du2, hh2, h1, k1, k2, k3, None, cA, cB, w9, out, partb, dwp, None, 0, N, C, H, W, act, 0,
                    None if bf else torch.empty(N * C * hb.query("uncr_dw_slots_bwd", H), device=dev), E._stream())
        torch.cuda.synchronize()
        print("done")
    elif what == "ablate":
        Cin, Cout = 128, 256
        x = torch.randn(N, Cin, P, device=dev); W = torch.randn(Cout, Cin, device=dev) * 0.05
        Wt = E.pack_wt(W, transpose=True); out = torch.empty(N, Cout, P, device=dev)
        from uncrtaints_amd import hip_backend as hb
        for flags, name in [(0, "full")]:
            fn = lambda: hb.call("uncr_pw_gemm", x, None, Wt, out, None, None, None, None, None, 0, None, None, None, None, None, None, N, Cin, Cout, P, 0, flags, 0, 0, None, None, 0, None, 0, P, E._stream())
            ms = timeit(fn, iters)
            print(f"{name:28s}: {ms*1e3:.1f} us  {2.0*N*P*Cin*Cout/ms/1e9:.1f} TF")
    elif what == "mfma":
        from uncrtaints_amd import hip_backend as hb
        out = torch.zeros(4, device=dev)
        for blocks in (256, 512, 1024):
            its = 4000
            fn = lambda: hb.call("uncr_debug_mfma_probe", out, blocks, its, E._stream())
            ms = timeit(fn, 5)
            fl = blocks * 4 * its * 8 * (2.0 * 32 * 32 * 2)
            print(f"mfma probe blocks={blocks}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF")
            fn = lambda: hb.call("uncr_debug_mfma_probe_bf16", out, blocks, its, E._stream())
            ms = timeit(fn, 5)
            fl = blocks * 4 * its * 8 * (2.0 * 32 * 32 * 16)
            print(f"bf16 mfma probe blocks={blocks}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF")
    elif what == "copy":
        a = torch.randn(256 * 1024 * 1024 // 4, device=dev); b = torch.empty_like(a)
        ms = timeit(lambda: b.copy_(a), iters)
        print(f"torch copy 256MB: {ms*1e3:.1f} us  {2*a.numel()*4/ms/1e6:.0f} GB/s")
        planes, PP = 1024, 65536
        x = torch.randn(planes, PP, device=dev); h = torch.randn(planes, PP, device=dev); y = torch.empty_like(x)
        kA = torch.randn(planes, device=dev); kB = torch.randn(planes, device=dev)
        ms = timeit(lambda: E.ew(E.EW_RESIDUAL, x, b=h, out=y, k=(kA, kB, None, None), want_part=True, planes=planes, P=PP), iters)
        print(f"ew residual 3x256MB: {ms*1e3:.1f} us  {3*x.numel()*4/ms/1e6:.0f} GB/s")

if __name__ == "__main__":
    main()
