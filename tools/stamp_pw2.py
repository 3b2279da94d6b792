"""Per-phase s_memtime stamps of the pw2 forward GEMM (library built with -DPWS_STAMP): UNCR_HIP_LIB=... python tools/stamp_pw2.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncrtaints_amd import engine as E

N, P, dev = 4, 65536, "cuda"
torch.manual_seed(0)
CASES = ((256, 128, 2, "pw2 fwd"), (128, 256, 1, "pw1 fwd"), (128, 256, 3, "dz + pass-B"))
for (Cin, Cout, pro, nm) in CASES:
    h2 = torch.randn(N, Cin, P, device=dev)
    W2 = E.pack_wt(torch.randn(Cout, Cin, device=dev) * 0.05, transpose=True)
    k2 = tuple(torch.rand(N * Cin, device=dev) for _ in range(3))
    ub2 = (k2[0].view(N, Cin) * h2.abs().amax(dim=2) + k2[1].view(N, Cin)).reshape(-1).contiguous()
    out = torch.empty(N, Cout, P, device=dev)
    st = torch.zeros(8192 * 8, device=dev)
    if pro == 3:
        # (the dz kernel reads e3 as an epilogue coefficient: the stamps overwrite the buffer, the values are irrelevant for timing)
        x2 = torch.randn(N, Cin, P, device=dev); aux = torch.randn(N, Cout, P, device=dev)
        ek = tuple(torch.rand(N * Cout, device=dev) for _ in range(3))
        a45 = torch.full((N, 1), 4.5, device=dev)
        fn = lambda: E.pw_gemm(h2, W2, N, Cin, Cout, P, pro=3, k=(k2[0], k2[1], k2[2]), x2=x2, epi=3, aux=aux, ek=ek + (st,), in_amax=a45, in2_amax=a45, out=out)
    else:
        fn = lambda: E.pw_gemm(h2, W2, N, Cin, Cout, P, pro=pro, k=k2 if pro == 2 else (k2[0], k2[1], None), epi=1, in_amax=ub2, out=out, ek=(None, None, None, st))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    s = st.view(-1, 8).cpu(); s = s[s[:, 3] > 0]
    nchunk = s[:, 3] * (Cin // 32)
    print(f"{nm}: {ms*1e3:.1f} us, blocks {len(s)}, tiles/block {s[:,3].mean():.1f}; prologue {s[:,0].mean():.0f} loop {s[:,1].mean():.0f} ticks "
          f"(= {s[:,1].mean()/ms/1e3:.0f} ticks/us); per tile: epilogue {(s[:,2]/s[:,3]).mean():.0f}; per chunk: kstep0 {(s[:,4]/nchunk).mean():.0f} "
          f"stage {(s[:,5]/nchunk).mean():.0f} request+barrier {(s[:,6]/nchunk).mean():.0f} kstep1 {(s[:,7]/nchunk).mean():.0f} "
          f"sum {((s[:,4]+s[:,5]+s[:,6]+s[:,7])/nchunk).mean():.0f}")
