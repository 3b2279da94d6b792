"""uncr_prenorm_bwd_finish against an fp64 re-evaluation of its own inputs, for every call of one backward pass (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import dev
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E, hip_backend as hb
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

kw = eval("dict(" + sys.argv[1] + ")")
B, T, H, W = (int(v) for v in sys.argv[2].split(","))
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=7)
x[B - 1, T - 1] = 0.0
torch.manual_seed(6)
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
m.temporal_aggregator.attn_dropout.p = 0.0
m = m.to("cuda").train()
calls = []
orig = hb.call
def spy(name, *a):
    r = orig(name, *a)
    if name == "uncr_prenorm_bwd_finish":
        calls.append([t.clone() if torch.is_tensor(t) else t for t in a])
    return r
hb.call = spy; E.hb.call = spy
wg = []
_pw = E.pw_wgrad
def spy_wg(d, x, N, Cd, Cx, P, **k):
    r = _pw(d, x, N, Cd, Cx, P, **k)
    if k.get("partials"):
        wg.append(dict(d=d.clone(), x=x.clone(), dk=[t.clone() if t is not None else None for t in k["dk"]], d2=k["d2"].clone(), xk=k["xk"],
                       N=N, Cd=Cd, Cx=Cx, P=P, d_amax=k.get("d_amax"), x_ub=k.get("x_ub")))
    return r
E.pw_wgrad = spy_wg
out = m(dev(x), batch_positions=dev(dates))
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
l.backward()
hb.call = orig; E.hb.call = orig
# the fp64 oracle's gradient at the same place (h1 of the first encoder block = the last wgrad call), frame by frame
state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
from gpu_util import pool_branch
cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **kw)
pidx, _ = pool_branch(m, state, x, dates, cfg)
ORC = {}
for dtype in (torch.float64, torch.float32):
    pt = {k: (v.clone().to(dtype).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
              else (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in state.items()}
    taps = {}
    o = orc.forward(pt, x.to(dtype), dates.to(dtype), cfg, training=True, pool_idx=pidx, taps=taps)
    taps["in_block.0.h1"].retain_grad()
    orc.loss_from_output(o, y.to(dtype), cfg).backward()
    ORC[dtype] = taps["in_block.0.h1"].grad.double().reshape(B * T, -1, H * W)
print("oracle d(h1) of in_block.0 per frame, max |.|: fp64", " ".join(f"{float(v):.1e}" for v in ORC[torch.float64].abs().amax(dim=(1, 2))))
print("                                               fp32", " ".join(f"{float(v):.1e}" for v in ORC[torch.float32].abs().amax(dim=(1, 2))))
for i, a in enumerate(calls):
    wpart, nbx, cop, cip, W1, part_b, NPB, part_f, NPF, c1, c2, c3, cmu, A0, B0, part0, dW1, N, Ch, C, P = a[:21]
    wp = wpart.double().view(N, nbx, cop, cip)[:, :, :Ch, :C].sum(1)            # R[n,k,c]
    R = wp.float().double()                                                       # (the kernel rounds R to fp32 once)
    sb = part_b.double().view(N * Ch, NPB, 2)[:, :, 0].sum(1)
    sf = part_f.double().view(N * Ch, NPF, 2)[:, :, 0].sum(1) if part_f is not None else torch.zeros_like(sb)
    mu = cmu.double() if cmu is not None else torch.zeros_like(sb)
    S = (c1.double() * sb + c2.double() * (sf - mu * P) + c3.double() * P).view(N, Ch)
    ref = torch.einsum("nc,nkc->kc", A0.double().view(N, C), R) + torch.einsum("nc,nk->kc", B0.double().view(N, C), S)
    err = float((dW1.double() - ref).abs().max() / ref.abs().max())
    t1, t2 = torch.einsum("nc,nkc->kc", A0.double().view(N, C), R), torch.einsum("nc,nk->kc", B0.double().view(N, C), S)
    print(f"call {i}: N={N} Ch={Ch} C={C} P={P} nbx={nbx} NPB={NPB} NPF={NPF} cmu={'yes' if cmu is not None else 'no'}: dW1 vs fp64 of its inputs "
          f"{err:.2e}; |A0*R| max {float(t1.abs().max()):.3e} |B0*S| max {float(t2.abs().max()):.3e} |dW1| max {float(ref.abs().max()):.3e}; "
          f"|S| max {float(S.abs().max()):.3e} |sum du1| max {float(sb.abs().max()):.3e}")

for i, (a, w) in enumerate(zip(calls, wg)):
    wpart, nbx, cop, cip = a[:4]
    N, Ch, C, P = w["N"], w["Cd"], w["Cx"], w["P"]
    R = wpart.double().view(N, nbx, cop, cip)[:, :, :Ch, :C].sum(1)
    c1, c2, c3, mu = (t.double().view(N, Ch, 1) if t is not None else 0.0 for t in (w["dk"] + [None])[:4])
    d, h1, xx = w["d"].double().view(N, Ch, P), w["d2"].double().view(N, Ch, P), w["x"].double().view(N, C, P)
    dn = c1 * d + c2 * (h1 - mu) + c3
    Rr = torch.einsum("nkp,ncp->nkc", dn, xx)
    per = (R - Rr).abs().amax(dim=(1, 2)) / Rr.abs().amax()
    if i == len(calls) - 1:
        print("HIP d(h1) of in_block.0 per frame, max |.|:       ", " ".join(f"{float(v):.1e}" for v in dn.abs().amax(dim=(1, 2))))
        print("   error vs the fp64 oracle per frame (of the largest):", " ".join(f"{float(v):.1e}" for v in (dn.cpu() - ORC[torch.float64]).abs().amax(dim=(1, 2)) / ORC[torch.float64].abs().max()))
    print(f"wgrad call {i}: R vs fp64 of its inputs: max {float(per.max()):.2e} (frames: {' '.join(f'{float(v):.1e}' for v in per)}); |d| max {float(d.abs().max()):.3e} "
          f"|dn| max {float(dn.abs().max()):.3e} |c1*d| max {float((c1 * d).abs().max()):.3e} |x| max {float(xx.abs().max()):.3e}; d_amax given: {w['d_amax'] is not None}")
