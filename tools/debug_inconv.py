"""in_conv's weight gradient at an odd size: the weight-gradient GEMM against fp64 of its own inputs, and its inputs' tails (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import dev
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U
kw = eval("dict(" + sys.argv[1] + ")")
B, T, H, W = (int(v) for v in sys.argv[2].split(","))
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=347)
torch.manual_seed(247)
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
m.temporal_aggregator.attn_dropout.p = 0.0
m = m.to("cuda").train()
cap = []
_pw = E.pw_wgrad
def spy(d, x_, N, Cd, Cx, P, **k):
    r = _pw(d, x_, N, Cd, Cx, P, **k)
    if Cx == 15:
        cap.append(dict(d=d.clone(), x=x_.clone(), k=k, N=N, Cd=Cd, Cx=Cx, P=P, out=[t.clone() if t is not None else None for t in r], geom=E.current_geom()))
    return r
E.pw_wgrad = spy
out = m(dev(x), batch_positions=dev(dates))
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
l.backward()
for c in cap:
    N, Cd, Cx, P, k, g = c["N"], c["Cd"], c["Cx"], c["P"], c["k"], c["geom"]
    d, xx = c["d"].double().view(N, Cd, P), c["x"].double().view(N, Cx, P)
    print("geom", g, "pro_d", k.get("pro_d"), "rowsum", k.get("rowsum"), "per_frame", k.get("per_frame"))
    if k.get("pro_d") == E.PRO_NORMBWD:
        c1, c2, c3 = (t.double().view(N, Cd, 1) for t in k["dk"][:3])
        mu = k["dk"][3].double().view(N, Cd, 1) if len(k["dk"]) > 3 and k["dk"][3] is not None else 0.0
        dn = c1 * d + c2 * (k["d2"].double().view(N, Cd, P) - mu) + c3
    else:
        dn = d
    ref = torch.einsum("nkp,ncp->kc", dn, xx)
    got = c["out"][0].double().reshape(-1, Cd, Cx).sum(0) if c["out"][0].dim() == 3 else c["out"][0].double()
    print(f"   dW vs fp64 of its inputs: {float((got - ref).abs().max() / ref.abs().max()):.2e}")
    if g is not None:
        vp = g.P
        print(f"   tails: |x| {float(xx[:, :, vp:].abs().max()):.2e} |d| {float(d[:, :, vp:].abs().max()):.2e} |d2| "
              f"{float(k['d2'].double().view(N, Cd, P)[:, :, vp:].abs().max()) if k.get('d2') is not None else 0:.2e}")
        refv = torch.einsum("nkp,ncp->kc", dn[:, :, :vp], xx[:, :, :vp])
        print(f"   dW vs fp64 over the valid pixels only: {float((got - refv).abs().max() / refv.abs().max()):.2e}")

# the oracle's gradient w.r.t. c0 (in_conv's pre-norm output) against HIP's dn, and the pieces of dn
from gpu_util import pool_branch
state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **kw)
pidx, _ = pool_branch(m, state, x, dates, cfg)
pt = {k: (v.clone().double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.clone().double() if v.dtype.is_floating_point else v.clone())) for k, v in state.items()}
taps = {}
o = orc.forward(pt, x.double(), dates.double(), cfg, training=True, pool_idx=pidx, taps=taps)
taps["c0"].retain_grad(); taps["a0"].retain_grad()
orc.loss_from_output(o, y.double(), cfg).backward()
c = cap[-1]
N, Cd, P, g = c["N"], c["Cd"], c["P"], c["geom"]
vp = g.P if g is not None else P
k = c["k"]
d = c["d"].double().view(N, Cd, P)[:, :, :vp].cpu()
c0h = k["d2"].double().view(N, Cd, P)[:, :, :vp].cpu()
c1, c2, c3 = (t.double().view(N, Cd, 1).cpu() for t in k["dk"][:3])
mu = k["dk"][3].double().view(N, Cd, 1).cpu() if len(k["dk"]) > 3 and k["dk"][3] is not None else 0.0
dn = c1 * d + c2 * (c0h - mu) + c3
go = taps["c0"].grad.reshape(N, Cd, vp)
da0 = taps["a0"].grad.reshape(N, Cd, vp)
c0o = taps["c0"].detach().reshape(N, Cd, vp)
print(f"c0: hip vs fp64 {float((c0h - c0o).abs().max() / c0o.abs().max()):.2e}")
print(f"d(c0) = dn: hip vs fp64 {float((dn - go).abs().max() / go.abs().max()):.2e}")
# du0 = d(a0) * relu mask
A = m.in_conv.conv.conv[1]
mask = (taps["a0"].detach().reshape(N, Cd, vp) > 0).double()
du0o = da0 * mask
print(f"du0 (masked gradient of a0): hip vs fp64 {float((d - du0o).abs().max() / du0o.abs().max()):.2e}; mask disagreements {int(((d != 0) != (du0o != 0)).sum())}")
# instance-norm backward coefficients from the oracle's own du0, c0
mu_o = c0o.mean(-1, keepdim=True); var_o = c0o.var(-1, unbiased=False, keepdim=True); rstd_o = 1 / torch.sqrt(var_o + 1e-5)
c1o = rstd_o.expand(N, Cd, 1); c2o = -rstd_o ** 3 * (du0o * (c0o - mu_o)).mean(-1, keepdim=True); c3o = -rstd_o * du0o.mean(-1, keepdim=True)
for nm, h, r in (("c1", c1, c1o), ("c2", c2, c2o), ("c3", c3, c3o)):
    print(f"   {nm}: hip vs fp64 {float((h - r).abs().max() / r.abs().max()):.2e} (|.| max {float(r.abs().max()):.3e})")
