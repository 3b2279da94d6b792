"""Where does the outlier of tests/test_anysize.py (use_v, seed 0, random norm affines, one padded date) enter?  (GPU box)
Compares the gradient at the two stage boundaries (encoder output per frame, aggregated features) with the fp64 oracle on the same branch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import rel_err
from gpu_util import dev, pool_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

kw = eval("dict(" + sys.argv[1] + ")") if len(sys.argv) > 1 else dict(use_v=True)
B, T, H, W = (int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 50, 46)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **kw)
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=7 + seed)
x[B - 1, T - 1] = 0.0
torch.manual_seed(6 + seed)
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
g_ = torch.Generator().manual_seed(16)
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g_))
        mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g_))
    if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)) and mod.weight is not None:
        mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g_))
        mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g_))
state = {k: v.detach().clone() for k, v in m.state_dict().items()}
m.temporal_aggregator.attn_dropout.p = 0.0
if kw.get("use_v"):
    m.temporal_encoder.dropout.p = 0.0
m = m.to("cuda").train()
m.keep_boundaries = True
from uncrtaints_amd import engine as E
cap = {}
_ab, _vb, _ib = E.ltae_attention_backward, E.ltae_values_backward, E.include_v_backward


def spy_ab(datt, sv, p, n_head, d_k, dy1_extra=None):
    cap["datt_in"], cap["dy1_extra"] = datt.clone(), None if dy1_extra is None else dy1_extra.clone()
    r = _ab(datt, sv, p, n_head, d_k, dy1_extra=dy1_extra)
    cap["ddown"] = r[0].clone()
    return r


def spy_ib(dout, sv):
    r = _ib(dout, sv)
    cap["dg0"], cap["dv"] = r[0].clone(), r[1].clone()
    return r


_vf = E.ltae_values_forward


def spy_vf(*a, **k):
    v, saved = _vf(*a, **k)
    cap["val"] = saved
    return v, saved


E.ltae_attention_backward, E.include_v_backward, E.ltae_values_forward = spy_ab, spy_ib, spy_vf
xg = dev(x).requires_grad_(True)
out = m(xg, batch_positions=dev(dates))
m._boundary_enc.retain_grad()
m._boundary_agg.retain_grad()
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
l.backward()
pidx, _ = pool_branch(m, state, x, dates, cfg)
P = H * W
cut = lambda t, n, c: t.detach().reshape(n, c, -1)[:, :, :P].double().cpu()
res = {}
for dtype in (torch.float64, torch.float32):
    pt = {k: (v.clone().to(dtype).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
              else (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in state.items()}
    xo = x.to(dtype).clone().requires_grad_(True)
    taps = {}
    o = orc.forward(pt, xo, dates.to(dtype), cfg, training=True, pool_idx=pidx, taps=taps)
    for k in ("e", "agg", "down", "attn"):
        taps[k].retain_grad()
    orc.loss_from_output(o, y.to(dtype), cfg).backward()
    res[dtype] = dict(de=taps["e"].grad.reshape(B * T, -1, P).double(), dagg=taps["agg"].grad.reshape(B, -1, P).double(),
                      ddown=taps["down"].grad.double(), dattn=taps["attn"].grad.double(), dx=xo.grad.reshape(B * T, -1, P).double(), e=taps["e"].detach().reshape(B * T, -1, P).double())
C = res[torch.float64]["de"].shape[1]
hip = dict(de=cut(m._boundary_enc.grad, B * T, C), dagg=cut(m._boundary_agg.grad, B, C), dx=xg.grad.reshape(B * T, -1, P).double().cpu(),
           e=cut(m._boundary_enc, B * T, C))
t64, t32 = res[torch.float64], res[torch.float32]
print(f"encoder output e: hip {rel_err(hip['e'].numpy(), t64['e'].numpy()):.2e} cpu32 {rel_err(t32['e'].numpy(), t64['e'].numpy()):.2e}")
print(f"d(agg): hip {rel_err(hip['dagg'].numpy(), t64['dagg'].numpy()):.2e} cpu32 {rel_err(t32['dagg'].numpy(), t64['dagg'].numpy()):.2e}")
for f in range(B * T):
    sc = float(t64["de"].abs().max())
    eh = float((hip["de"][f] - t64["de"][f]).abs().max()) / sc
    ec = float((t32["de"][f] - t64["de"][f]).abs().max()) / sc
    mag = float(t64["de"][f].abs().max()) / sc
    spread = float(hip["e"][f].std(dim=-1).max())
    print(f"d(e) frame {f}: |.|max {mag:.2e} of the largest; error hip {eh:.2e} cpu32 {ec:.2e}; pixel-to-pixel std of e (hip) {spread:.2e}")
    sx = float(t64["dx"].abs().max())
    print(f"   dx frame {f}: hip {float((hip['dx'][f] - t64['dx'][f]).abs().max()) / sx:.2e} cpu32 {float((t32['dx'][f] - t64['dx'][f]).abs().max()) / sx:.2e}")

dd_h = cap["ddown"].double().cpu().reshape(t64["ddown"].shape)
sc = float(t64["ddown"].abs().max())
for f in range(B * T):
    b, t = divmod(f, T)
    print(f"d(down) frame {f}: |.|max {float(t64['ddown'][b, t].abs().max()) / sc:.2e}; error hip {float((dd_h[b, t] - t64['ddown'][b, t]).abs().max()) / sc:.2e} "
          f"cpu32 {float((t32['ddown'][b, t] - t64['ddown'][b, t]).abs().max()) / sc:.2e}")
da_h = cap["datt_in"].double().cpu()
print("attention gradient reaching the attention backward:", tuple(da_h.shape), "oracle", tuple(t64["dattn"].shape))
if da_h.numel() == t64["dattn"].numel():
    da_h = da_h.reshape(t64["dattn"].shape)
    sa = float(t64["dattn"].abs().max())
    for t in range(T):
        print(f"d(attn) date {t}: |.|max {float(t64['dattn'][:, :, t].abs().max()) / sa:.2e}; hip {float((da_h - t64['dattn'])[:, :, t].abs().max()) / sa:.2e} "
              f"cpu32 {float((t32['dattn'] - t64['dattn'])[:, :, t].abs().max()) / sa:.2e}; hip |.|max {float(da_h[:, :, t].abs().max()) / sa:.2e}")
# the dense part of d(e): everything except the arg-max pixels
idx = pidx.reshape(B * T, C, -1).to(torch.long)
on = torch.zeros(B * T, C, P, dtype=torch.bool)
on.scatter_(2, idx, True)
for f in range(B * T):
    sc = float(t64["de"].abs().max())
    d = (hip["de"][f] - t64["de"][f]).abs()
    print(f"d(e) frame {f}: error off the arg-max pixels {float(d[~on[f]].max()) / sc:.2e}, on them {float(d[on[f]].max()) / sc:.2e}")

if "val" in cap:
    sv = cap["val"]
    m1 = sv["m1"].double()                      # [B, C, S] pre-norm output of the value MLP's Linear
    nf = sv["nf"]
    mu, var = m1.mean(dim=(0, 2)), m1.var(dim=(0, 2), unbiased=False)
    ratio = mu.abs() / var.sqrt()
    k = int(ratio.argmax())
    print(f"value MLP BatchNorm input: max |mean| / std over channels {float(ratio.max()):.1f} (channel {k}: mean {float(mu[k]):.4e}, std {float(var[k].sqrt()):.4e})")
    rstd64 = 1.0 / torch.sqrt(var + 1e-5)
    print(f"   mean error {float(((nf.mean.double() - mu).abs() / var.sqrt()).max()):.2e} (in stds); rstd relative error max "
          f"{float(((nf.rstd.double() - rstd64).abs() / rstd64).max()):.2e} at channel {int(((nf.rstd.double() - rstd64).abs() / rstd64).argmax())}")
