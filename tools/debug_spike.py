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
PIN = "--pin" in sys.argv
if kw.get("use_v"):
    m.temporal_encoder.keep_relu_branch = True
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
_ew, _pg, _hc = E.ew, E.pw_gemm, E.hb.call
state_flag = {"in_vb": False}
_vb0 = E.ltae_values_backward


def spy_vb(*a, **k):
    state_flag["in_vb"] = True
    r = _vb0(*a, **k)
    state_flag["in_vb"] = False
    return r


def spy_ew(op, a, **k):
    r = _ew(op, a, **k)
    if state_flag["in_vb"] and op == E.EW_RELU_BWD:
        cap["dr_in"], cap["du"] = a.clone(), k["out"].clone()
    return r


def spy_pg(x, *a, **k):
    r = _pg(x, *a, **k)
    if state_flag["in_vb"] and k.get("pro") == E.PRO_NORMBWD:
        cap["dvh"] = r[0].clone()
        cap["kk"] = [t.clone() if t is not None else None for t in k["k"]]
    return r


E.ltae_values_backward, E.ew, E.pw_gemm = spy_vb, spy_ew, spy_pg
xg = dev(x).requires_grad_(True)
out = m(xg, batch_positions=dev(dates))
m._boundary_enc.retain_grad()
m._boundary_agg.retain_grad()
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
l.backward()
pidx, _ = pool_branch(m, state, x, dates, cfg)
vmask = None
if PIN:
    from gpu_util import value_relu_mask
    vmask = value_relu_mask(m)
    print("value-MLP ReLU mask pinned:", tuple(vmask["temporal_encoder.mlp"].shape), int(vmask["temporal_encoder.mlp"].sum()), "ones")
P = H * W
cut = lambda t, n, c: t.detach().reshape(n, c, -1)[:, :, :P].double().cpu()
res = {}
for dtype in (torch.float64, torch.float32):
    pt = {k: (v.clone().to(dtype).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
              else (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in state.items()}
    xo = x.to(dtype).clone().requires_grad_(True)
    taps = {}
    o = orc.forward(pt, xo, dates.to(dtype), cfg, training=True, pool_idx=pidx, taps=taps, relu_masks=vmask)
    for k in ("e", "agg", "down", "attn", "vals", "val_y", "val_vh", "val_m1", "val_r"):
        if k in taps:
            taps[k].retain_grad()
    orc.loss_from_output(o, y.to(dtype), cfg).backward()
    res[dtype] = dict(de=taps["e"].grad.reshape(B * T, -1, P).double(), dagg=taps["agg"].grad.reshape(B, -1, P).double(),
                      extra={k: taps[k].grad.double() for k in ("vals", "val_y", "val_vh", "val_m1", "val_r") if k in taps},
                      r_val=taps["val_r"].detach().double() if "val_r" in taps else None, m1_val=taps["val_m1"].detach().double() if "val_m1" in taps else None, ddown=taps["down"].grad.double(), dattn=taps["attn"].grad.double(), dx=xo.grad.reshape(B * T, -1, P).double(), e=taps["e"].detach().reshape(B * T, -1, P).double())
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

if "dv" in cap and "vals" in t64["extra"]:
    S = 32 * 32
    def cmp(tag, h, key):
        a, b = t64["extra"][key], t32["extra"][key]
        sc = float(a.abs().max())
        print(f"{tag}: |.|max {sc:.3e}; error hip {float((h - a).abs().max()) / sc:.2e} cpu32 {float((b - a).abs().max()) / sc:.2e}")
    cmp("d(values)", cap["dv"].double().cpu().reshape(t64["extra"]["vals"].shape), "vals")
    dy1 = cap["dy1_extra"].double().cpu()               # [B*T, D, S] -> [n = B*S, T, D]
    D = dy1.shape[1]
    cmp("d(y) through the values", dy1.reshape(B, T, D, S).permute(0, 3, 1, 2).reshape(B * S, T, D), "val_y")

    n = B * S
    Cv = cap["du"].shape[1]
    tr = lambda t: t.double().cpu().reshape(B, -1, S).permute(0, 2, 1).reshape(n, -1)       # [B, C, S] -> [n, C]
    cmp("d(r) = gradient after the out_norm GroupNorm backward", tr(cap["dr_in"]), "val_r")
    # d(m1): HIP keeps du (after the ReLU mask) and applies the BatchNorm backward as a prologue c1*du + c2*(m1 - mu) + c3
    kk = cap["kk"]
    c1, c2, c3 = (t.double().cpu().view(B, Cv, 1) for t in kk[:3])
    mu = kk[3].double().cpu().view(B, Cv, 1) if len(kk) > 3 and kk[3] is not None else 0.0
    m1h = cap["val"]["m1"].double().cpu()
    dm1 = c1 * cap["du"].double().cpu() + c2 * (m1h - mu) + c3
    cmp("d(m1) = gradient ahead of the value MLP's BatchNorm", tr(dm1), "val_m1")
    cmp("d(vh) = gradient of the attention-weighted values", tr(cap["dvh"]), "val_vh")

    # the BatchNorm backward's coefficients against fp64 formulas on HIP's own du and m1
    duh = cap["du"].double().cpu()
    gam = state["temporal_encoder.mlp.1.weight"].double().view(1, Cv, 1)
    mu64 = m1h.mean(dim=(0, 2), keepdim=True)
    var64 = m1h.var(dim=(0, 2), unbiased=False, keepdim=True)
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    c1r = (gam * rstd64).expand(B, Cv, 1)
    c2r = (-gam * rstd64 ** 3 * (duh * (m1h - mu64)).mean(dim=(0, 2), keepdim=True)).expand(B, Cv, 1)
    c3r = (-gam * rstd64 * duh.mean(dim=(0, 2), keepdim=True)).expand(B, Cv, 1)
    for nm, h, r in (("c1", c1, c1r), ("c2", c2, c2r), ("c3", c3, c3r)):
        k = int(((h - r).abs() / r.abs().max()).argmax())
        print(f"   {nm}: max |hip - fp64| / max|.| = {float((h - r).abs().max() / r.abs().max()):.2e} (worst channel {k % Cv}: hip {float(h.reshape(-1)[k]):.6e} fp64 {float(r.reshape(-1)[k]):.6e})")
    if torch.is_tensor(mu):
        print(f"   mu: max |hip - fp64| = {float((mu - mu64).abs().max()):.2e}")
    dm1r = c1r * duh + c2r * (m1h - mu64) + c3r
    sc = float(t64["extra"]["val_m1"].abs().max())
    print(f"   d(m1) from the fp64 coefficients on HIP's du, m1: error vs the oracle {float((tr(dm1r) - t64['extra']['val_m1']).abs().max()) / sc:.2e}")
    ch = int(((tr(dm1) - t64["extra"]["val_m1"]).abs().max(dim=0).values).argmax())
    print(f"   worst channel of d(m1): {ch}; |mean|/std of m1 there {float(mu64.view(-1)[ch].abs() / var64.view(-1)[ch].sqrt()):.1f}; "
          f"sum|du| {float(duh[:, ch].abs().sum()):.3e} vs |sum du| {float(duh[:, ch].sum().abs()):.3e}; max|du| {float(duh[:, ch].abs().max()):.3e}")

    nfv = cap["val"]["nf"]
    uh = nfv.A.double().cpu().view(B, Cv, 1) * m1h + nfv.B.double().cpu().view(B, Cv, 1)
    mask_h = tr(uh) > 0
    for nm, t in (("fp64", t64), ("fp32", t32)):
        mo = t["r_val"] > 0
        diff = (mask_h != mo)
        drm = t64["extra"]["val_r"].abs()
        print(f"   ReLU mask of the value MLP: hip vs oracle {nm}: {int(diff.sum())} of {diff.numel()} elements differ; |d(r)| there max "
              f"{float(drm[diff].max()) if diff.any() else 0.0:.3e} (max anywhere {float(drm.max()):.3e}); |u| there (hip) max {float(tr(uh).abs()[diff].max()) if diff.any() else 0.0:.3e}")
    print(f"   m1: hip vs fp64 max abs {float((tr(m1h) - t64['m1_val']).abs().max()):.3e}; fp32 vs fp64 {float((t32['m1_val'] - t64['m1_val']).abs().max()):.3e}")

    # channel `ch`: every ingredient of dm1 = g*rstd*(du - mean(du) - xhat*mean(du*xhat)), HIP's tensors against the fp64 oracle's
    mo = vmask["temporal_encoder.mlp"].double() if vmask is not None else (t64["r_val"] > 0).double()
    du_o = t64["extra"]["val_r"] * mo                       # [n, C]
    du_h = tr(duh)
    m1_o, m1_hh = t64["m1_val"], tr(m1h)
    for c_ in (ch,):
        a, b = du_h[:, c_], du_o[:, c_]
        xa, xb = m1_hh[:, c_], m1_o[:, c_]
        sa, sb = xa.std(unbiased=False), xb.std(unbiased=False)
        xha, xhb = (xa - xa.mean()) / torch.sqrt(sa ** 2 + 1e-5), (xb - xb.mean()) / torch.sqrt(sb ** 2 + 1e-5)
        print(f"   channel {c_}: du max abs diff {float((a - b).abs().max()):.3e} (max|du| {float(b.abs().max()):.3e}); mean(du) hip {float(a.mean()):.6e} "
              f"oracle {float(b.mean()):.6e}; mean(du*xhat) hip {float((a * xha).mean()):.6e} oracle {float((b * xhb).mean()):.6e}; "
              f"std hip {float(sa):.6e} oracle {float(sb):.6e}; xhat max abs diff {float((xha - xhb).abs().max()):.3e}")
        k_ = int((a - b).abs().argmax())
        print(f"      largest du difference at sample {k_}: hip {float(a[k_]):.6e} oracle {float(b[k_]):.6e}; d(r) hip {float(tr(cap['dr_in'])[k_, c_]):.6e} "
              f"oracle {float(t64['extra']['val_r'][k_, c_]):.6e}; u hip {float(tr(uh)[k_, c_]):.3e}; mask hip {bool(mask_h[k_, c_])} oracle {bool(mo[k_, c_])}")
