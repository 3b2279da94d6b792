"""Where does the encoder-side gradient deviate?  Compares d(loss)/d(e) (the gradient entering the last encoder block) and
d(loss)/d(agg) between the HIP path and the fp64 oracle for one shape.  usage: python tools/debug_stage_grad.py B T H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src import losses
B, T, H, W = (int(a) for a in sys.argv[1:5])
def rel(a, b): return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
cfg = orc.OracleConfig(attn_dropout=0.0)
p = orc.init_params(cfg, seed=11)
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=12)
res = {}
for dt in (torch.float64, torch.float32):
    pt = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.to(dt).clone() if v.dtype.is_floating_point else v.clone())) for k, v in p.items()}
    taps = {}
    o = orc.forward(pt, x.to(dt), dates.to(dt), cfg, training=True, taps=taps)
    for k in ("e", "agg", "down"): taps[k].retain_grad()
    orc.loss_from_output(o, y.to(dt), cfg).backward()
    res[dt] = {k: taps[k].grad.clone() for k in ("e", "agg", "down")}
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
m.load_state_dict(p, strict=True); m.temporal_aggregator.attn_dropout.p = 0.0; m = m.cuda().train()
rec = {}
orig_mb, orig_stage = E.mbconv_backward, E.ltae_stage_backward
def mb(dy, sv, pp, need_dx=True, dy_part=None):
    if sv["dims"][0] == B * T: rec["de"] = dy.detach().clone()
    return orig_mb(dy, sv, pp, need_dx, dy_part)
def st(dg, sv, pp, nh, dk):
    rec["dagg"] = dg.detach().clone()
    return orig_stage(dg, sv, pp, nh, dk)
E.mbconv_backward, E.ltae_stage_backward = mb, st
out = m(x.cuda(), batch_positions=dates.cuda())
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode="diag")(out[:, :, :13], y.cuda(), out[:, :, 13:26]); l.backward()
t64, t32 = res[torch.float64], res[torch.float32]
print("d/d(agg): hip %.2e cpu32 %.2e" % (rel(rec["dagg"], t64["agg"]), rel(t32["agg"], t64["agg"])))
de_h = rec["de"].view(B, T, 128, H, W)
t64["e"], t32["e"] = t64["e"].reshape(de_h.shape), t32["e"].reshape(de_h.shape)
print("d/d(e)  : hip %.2e cpu32 %.2e" % (rel(de_h, t64["e"]), rel(t32["e"], t64["e"])))
diff = (de_h.double().cpu() - t64["e"]).abs()
idx = torch.nonzero(diff > 1e-3 * t64["e"].abs().max())
print("elements off by > 1e-3 of max:", idx.shape[0], "of", diff.numel(), idx[:10].tolist())
