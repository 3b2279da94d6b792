"""Encoder MBConv backward in isolation: gradient entering (d e), leaving (d a0) and the block's parameter gradients, HIP vs the
fp64 oracle.  usage: python tools/debug_enc_grad.py B T H W [param-seed batch-seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src import losses
B, T, H, W = (int(a) for a in sys.argv[1:5])
ps, bs = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (3, 4)
def rel(a, b): return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
torch.set_num_threads(16)
cfg = orc.OracleConfig(attn_dropout=0.0)
p = orc.init_params(cfg, seed=ps)
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=bs)
res = {}
for dt in (torch.float64, torch.float32):
    pt = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.to(dt).clone() if v.dtype.is_floating_point else v.clone())) for k, v in p.items()}
    taps = {}
    o = orc.forward(pt, x.to(dt), dates.to(dt), cfg, training=True, taps=taps)
    for k in ("e", "a0", "c0"): taps[k].retain_grad()
    orc.loss_from_output(o, y.to(dt), cfg).backward()
    res[dt] = dict(e=taps["e"].grad.clone(), a0=taps["a0"].grad.clone(), c0=taps["c0"].grad.clone(),
                   g={k: v.grad.clone() for k, v in pt.items() if getattr(v, "grad", None) is not None})
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
m.load_state_dict(p, strict=True); m.temporal_aggregator.attn_dropout.p = 0.0; m = m.cuda().train()
rec = {}
orig_mb, orig_ic = E.mbconv_backward, E.inconv_backward
def mb(dy, sv, pp, need_dx=True, dy_part=None):
    r = orig_mb(dy, sv, pp, need_dx, dy_part)
    if sv["dims"][0] == B * T:
        rec["de"], rec["da0"], rec["g"] = dy.detach().clone(), r[0].detach().clone(), {k: v.detach().clone() for k, v in r[1].items()}
    return r
E.mbconv_backward = mb
out = m(x.cuda(), batch_positions=dates.cuda())
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode="diag")(out[:, :, :13], y.cuda(), out[:, :, 13:26]); l.backward()
t64, t32 = res[torch.float64], res[torch.float32]
print("d/d(e)  : hip %.2e cpu32 %.2e" % (rel(rec["de"].view_as(t64["e"]), t64["e"]), rel(t32["e"], t64["e"])))
# the fused path hands back du0 = d a0 * relu mask: compare with d c0's pre-norm gradient is not available; use the mask of the oracle
da0 = rec["da0"].view_as(t64["a0"]).double().cpu()
mask = (t64["a0"] != 0)          # a0 = relu(.) : zero where the ReLU is off (the fused kernel applies that mask already)
print("d/d(a0) (masked): hip %.2e cpu32 %.2e" % (rel(da0 * mask, t64["a0"] * mask), rel(t32["a0"] * mask, t64["a0"] * mask)))
names = dict(n0w="conv.norm.weight", n0b="conv.norm.bias", w1="conv.fn.0.weight", n1w="conv.fn.1.weight", n1b="conv.fn.1.bias",
             wdw="conv.fn.3.weight", n2w="conv.fn.4.weight", n2b="conv.fn.4.bias", se1="conv.fn.6.fc.0.weight", se2="conv.fn.6.fc.2.weight",
             w2="conv.fn.7.weight", n3w="conv.fn.8.weight", n3b="conv.fn.8.bias")
for k, nm in names.items():
    full = "in_block.0." + nm
    print("  %-4s hip %.2e cpu32 %.2e" % (k, rel(rec["g"][k].reshape(t64["g"][full].shape), t64["g"][full]), rel(t32["g"][full], t64["g"][full])))
