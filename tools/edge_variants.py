"""Variants at edge shapes against the fp64 oracle (run on the GPU box): python tools/edge_variants.py [variants]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src import losses
def rel(a, b): return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
iso = dict(covmode="iso", out_conv=[14])
uv64 = dict(use_v=True, encoder_widths=[64], decoder_widths=[64] * 2)
cases = [("use_v", uv64, uv64, (2, 3, 64, 64)), ("use_v", dict(use_v=True), dict(use_v=True), (2, 3, 64, 64))]
if len(sys.argv) > 1 and sys.argv[1] == "variants":
    cases = [("use_v", dict(use_v=True), dict(use_v=True), (2, 3, 80, 64)),
             ("residual", dict(block_type="residual", decoder_widths=[128, 128]), dict(block_type="residual", decoder_widths=[128, 128]), (1, 2, 64, 128)),
             ("att_mean", dict(agg_mode="att_mean"), dict(agg_mode="att_mean"), (2, 3, 80, 64)),
             ("mean", dict(agg_mode="mean"), dict(agg_mode="mean"), (2, 3, 48, 128)),
             ("iso", iso, iso, (1, 6, 96, 64)),
             ("instance", dict(encoder_norm="instance", decoder_norm="instance"), dict(encoder_norm="instance", decoder_norm="instance"), (1, 3, 64, 96))]
for name, okw, mkw, (B, T, H, W) in cases:
    cfg = orc.OracleConfig(attn_dropout=0.0, **okw)
    p = orc.init_params(cfg, seed=11)
    x, y, dates = orc.synthetic_batch(B, T, H, W, seed=12)
    if name == "mean": x[1, 0] = 0
    mk = dict(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
    mk.update(mkw)
    m = U.UNCRTAINTS(**mk); m.load_state_dict(p, strict=True)
    if hasattr(m, "temporal_aggregator"): m.temporal_aggregator.attn_dropout.p = 0.0
    if name == "use_v": m.temporal_encoder.dropout.p = 0.0; cfg.ltae_dropout = 0.0
    m = m.cuda()
    res = []
    for training in (False, True):
        pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
        o = orc.forward(pt, x, dates, cfg, training=training)
        m.train(training)
        m.load_state_dict(p, strict=True)
        out = m(x.cuda(), batch_positions=dates.cuda())
        res.append(rel(out.detach(), o.detach()))
        if training:
            # gradients on the max-pool branch the HIP forward took (an arg-max flip at a near-tie is not an error)
            from gpu_util import pool_branch
            pidx, flips = pool_branch(m, p, x, dates, cfg) if not getattr(cfg, "is_mono", False) else (None, 0)
            pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
            o = orc.forward(pt, x, dates, cfg, training=True, pool_idx=pidx, update_running=False)
            lo = orc.loss_from_output(o, y, cfg); lo.backward()
            nv = 13 if cfg.covmode == "diag" else 1
            crit = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode=cfg.covmode)
            l, _ = crit(out[:, :, :13], y.cuda(), out[:, :, 13:13 + nv]); l.backward()
            p64 = {k: (v.double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.double().clone() if v.dtype.is_floating_point else v.clone())) for k, v in p.items()}
            o64 = orc.forward(p64, x.double(), dates.double(), cfg, training=True, pool_idx=pidx, update_running=False)
            orc.loss_from_output(o64, y.double(), cfg).backward()
            gmax = max(q.grad.abs().max().item() for q in p64.values() if getattr(q, "grad", None) is not None)
            rows = []
            for k, v in m.named_parameters():
                g64 = p64[k].grad
                if g64 is None or g64.abs().max().item() < 1e-6 * gmax: continue
                rows.append((rel(v.grad, g64), rel(pt[k].grad, g64), k))
            rows.sort(reverse=True)
            res.append(rows[:4])
    print(name, (B, T, H, W), "eval %.2e train %.2e" % (res[0], res[1]))
    for hip, cpu, k in res[2]: print("     hip-vs-fp64 %.2e  cpu32-vs-fp64 %.2e  %s" % (hip, cpu, k))
