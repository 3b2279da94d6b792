"""Non-default widths / input dims against the oracle: either parity or a loud error (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src import losses
def rel(a, b): return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
for name, okw, mkw in [("input_dim=13", dict(input_dim=13), dict(input_dim=13)),
                       ("widths 64", dict(encoder_widths=[64], decoder_widths=[64] * 5), dict(encoder_widths=[64], decoder_widths=[64] * 5)),
                       ("widths 32", dict(encoder_widths=[32], decoder_widths=[32] * 2), dict(encoder_widths=[32], decoder_widths=[32] * 2)),
                       ("widths 96", dict(encoder_widths=[96], decoder_widths=[96] * 2), dict(encoder_widths=[96], decoder_widths=[96] * 2)),
                       ("2 decoder blocks", dict(decoder_widths=[128] * 2), dict(decoder_widths=[128] * 2)),
                       ("n_head 8", dict(n_head=8), dict(n_head=8)), ("n_head 4", dict(n_head=4), dict(n_head=4)),
                       ("n_head 32", dict(n_head=32), dict(n_head=32))]:
    try:
        cfg = orc.OracleConfig(attn_dropout=0.0, **okw)
        p = orc.init_params(cfg, seed=4)
        ind = okw.get("input_dim", 15)
        x, y, dates = orc.synthetic_batch(1, 3, 64, 64, seed=5)
        x = x[:, :, :ind].contiguous()
        mk = dict(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0); mk.update(mkw)
        m = U.UNCRTAINTS(**mk); m.load_state_dict(p, strict=True); m.temporal_aggregator.attn_dropout.p = 0.0; m = m.cuda().train()
        pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
        o = orc.forward(pt, x, dates, cfg, training=True); orc.loss_from_output(o, y, cfg).backward()
        out = m(x.cuda(), batch_positions=dates.cuda())
        l, _ = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode="diag")(out[:, :, :13], y.cuda(), out[:, :, 13:26]); l.backward()
        gmax = max(v.grad.abs().max().item() for v in pt.values() if getattr(v, "grad", None) is not None)
        worst = max((rel(v.grad, pt[k].grad), k) for k, v in m.named_parameters() if pt[k].grad is not None and pt[k].grad.abs().max().item() > 1e-6 * gmax)
        print(f"{name}: out {rel(out.detach(), o.detach()):.2e}  worst grad {worst[0]:.2e} {worst[1]}")
    except Exception as e:
        print(f"{name}: {type(e).__name__}: {str(e)[:150]}")
