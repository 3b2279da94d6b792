"""Non-default widths / input dims against the oracle: either parity or a loud error (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src import losses
def rel(a, b): return ((a.double().cpu() - b.double()).abs().max() / b.double().abs().max()).item()
CASES = [("scale_by 10", dict(scale_by=10.0)), ("no positional encoding", dict(positional_encoding=False)),
         ("d_model 128", dict(d_model=128)), ("d_k 8", dict(d_k=8)), ("mean without sigmoid", dict(out_nonlin_mean=False)),
         ("pad_value 1", dict(pad_value=1.0)), ("use_v widths 64", dict(use_v=True, encoder_widths=[64], decoder_widths=[64] * 2)),
         ("d_model 512", dict(d_model=512))]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] == sys.argv[1]]
for name, okw in CASES:
    mkw = okw
    try:
        cfg = orc.OracleConfig(attn_dropout=0.0, **okw)
        p = orc.init_params(cfg, seed=4)
        ind = okw.get("input_dim", 15)
        x, y, dates = orc.synthetic_batch(2, 3, 64, 64, seed=5)
        if okw.get('pad_value') == 1.0: x[1, 2] = 1.0
        if okw.get('use_v'): cfg.ltae_dropout = 0.0
        if okw.get('scale_by', 1.0) != 1.0: y = y * okw['scale_by']
        x = x[:, :, :ind].contiguous()
        mk = dict(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0); mk.update(mkw)
        m = U.UNCRTAINTS(**mk); m.load_state_dict(p, strict=True); m.temporal_aggregator.attn_dropout.p = 0.0
        if okw.get('use_v'): m.temporal_encoder.dropout.p = 0.0
        m = m.cuda().train()
        pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
        o = orc.forward(pt, x, dates, cfg, training=True); orc.loss_from_output(o, y, cfg).backward()
        out = m(x.cuda(), batch_positions=dates.cuda())
        l, _ = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode="diag")(out[:, :, :13], y.cuda(), out[:, :, 13:26]); l.backward()
        gmax = max(v.grad.abs().max().item() for v in pt.values() if getattr(v, "grad", None) is not None)
        worst = max((rel(v.grad, pt[k].grad), k) for k, v in m.named_parameters() if pt[k].grad is not None and pt[k].grad.abs().max().item() > 1e-6 * gmax)
        print(f"{name}: out {rel(out.detach(), o.detach()):.2e}  worst grad {worst[0]:.2e} {worst[1]}", flush=True)
    except Exception as e:
        print(f"{name}: {type(e).__name__}: {str(e)[:150]}", flush=True)
