"""encoder_norm='instance' with a zero-padded date: is the padded frame exactly zero behind every stage, and where do the gradients leave
the oracle's?  (GPU box)   python tools/debug_instance_pad.py "agg_mode='att_mean', encoder_norm='instance', decoder_widths=[128]" 1,2,40,100"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import rel_err
from gpu_util import dev, oracle_run, pool_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

kw = eval("dict(" + sys.argv[1] + ")") if len(sys.argv) > 1 else dict(encoder_norm="instance")
B, T, H, W = (int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 40, 100)
cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **kw)
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=403)
if "--nopad" not in sys.argv:
    x[B - 1, T - 1] = 0.0
torch.manual_seed(303)
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
state = {k: v.detach().clone() for k, v in m.state_dict().items()}
m.temporal_aggregator.attn_dropout.p = 0.0
m = m.to("cuda").train()
m.keep_boundaries = True
# spy: every mbconv forward's saved tensors of the padded frame, every backward's dy of the padded frame
_mf, _mb = E.mbconv_forward, E.mbconv_backward
log = []


def spy_mf(xx, p, spec, training, *a, **k):
    r = _mf(xx, p, spec, training, *a, **k)
    sv = r[1]
    N = xx.shape[0]
    if N == B * T and "h1" in sv:
        f = N - 1
        pl = lambda t: t.reshape(N, -1)[f]
        log.append("fwd padded frame: |x| %.3e |h1| %.3e |h2| %.3e |h3| %.3e |y| %.3e ; A0 %.3e B0 %.3e A1 %.3e B1 %.3e s %s" % (
            float(pl(sv["x"]).abs().max()), float(pl(sv["h1"]).abs().max()), float(pl(sv["h2"]).abs().max()),
            float(pl(sv["h3"]).abs().max()), float(pl(r[0]).abs().max()), float(sv["n0"].A.view(N, -1)[f].abs().max()),
            float(sv["n0"].B.view(N, -1)[f].abs().max()), float(sv["n1"].A.view(N, -1)[f].abs().max()),
            float(sv["n1"].B.view(N, -1)[f].abs().max()), sv["s"].view(N, -1)[f][:3].tolist()))
    return r


def spy_mb(dy, sv, p, *a, **k):
    N = sv["dims"][0]
    if N == B * T:
        log.append("bwd padded frame: |dy| %.3e (other frames %.3e)" % (float(dy.reshape(N, -1)[N - 1].abs().max()), float(dy.reshape(N, -1)[:N - 1].abs().max())))
    r = _mb(dy, sv, p, *a, **k)
    if N == B * T and r[0] is not None:
        log.append("bwd padded frame: |dx| %.3e (other frames %.3e)" % (float(r[0].reshape(N, -1)[N - 1].abs().max()), float(r[0].reshape(N, -1)[:N - 1].abs().max())))
    return r


E.mbconv_forward, E.mbconv_backward = spy_mf, spy_mb
if "--nopart" in sys.argv:          # the backward statistics riding on the gradients are dropped: every consumer recomputes them
    E.claim_part = lambda *a, **k: None
for a in sys.argv:
    if a.startswith("--dev="):
        E.dev_options(**{kv.split("=")[0]: bool(int(kv.split("=")[1])) for kv in a[6:].split(",")}).__enter__()
out = m(dev(x), batch_positions=dev(dates))
m._boundary_enc.retain_grad()
m._boundary_agg.retain_grad()
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
l.backward()
for s in log:
    print(s)
pidx, _ = pool_branch(m, state, x, dates, cfg)
taps = {}
ot, lo, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx)
_, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx)
print("out", rel_err(out.detach().cpu().numpy(), ot.numpy()), "loss", l.item(), lo.item())
pt = {k: ((v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v.clone()) for k, v in state.items()}
e64 = orc.forward(pt, x.double(), dates.double(), cfg, training=True, taps=taps, pool_idx=pidx)
taps["e"].retain_grad(); taps["agg"].retain_grad()
orc.loss_from_output(e64, y.double(), cfg).backward()
de_h = m._boundary_enc.grad.reshape(B * T, -1, H * W).double().cpu()
de_o = taps["e"].grad.reshape(B * T, -1, H * W)
for f in range(B * T):
    print("frame %d: d(e) rel err %.3e (|d e| %.3e)" % (f, float((de_h[f] - de_o[f]).abs().max() / de_o.abs().max()), float(de_o[f].abs().max())))
print("d(agg) rel err %.3e" % float((m._boundary_agg.grad.double().cpu().reshape(-1) - taps["agg"].grad.reshape(-1)).abs().max() / taps["agg"].grad.abs().max()))
print("oracle fp64 padded frame: |a0| %.3e |e| %.3e" % (float(taps["a0"][B * T - 1].abs().max()), float(taps["e"][B * T - 1].abs().max())))
for k, v in m.named_parameters():
    if v.grad is None or g64.get(k) is None or float(g64[k].abs().max()) < 1e-9:
        continue
    eh, ec = rel_err(v.grad.cpu().numpy(), g64[k].numpy()), rel_err(g32[k].numpy(), g64[k].numpy())
    if eh > 1e-4:
        a, b = v.grad.double().cpu().reshape(-1), g64[k].reshape(-1)
        print(f"  {eh:.2e} cpu {ec:.2e} {k}: |hip| {float(a.norm()):.3e} |oracle| {float(b.norm()):.3e} cos {float(a @ b / (a.norm() * b.norm())):.4f} "
              f"|hip - oracle| {float((a - b).norm()):.3e}")

if "--fd" in sys.argv:
    # directional finite differences of the HIP forward itself along the two candidate gradients of one weight: which one is the slope?
    crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
    for name in ("in_block.0.conv.fn.7.weight", "in_block.0.conv.fn.0.weight"):
        prm = dict(m.named_parameters())[name]
        gh, go = prm.grad.detach().double().clone(), g64[name].double().to(prm.device).reshape(prm.shape)
        w0 = prm.detach().clone()

        def loss_at(w):
            with torch.no_grad():
                prm.copy_(w)
                o = m(dev(x), batch_positions=dev(dates))
                return float(crit(o[:, :, :13], dev(y), o[:, :, 13:26])[0])

        for tag, v in (("hip", gh), ("oracle", go)):
            u = (v / v.norm()).float()
            for eps in (1e-2, 3e-3):
                lp, lm = loss_at(w0 + eps * u), loss_at(w0 - eps * u)
                print(f"{name}: along the {tag} gradient, eps {eps:g}: finite difference {(lp - lm) / (2 * eps):+.5e}; <g_hip, u> {float((gh * u.double()).sum()):+.5e} "
                      f"<g_oracle, u> {float((go * u.double()).sum()):+.5e}")
        with torch.no_grad():
            prm.copy_(w0)
