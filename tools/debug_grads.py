"""Per-stage gradient comparison HIP vs fp64 oracle (debug aid; run on the GPU box)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from conftest import load_golden, rel_err
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src import losses

name = sys.argv[1] if len(sys.argv) > 1 else "g1_iso_t6"
g = load_golden(name)
meta = json.loads(str(g["meta"])); cov = meta["covmode"]
state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
x, y, d = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
cfg = orc.OracleConfig(covmode=cov, out_conv=[13 + (13 if cov == "diag" else 1)], attn_dropout=0.0)

def run_oracle(dt):
    pt = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.to(dt) if v.dtype.is_floating_point else v.clone())) for k, v in state.items()}
    xg = x.to(dt).clone().requires_grad_(True)
    taps = {}
    out = orc.forward(pt, xg, d.to(dt), cfg, training=True, taps=taps)
    for k in ("a0", "e", "agg", "attn", "dec0", "dec4", "pre_head", "c0"):
        taps[k].retain_grad()
    loss = orc.loss_from_output(out, y.to(dt), cfg); loss.backward()
    return {k: taps[k].grad for k in ("a0", "e", "agg", "attn", "dec0", "dec4", "pre_head", "c0")}, xg.grad, taps

g64, dx64, t64 = run_oracle(torch.float64)
g32, dx32, t32 = run_oracle(torch.float32)

m = U.UNCRTAINTS(input_dim=15, out_conv=[13 + (13 if cov == "diag" else 1)], out_nonlin_mean=True, out_nonlin_var="softplus", covmode=cov, scale_by=1.0)
m.load_state_dict(state); m.temporal_aggregator.attn_dropout.p = 0.0
m = m.cuda().train()
grads = {}
def hook(name):
    def h(gr): grads[name] = gr.detach().cpu()
    return h
# monkeypatch to tap intermediate tensors
orig_in = m.in_conv.smart_forward
def in_sf(inp):
    o = orig_in(inp); o.register_hook(hook("a0")); return o
m.in_conv.smart_forward = in_sf
orig_blk = m.in_block[0].forward
def blk_f(inp):
    o = orig_blk(inp); o.register_hook(hook("e")); return o
m.in_block[0].forward = blk_f
for i in (0, 4):
    def mk(i):
        orig = m.out_block[i].smart_forward
        def f(inp):
            if i == 0: inp.register_hook(hook("agg"))
            o = orig(inp); o.register_hook(hook(f"dec{i}")); return o
        m.out_block[i].smart_forward = f
    mk(i)
xg = x.cuda().requires_grad_(True)
out = m(xg, batch_positions=d.cuda())
crit = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode=cov)
l, _ = crit(out[:, :, :13], y.cuda(), out[:, :, 13:m.vars_idx]); l.backward()
B, T = x.shape[:2]
for k in ("dec4", "dec0", "agg", "e", "a0"):
    ref64 = g64[k]; ref32 = g32[k]
    got = grads[k].reshape(ref64.shape).double()
    print(f"{k:6s} hip-vs-64 {rel_err(got.numpy(), ref64.numpy()):.3e}  cpu32-vs-64 {rel_err(ref32.double().numpy(), ref64.numpy()):.3e}")
print(f"dx     hip-vs-64 {rel_err(xg.grad.cpu().double().numpy(), dx64.numpy()):.3e}  cpu32-vs-64 {rel_err(dx32.double().numpy(), dx64.numpy()):.3e}")
# per-frame dx error
for b in range(B):
    for t in range(T):
        print(b, t, f"{rel_err(xg.grad[b,t].cpu().double().numpy(), dx64[b,t].numpy()):.3e}", f"max|dx64|={dx64[b,t].abs().max():.3e}")
