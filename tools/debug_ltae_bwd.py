"""Isolate the accuracy of the L-TAE stage backward pieces on the g1_iso_t6 fixture (run on the GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from conftest import load_golden, rel_err
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E

g = load_golden("g1_iso_t6")
state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
x, y, d = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
cfg = orc.OracleConfig(covmode="iso", out_conv=[14], attn_dropout=0.0)

def run(dt):
    pt = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.to(dt).clone() if v.dtype.is_floating_point else v.clone())) for k, v in state.items()}
    taps = {}
    out = orc.forward(pt, x.to(dt), d.to(dt), cfg, training=True, taps=taps)
    for k in ("e", "agg", "attn", "down"): taps[k].retain_grad()
    orc.loss_from_output(out, y.to(dt), cfg).backward()
    return taps, pt
t64, p64 = run(torch.float64)
t32, p32 = run(torch.float32)
B, T = x.shape[:2]
dev = "cuda"
f = lambda t: t.detach().float().contiguous().to(dev)
pad = torch.zeros(B, T, dtype=torch.int32, device=dev)
# (1) aggregate backward with fp64-truth inputs cast to fp32
e5 = t64["e"].reshape(B, T, 128, 64, 64)
gfw, sv, _ = E.aggregate_forward(f(e5), f(t64["attn"]), pad, False, 0.0, 1, None)
print("agg fwd err", rel_err(gfw.cpu().double().numpy(), t64["agg"].detach().numpy()))
de, datt = E.aggregate_backward(f(t64["agg"].grad), sv)
# truth: d e from aggregation only = e.grad minus maxpool path; compare datt instead
print("datt: hip-vs-64", rel_err(datt.cpu().double().numpy(), t64["attn"].grad.numpy()), " cpu32-vs-64", rel_err(t32["attn"].grad.double().numpy(), t64["attn"].grad.numpy()))
# (2) attention backward given truth datt
p = dict(in_norm_w=f(state["temporal_encoder.in_norm.weight"]), in_norm_b=f(state["temporal_encoder.in_norm.bias"]),
         inconv_w=f(state["temporal_encoder.inconv.weight"]), inconv_b=f(state["temporal_encoder.inconv.bias"]),
         fc_w=f(state["temporal_encoder.attention_heads.fc1_k.weight"]), fc_b=f(state["temporal_encoder.attention_heads.fc1_k.bias"]),
         Q=f(state["temporal_encoder.attention_heads.Q"]))
denom = torch.pow(torch.tensor(1000.0), 2 * (torch.arange(16).float() // 2) / 16).to(dev)
att, sva = E.ltae_attention_forward(f(t64["down"]), f(d), pad, p, denom, 16, 4)
print("att fwd err", rel_err(att.cpu().double().numpy(), t64["attn"].detach().numpy()))
ddown, gr = E.ltae_attention_backward(f(t64["attn"].grad), sva, p, 16, 4)
E.join_side()
print("ddown: hip-vs-64", rel_err(ddown.cpu().double().numpy().reshape(-1), t64["down"].grad.numpy().reshape(-1)), " cpu32-vs-64", rel_err(t32["down"].grad.double().numpy(), t64["down"].grad.numpy()))
dd = (ddown.cpu().double().reshape(B, T, 128, 32, 32) - t64["down"].grad).abs()
print("per-frame ddown err / global max:", [float(dd[0, t].max() / t64["down"].grad.abs().max()) for t in range(T)])
for k_h, k_o in (("Q", "temporal_encoder.attention_heads.Q"), ("fc_w", "temporal_encoder.attention_heads.fc1_k.weight"), ("inconv_w", "temporal_encoder.inconv.weight"), ("in_norm_w", "temporal_encoder.in_norm.weight")):
    print(k_h, "hip-vs-64", rel_err(gr[k_h].cpu().double().numpy().reshape(-1), p64[k_o].grad.numpy().reshape(-1)), "cpu32-vs-64", rel_err(p32[k_o].grad.double().numpy(), p64[k_o].grad.numpy()))
