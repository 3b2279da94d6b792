"""Which part of the HIP path carries the noise of the ill-conditioned (cancellation-dominated) gradients?  Runs the golden
fixture's network under runtime switches (exact-split GEMMs on/off, row-streaming depthwise kernels on/off, fused dx on/off) and
prints, per setting, the parameters whose distance from the fp64 truth exceeds the CPU fp32 path's by the largest factor.
    python tools/probe_grad_noise.py [variant kwargs as key=value ...]      (UNCR_HIP_LIB selects a variant library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from conftest import load_golden, rel_err
from gpu_util import oracle_run, is_zero_grad
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E, hip_backend as hb
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

kw = {}
SIZE = (2, 3, 64, 64)
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "size":
        SIZE = tuple(int(t) for t in v.split("x"))
        continue
    kw[k] = int(v) if v.lstrip("-").isdigit() else v
torch.set_num_threads(16)
cfg = orc.OracleConfig(attn_dropout=0.0, **kw)
if kw or SIZE != (2, 3, 64, 64):
    state = orc.init_params(cfg, seed=3)
    x, y, dates = orc.synthetic_batch(*SIZE, seed=4)
else:
    g = load_golden("g1_diag_t3")
    state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
_, _, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32)
_, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64)


def run(tag):
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
    m.load_state_dict(state, strict=True)
    m.temporal_aggregator.attn_dropout.p = 0.0
    m = m.cuda().train()
    out = m(x.cuda(), batch_positions=dates.cuda())
    l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], y.cuda(), out[:, :, 13:26])
    l.backward()
    rows = []
    for k, p in m.named_parameters():
        if is_zero_grad(k, g64):
            continue
        t = g64[k].numpy()
        eh, ec = rel_err(p.grad.cpu().numpy(), t), rel_err(g32[k].numpy(), t)
        rows.append((eh, ec, k))
    rows.sort(reverse=True)
    over = [r for r in rows if r[0] > 5e-5]
    enc = [r for r in rows if r[2].startswith(("in_conv", "in_block"))]
    dec = [r for r in rows if r[2].startswith(("out_block", "out_conv"))]
    gm = lambda rs, i: float(np.exp(np.mean([np.log(max(r[i], 1e-12)) for r in rs]))) if rs else 0.0
    print(f"== {tag}: {len(over)} gradients further than 5e-5 from truth; geometric-mean distance from truth: encoder hip {gm(enc, 0):.2e} "
          f"cpu {gm(enc, 1):.2e} | decoder hip {gm(dec, 0):.2e} cpu {gm(dec, 1):.2e}; worst:")
    for eh, ec, k in rows[:6]:
        print(f"   hip {eh:.2e}  cpu {ec:.2e}  x{eh / max(ec, 1e-12):5.1f}  {k}")
    E._PACK_CACHE.clear()


run("default")
E._H2_FWD = E._H2_BWD = False; run("exact bf16 split everywhere"); E._H2_FWD = E._H2_BWD = True
E._DW_VARIANT = 1; run("LDS-tiled depthwise"); E._DW_VARIANT = 0
E._FUSED_DX = False; run("unfused dx"); E._FUSED_DX = True
