"""Is the excess gradient noise of the fp16 two-part forward GEMMs the (coherent) representation error of the WEIGHTS?
Runs the g1_diag_t3_pad fixture with {fp16x2, bf16x3} x {weights as given, weights pre-rounded to what two fp16 parts hold}
and prints the distance of the noisiest gradients from an fp64 oracle run on the same weights (run on the GPU box)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden, rel_err
from gpu_util import dev, oracle_run, pool_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src import losses
import test_gpu_model as TG

name = sys.argv[1] if len(sys.argv) > 1 else "g1_diag_t3_pad"
g = load_golden(name)
cov = json.loads(str(g["meta"]))["covmode"]
state0 = TG._state(load_golden("g1_diag_t3") if "state/in_conv.conv.conv.0.weight" not in g.files else g)
xc, yc, dc = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))


def two_part(w):
    w2 = w.reshape(w.shape[0], -1).double()
    amax = w2.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    S = 2.0 ** (15 - torch.ceil(torch.log2(amax) + 1e-12))
    h = (w2 * S).float().half()
    l = ((w2 * S).float() - h.float()).half()
    return ((h.double() + l.double()) / S).float().reshape(w.shape)


for quant in (False,):
    state = {k: v.clone() for k, v in state0.items()}
    if quant:
        for k in state:
            if k.endswith("conv.fn.0.weight") or k.endswith("conv.fn.7.weight"):
                state[k] = two_part(state[k])
    cfg = orc.OracleConfig(covmode=cov, out_conv=[13 + (13 if cov == "diag" else 1)], attn_dropout=0.0)
    for h2 in (True, False):
        E._H2_FWD = h2
        m = TG._build(cov, state)
        m.train()
        crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode=cov)
        out = m(dev(xc), batch_positions=dev(dc))
        l, _ = crit(out[:, :, :13], dev(yc), out[:, :, 13:m.vars_idx])
        l.backward()
        pidx, flips = pool_branch(m, state, xc, dc, cfg)
        _, _, _, g64, _ = oracle_run(state, xc, yc, dc, cfg, torch.float64, pool_idx=pidx)
        _, _, _, g32, _ = oracle_run(state, xc, yc, dc, cfg, torch.float32, pool_idx=pidx)
        rows = []
        for k, v in m.named_parameters():
            t = g64[k].numpy()
            if np.abs(t).max() == 0:
                continue
            rows.append((rel_err(v.grad.double().cpu().numpy(), t), rel_err(g32[k].double().numpy(), t), k))
        rows = [r for r in rows if r[0] < 1e-2 and r[1] < 1e-2]
        rows.sort(reverse=True)
        gm = float(np.exp(np.mean([np.log(max(r[0], 1e-12)) for r in rows if r[0] < 1e-2])))
        gmc = float(np.exp(np.mean([np.log(max(r[1], 1e-12)) for r in rows if r[0] < 1e-2])))
        print(f"== quantised weights {quant}, fp16x2 {h2}: geo-mean distance from fp64: hip {gm:.2e} cpu {gmc:.2e}; worst:")
        for r in rows[:14]:
            print(f"     hip {r[0]:.2e} cpu {r[1]:.2e} {r[2]}")
