"""Attribution of the gradient rounding noise on the reference-generated fixtures.  Run on the GPU box:
    python tools/parity_attribution.py [fixture ...] > gpurun_out/parity_attribution.log
For every fixture and every ONE-switch-at-a-time setting of the rounding sources (engine.dev_options) it records
  * the forward distance from the fp64 oracle: outputs (relative to max) and the largest RELATIVE error of a predicted variance
    (= absolute error of the head's variance pre-activation where softplus ~ exp, the quantity MGNLL's 1 / var weighting amplifies),
  * e_got (distance from the fp64 gradient) of the worst gradients and of the watched ones,
  * `injected`: the fp64 oracle's gradients when ONLY its output is replaced by the HIP output (everything else exact): the share of
    e_got that is the forward noise seen through the loss, as opposed to rounding inside the backward kernels.
Writes profiles-ready JSON to gpurun_out/parity_attribution.json.  The oracle is the checker here (tools/ is test infrastructure)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import load_golden, rel_err
from gpu_util import is_zero_grad, oracle_run, pool_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

WATCH = ["out_block.1.conv.fn.1.weight", "out_block.2.conv.fn.6.fc.2.weight", "out_block.2.conv.fn.1.weight",
         "out_block.4.conv.fn.3.weight"]
SETTINGS = [
    ("shipped", {}),
    ("a_inconv_stats_from_stored_tensor", {"inconv_moments": False}),
    ("c_exact_bf16_split_forward_gemms", {"h2_fwd": False}),
    ("c_exact_bf16_split_dz", {"h2_bwd": False}),
    ("c_exact_bf16_split_dx", {"h2_dx": False}),
    ("c_exact_bf16_split_dw2", {"h2_wgrad": False}),
    ("c_exact_bf16_split_everywhere", {"h2_fwd": False, "h2_bwd": False, "h2_dx": False, "h2_wgrad": False}),
    ("bn_finalised_by_its_own_launch", {"bn_consumer": False}),
    ("unfused_dx", {"fused_dx": False}),
    ("a_plus_c_everywhere", {"inconv_moments": False, "h2_fwd": False, "h2_bwd": False, "h2_dx": False, "h2_wgrad": False}),
]
SEED_SETTINGS = ("shipped", "a_inconv_stats_from_stored_tensor", "c_exact_bf16_split_forward_gemms", "a_plus_c_everywhere")
# a variant library (tools/build_variant.sh, selected with UNCR_HIP_LIB at process start) is run with the shipped switches only and
# recorded under ATTR_TAG; results are merged into the JSON of earlier invocations
VARIANT_TAG = os.environ.get("ATTR_TAG", "")


def hip_run(state, x, y, dates, cov):
    m = U.UNCRTAINTS(input_dim=15, out_conv=[13 + (13 if cov == "diag" else 1)], out_nonlin_mean=True, out_nonlin_var="softplus",
                     covmode=cov, scale_by=1.0)
    m.load_state_dict(state, strict=True)
    m.temporal_aggregator.attn_dropout.p = 0.0
    m = m.cuda().train()
    out = m(x.cuda(), batch_positions=dates.cuda())
    l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode=cov)(out[:, :, :13], y.cuda(), out[:, :, 13:m.vars_idx])
    l.backward()
    torch.cuda.synchronize()
    return m, out.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()}


def injected_grads(state, x, y, dates, cfg, pidx, out_hip):
    """fp64 oracle, its output shifted onto the HIP output (a constant): d loss(out64 + delta) / d params."""
    pt = {k: (v.double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone().double()
              if v.dtype.is_floating_point else v.clone()) for k, v in state.items()}
    out = orc.forward(pt, x.double(), dates.double(), cfg, training=True, pool_idx=pidx)
    delta = (out_hip.double() - out.detach())
    loss = orc.loss_from_output(out + delta, y.double(), cfg)
    loss.backward()
    return {k: v.grad for k, v in pt.items() if isinstance(v, torch.Tensor) and v.requires_grad}


def errors(grads, g64):
    rows = []
    for k, g in grads.items():
        if k not in g64 or is_zero_grad(k, g64):
            continue
        rows.append((rel_err(g.double().numpy(), g64[k].numpy()), k))
    rows.sort(reverse=True)
    return rows


def fixture(name, res):
    # "g1_diag_t3@seed7": the fixture's weights on a fresh synthetic batch of the same shape (an independent draw of the rounding noise)
    base, _, seed = name.partition("@seed")
    g = load_golden(base)
    cov = json.loads(str(g["meta"]))["covmode"]
    state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
    if seed:
        x, y, dates = orc.synthetic_batch(*x.shape[:2], *x.shape[-2:], seed=int(seed))
    cfg = orc.OracleConfig(covmode=cov, out_conv=[13 + (13 if cov == "diag" else 1)], attn_dropout=0.0)
    m, _, _ = hip_run(state, x, y, dates, cov)
    pidx, flips = pool_branch(m, state, x, dates, cfg)
    out64, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx)
    out32, _, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx)

    def fwd_err(out):
        o, t = out.double(), out64
        return {"out_rel_max": float((o - t).abs().max() / t.abs().max()),
                "var_rel_elementwise_max": float(((o[:, :, 13:] - t[:, :, 13:]).abs() / t[:, :, 13:].abs()).max()),
                "var_rel_elementwise_rms": float((((o[:, :, 13:] - t[:, :, 13:]) / t[:, :, 13:]) ** 2).mean().sqrt()),
                "mean_abs_max": float((o[:, :, :13] - t[:, :, :13]).abs().max())}
    cpu_rows = errors(g32, g64)
    rec = {"pool_flips": flips, "cpu_fp32": {"forward": fwd_err(out32), "worst": [(k, e) for e, k in cpu_rows[:5]],
                                             "watch": {k: dict((kk, e) for e, kk in cpu_rows).get(k) for k in WATCH}}, "settings": {}}
    print(f"== {name}: cpu fp32 forward {rec['cpu_fp32']['forward']}", flush=True)
    runs = [(VARIANT_TAG, {})] if VARIANT_TAG else (SETTINGS if not seed else [s for s in SETTINGS if s[0] in SEED_SETTINGS])
    for tag, opts in runs:
        with E.dev_options(**opts):
            _, out, grads = hip_run(state, x, y, dates, cov)
        rows = errors(grads, g64)
        inj = errors(injected_grads(state, x, y, dates, cfg, pidx, out), g64)
        d = dict((k, e) for e, k in rows)
        di = dict((k, e) for e, k in inj)
        rec["settings"][tag] = {"options": opts, "forward": fwd_err(out), "worst": [(k, e) for e, k in rows[:6]],
                                "worst_injected_only": [(k, e) for e, k in inj[:3]],
                                "watch": {k: {"e_got": d.get(k), "injected_only": di.get(k)} for k in WATCH},
                                "n_over_1e-4": sum(1 for e, _ in rows if e > 1e-4)}
        f = rec["settings"][tag]["forward"]
        print(f"  {tag:40s} fwd out {f['out_rel_max']:.2e} var-rel max {f['var_rel_elementwise_max']:.2e} rms {f['var_rel_elementwise_rms']:.2e} | "
              f"worst {rows[0][1]} {rows[0][0]:.3e} (injected-only {di.get(rows[0][1], 0):.3e}); over 1e-4: {rec['settings'][tag]['n_over_1e-4']}",
              flush=True)
    res[name] = rec


if __name__ == "__main__":
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    names = [a for a in sys.argv[1:]] or ["g1_diag_t3"]
    res = {}
    for n in names:
        fixture(n, res)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "parity_attribution.json")
    if VARIANT_TAG and os.path.exists(path):
        old = json.load(open(path))
        for n, rec in res.items():
            old.setdefault(n, rec)["settings"].update(rec["settings"])
        res = old
    json.dump(res, open(path, "w"), indent=1)
