"""Is the forward rounding error of the HIP path coherent (biased) or noise-like?  Every stored activation of the golden fixture's
train-mode forward against the fp64 oracle's taps, per setting of the forward-GEMM split:
  rel_max  max|e| / max|t|            rms   rms(e) / rms(t)
  bias     mean(e) / rms(e)           (0 for zero-mean noise; +-1 for a pure offset)
  gain     sum(e*t) / sum(t*t)        (a multiplicative error: e = gain * t)
  chan     rms over channels of the per-(frame, channel) mean(e), / rms(e): the part of the error that is constant over a plane
    python tools/forward_bias_probe.py [fixture]            (GPU box; the oracle is the checker)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import load_golden
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src.backbones import uncrtaints as U

# the positional table in fp32 for every party (the reference computes it in fp32; at dates ~1500 its fp64 evaluation differs by 2e-5
# absolute, a shift COMMON to all fp32 evaluations that would otherwise hide each path's own rounding noise)
_pt = orc.positional_table
orc.positional_table = lambda dates, d, T, repeat: _pt(dates.float(), d, T, repeat).to(dates.dtype)
name = sys.argv[1] if len(sys.argv) > 1 else "g1_diag_t3"
g = load_golden(name)
cov = json.loads(str(g["meta"]))["covmode"]
state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
x, dates = torch.from_numpy(g["x"]), torch.from_numpy(g["dates"])
cfg = orc.OracleConfig(covmode=cov, out_conv=[13 + (13 if cov == "diag" else 1)], attn_dropout=0.0)


def oracle_taps(dtype, pool_idx):
    taps = {}
    pt = {k: (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in state.items()}
    with torch.no_grad():
        out = orc.forward(pt, x.to(dtype), dates.to(dtype), cfg, training=True, taps=taps, update_running=False, pool_idx=pool_idx)
    taps["out"] = out
    return {k: v.double() for k, v in taps.items() if isinstance(v, torch.Tensor)}


def hip_taps():
    rec, names = {}, iter(["in_block.0"] + [f"out_block.{i}" for i in range(5)])
    orig = E.mbconv_forward

    def spy(xx, p, spec, training, *a, **k):
        y, sv, party = orig(xx, p, spec, training, *a, **k)
        n = next(names)
        N, C, Ch, R, H, W = sv["dims"]
        rec[n + ".h1"] = sv["h1"].detach().double().cpu().view(N, Ch, H, W)
        rec[n + ".h2"] = sv["h2"].detach().double().cpu().view(N, Ch, H, W)
        rec[n + ".h3"] = sv["h3"].detach().double().cpu().view(N, C, H, W)
        rec[n + ".y"] = y.detach().double().cpu().view(N, C, H, W)
        return y, sv, party
    E.mbconv_forward = spy
    try:
        m = U.UNCRTAINTS(input_dim=15, out_conv=[13 + (13 if cov == "diag" else 1)], out_nonlin_mean=True, out_nonlin_var="softplus",
                         covmode=cov, scale_by=1.0)
        m.load_state_dict(state, strict=True)
        m.temporal_aggregator.attn_dropout.p = 0.0
        m = m.cuda().train()
        with torch.no_grad():
            out = m(x.cuda(), batch_positions=dates.cuda())
        rec["out"] = out.double().cpu()
        rec["attn"] = m._last_attention.detach().double().cpu()
        idx = m._last_pool_idx.detach().cpu().to(torch.long)
    finally:
        E.mbconv_forward = orig
    return rec, idx


def stats(e, t):
    e, t = e.reshape(t.shape), t
    er = float(e.pow(2).mean().sqrt())
    d = {"rel_max": float(e.abs().max() / t.abs().max()), "rms": er / float(t.pow(2).mean().sqrt()),
         "bias": float(e.mean()) / max(er, 1e-300), "gain": float((e * t).sum() / (t * t).sum())}
    if e.dim() == 4:
        d["chan"] = float(e.flatten(2).mean(2).pow(2).mean().sqrt()) / max(er, 1e-300)
    return d


KEYS = [f"{b}.{k}" for b in ["in_block.0"] + [f"out_block.{i}" for i in range(5)] for k in ("h1", "h2", "h3", "y")] + ["out"]
res = {}
first, pidx = hip_taps()
t64 = oracle_taps(torch.float64, pidx)
t64.update({"in_block.0.y": t64["e"], **{f"out_block.{i}.y": t64[f"dec{i}"] for i in range(5)}})
t32 = oracle_taps(torch.float32, pidx)
t32.update({"in_block.0.y": t32["e"], **{f"out_block.{i}.y": t32[f"dec{i}"] for i in range(5)}})
KEYS = ["a0", "attn", "agg"] + KEYS + ["pre_head"]
res["cpu_fp32"] = {k: stats(t32[k] - t64[k], t64[k]) for k in KEYS}
for tag, opts in (("shipped", {}), ("exact_split_forward", {"h2_fwd": False}), ("inconv_stored", {"inconv_moments": False})):
    with E.dev_options(**opts):
        rec, _ = hip_taps()
    res[tag] = {k: stats(rec[k] - t64[k].reshape(rec[k].shape), t64[k].reshape(rec[k].shape)) for k in KEYS if k in rec}
    res[tag + "_vs_cpu32"] = {k: stats(rec[k] - t32[k].reshape(rec[k].shape), t32[k].reshape(rec[k].shape)) for k in KEYS if k in rec}
for k in KEYS:
    print(k)
    for tag in res:
        if k not in res[tag]:
            continue
        s = res[tag][k]
        print(f"    {tag:22s} rel_max {s['rel_max']:.2e} rms {s['rms']:.2e} bias {s['bias']:+.3f} gain {s['gain']:+.2e}"
              + (f" chan {s['chan']:.3f}" if "chan" in s else ""))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "forward_bias_probe.json"), "w"), indent=1)
