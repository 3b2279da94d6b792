"""One fuzz case (tools/fuzz_configs.py seed), one parameter: per-channel gradient error of the HIP path against the fp64 oracle beside the
matching bias gradient -- is the error of a norm's gamma gradient proportional to its beta gradient (a residue in the saved mean)?
    python tools/debug_gamma.py <case> <param> [--wide] [--shape=B,T,H,W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
case, pname = int(sys.argv[1]), sys.argv[2]
import torch
import fuzz_configs as F
import gpu_util

F.WIDE = "--wide" in sys.argv
got = {}
real = gpu_util.oracle_run


def spy(state, x, y, dates, cfg, dt, **kw):
    r = real(state, x, y, dates, cfg, dt, **kw)
    got[dt] = r[3]
    return r


F.oracle_run = spy
from oracle import uncrtaints_oracle as orc
real_fwd = orc.forward
taps64 = {}


def fwd(p, x, *a, **kw):          # the fp64 run also records the tensors the norms see
    if x.dtype == torch.float64 and kw.get("training") and "taps" not in kw:
        kw["taps"] = taps64
    return real_fwd(p, x, *a, **kw)


orc.forward = fwd
import uncrtaints_amd.src.backbones.uncrtaints as U
realU = U.UNCRTAINTS


def keep(**mk):
    got["m"] = realU(**mk)
    return got["m"]


U.UNCRTAINTS = keep
F.run_case(case, [a for a in sys.argv[3:] if a.startswith("--shape=") or a.startswith("--kw=")])
m = got["m"]
gh = dict(m.named_parameters())[pname].grad.double().cpu().flatten()
g64, g32 = got[torch.float64][pname].double().flatten(), got[torch.float32][pname].double().flatten()
bname = pname.replace(".weight", ".bias")
b64 = got[torch.float64][bname].double().flatten()
sc = float(g64.abs().max())
eh, ec = (gh - g64), (g32 - g64)
print(f"{pname}: max|g| {sc:.3e}  hip err {float(eh.abs().max()) / sc:.2e}  cpu err {float(ec.abs().max()) / sc:.2e};  max|{bname.split('.')[-1]} grad| {float(b64.abs().max()):.3e}")
idx = eh.abs().argsort(descending=True)[:8]
for i in idx.tolist():
    print(f"  ch {i:4d}: g64 {g64[i]: .4e}  hip-g64 {eh[i]: .3e}  cpu-g64 {ec[i]: .3e}  bias-grad {b64[i]: .4e}  err/biasgrad {eh[i] / b64[i]: .3e}")
r = eh / b64
print(f"  err / bias-grad over channels: median {float(r.median()):.3e}  mean {float(r.mean()):.3e}  std {float(r.std()):.3e};  corr(err, biasgrad) = "
      f"{float(torch.corrcoef(torch.stack([eh, b64]))[0, 1]):.3f}")

blk, which = pname.split(".conv.")[0], pname.split(".fn.")[1].split(".")[0] if ".fn." in pname else None
tap = {"1": "h1", "4": "h2", "8": "h3"}.get(which)
for tap in ([tap, "h2"] if tap == "h1" else [tap]):
  if tap and f"{blk}.{tap}" in taps64:
    h = taps64[f"{blk}.{tap}"].detach()
    ch = h.permute(1, 0, 2, 3).reshape(h.shape[1], -1)
    mu, sd = ch.mean(1), ch.std(1)
    gam = dict(m.named_parameters())[pname].detach().double().cpu()
    for i in idx.tolist()[:3]:
        print(f"    ch {i:4d}: {tap} mean {mu[i]: .4e} std {sd[i]:.4e} |mean|/std {abs(mu[i]) / sd[i]:.1f}   gamma {gam[i]: .4e}   "
              f"(median std over channels {sd.median():.3e})")
