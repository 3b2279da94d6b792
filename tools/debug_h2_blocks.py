"""Where do the fp16 two-part forward GEMMs and the exact split first part ways on a fixture?  (run on the GPU box)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden
from gpu_util import dev
from uncrtaints_amd import engine as E
import test_gpu_model as TG

name = sys.argv[1] if len(sys.argv) > 1 else "g1_diag_t3"
g = load_golden(name)
cov = json.loads(str(g["meta"]))["covmode"]
state = TG._state(g)
xc, dc = torch.from_numpy(g["x"]), torch.from_numpy(g["dates"])
rec = {}
orig = E.mbconv_forward
def spy(x, p, spec, training, *a, **k):
    y, sv, party = orig(x, p, spec, training, *a, **k)
    rec[tag[0]].append({k2: sv[k2].detach().double().cpu() for k2 in ("x", "h1", "h2", "h3")} | {"y": y.detach().double().cpu(),
                        "ub0": None if sv["n0"].ub is None else sv["n0"].ub.cpu(), "A0": sv["n0"].A.cpu(), "B0": sv["n0"].B.cpu(),
                        "ub2": None if sv["n2"].ub is None else sv["n2"].ub.cpu(), "A2": sv["n2"].A.cpu(), "B2": sv["n2"].B.cpu(), "s": sv["s"].cpu()})
    return y, sv, party
E.mbconv_forward = spy
tag = [None]
for h2 in (True, False):
    tag[0] = h2; rec[h2] = []
    E._H2_FWD = h2
    m = TG._build(cov, state).train()
    with torch.no_grad():
        m(dev(xc), batch_positions=dev(dc))
for i, (a, b) in enumerate(zip(rec[True], rec[False])):
    line = f"block {i}:"
    for k in ("x", "h1", "h2", "h3", "y"):
        d = (a[k] - b[k]).abs()
        # per (frame, channel) plane: error relative to the plane's own max
        pl = (d.flatten(2).amax(2) / b[k].abs().flatten(2).amax(2).clamp_min(1e-30))
        line += f"  {k} {float(d.max() / b[k].abs().max()):.1e} (plane-rel max {float(pl.max()):.1e})"
    print(line)
    if a["ub0"] is not None:
        N = a["x"].shape[0]
        u = a["A0"].view(N, -1, 1, 1).double() * a["x"] + a["B0"].view(N, -1, 1, 1).double()
        tm = u.abs().flatten(2).amax(2)
        print(f"   pw1 operand: frame max {[round(float(v), 2) for v in tm.amax(1)]}, bound {[round(float(v), 2) for v in a['ub0'].view(N, -1).amax(1)]}; "
              f"smallest channel max {float(tm.min()):.2e}")
