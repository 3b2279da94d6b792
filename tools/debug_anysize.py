"""Any H x W (csrc/anysize.hip): where does the padded-plane path part ways with the oracle?  (GPU box)
    python tools/debug_anysize.py [B T H W]
Eval forward, train forward (per MBConv tap), loss and every gradient against the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import load_golden, rel_err
from gpu_util import oracle_run, pool_branch, is_zero_grad
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

B, T, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 2, 100, 100)
g = load_golden("g1_diag_t3")
state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
cfg = orc.OracleConfig(attn_dropout=0.0)
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=3)
geom = E.plan_geom(H, W)
print("geometry", geom)


def build():
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
    m.load_state_dict(state, strict=True)
    m.temporal_aggregator.attn_dropout.p = 0.0
    return m.cuda()


def cut(t, shape4):      # [N, C, 1, Pc] -> [N, C, H, W]
    n, c = shape4[:2]
    return t.detach().reshape(n, c, -1)[:, :, :H * W].reshape(n, c, H, W).double().cpu()


for training in (False, True):
    rec, names = {}, iter(["in_block.0"] + [f"out_block.{i}" for i in range(5)])
    orig = E.mbconv_forward

    def spy(xx, p, spec, tr, *a, **k):
        yy, sv, party = orig(xx, p, spec, tr, *a, **k)
        n = next(names)
        if "h1" in sv:
            N, C, Ch, R, _, _ = sv["dims"]
            rec[n + ".h1"], rec[n + ".h2"], rec[n + ".h3"] = cut(sv["h1"], (N, Ch)), cut(sv["h2"], (N, Ch)), cut(sv["h3"], (N, C))
        rec[n + ".y"] = cut(yy, yy.shape)
        return yy, sv, party
    E.mbconv_forward = spy
    m = build()
    m.train(training)
    with torch.no_grad():
        out = m(x.cuda(), batch_positions=dates.cuda())
    E.mbconv_forward = orig
    taps = {}
    pt = {k: v.clone() for k, v in state.items()}
    pidx = m._last_pool_idx.detach().cpu().to(torch.long)
    with torch.no_grad():
        oo = orc.forward(pt, x, dates, cfg, training=training, taps=taps, update_running=False, pool_idx=pidx)
    taps.update({"in_block.0.y": taps["e"], **{f"out_block.{i}.y": taps[f"dec{i}"] for i in range(5)}})
    print(f"== training={training}: out rel_err {rel_err(out.cpu().numpy(), oo.numpy()):.2e}")
    for k in rec:
        if k in taps:
            print(f"   {k:18s} {rel_err(rec[k].numpy(), taps[k].double().numpy()):.2e}")

_orig_bwd = E.mbconv_backward
def _chk_bwd(dy, sv, p, need_dx=True, dy_part=None):
    if sv.get("geom") is not None and E.current_geom() == sv["geom"]:
        gm = sv["geom"]
        for nm, t in (("dy", dy), ("x", sv["x"]), ("h1", sv["h1"]), ("h2", sv["h2"]), ("h3", sv["h3"])):
            tl = t.reshape(-1, gm.Pc)[:, gm.P:]
            print(f"      tail of {nm}: max |.| = {float(tl.abs().max()):.3e}  (valid max {float(t.reshape(-1, gm.Pc)[:, :gm.P].abs().max()):.3e})", "part given" if (nm == "dy" and dy_part is not None) else "")
        _CUR["sv"], _CUR["dy"] = sv, dy
    return _orig_bwd(dy, sv, p, need_dx, dy_part)
E.mbconv_backward = _chk_bwd
from uncrtaints_amd import hip_backend as hb
_CUR = {}
_orig_call = hb.call
def _call(name, *a):
    if name == "uncr_fix_wgrad_tail":
        _CUR["G_pre"] = a[0].clone()
        print("      [chk] fix_wgrad args:", [tuple(t.shape) if hasattr(t, "shape") else t for t in a[:9]])
    r = _orig_call(name, *a)
    sv = _CUR.get("sv")
    if sv is None:
        return r
    gm = sv["geom"]
    N, C, Ch, R, _, _ = sv["dims"]
    val = lambda t, c: t.reshape(N, c, gm.Pc)[:, :, :gm.P].double()
    if name == "uncr_fix_wgrad_tail":
        G, c2, c3, mu, B2 = a[0], a[4], a[5], a[6], a[7]
        _CUR["k3"] = (c2, c3, mu)
    if name == "uncr_norm_finalize_bwd" and "fin" not in _CUR:
        _CUR["fin"] = 1
    if name == "uncr_se_mlp_bwd":
        G = a[0]
        n2 = sv["n2"]
        # c1 is not among the fix arguments: recompute dh3 from the finalize outputs kept by the engine is not possible here, so compare
        # the x-operand side only through the row sums: sum_ci G[n,co,ci] * 1 vs sum_p dh3 * sum_ci z  (needs dh3) -> skip; check z sums instead
        z = torch.nn.functional.gelu(n2.A.view(N, Ch, 1).double() * val(sv["h2"], Ch) + n2.B.view(N, Ch, 1).double())
        _CUR["z"] = z
        c1, c2, c3, mu = (t.view(N, C, 1).double() for t in _CUR["dk"])
        dh3 = c1 * val(_CUR["dy"], C) + c2 * (val(_CUR["d2"], C) - mu) + c3
        Gref = torch.einsum("nop,nip->noi", dh3, z)
        print(f"      [chk] G before the fix: err {float((_CUR['G_pre'].double() - Gref).abs().max() / Gref.abs().max()):.2e}")
        print(f"      [chk] G: err {float((G.double() - Gref).abs().max() / Gref.abs().max()):.2e} (max {float(Gref.abs().max()):.3e}); tail term would be {float((gm.ntail * (c3 - c2 * mu)).abs().max()):.3e} x gelu(B)")
        print(f"      [chk] pooled: hip {float(sv['pooled'].double().abs().max()):.4e} ref {float(z.mean(-1).abs().max()):.4e} err {float((sv['pooled'].double() - z.mean(-1)).abs().max() / z.mean(-1).abs().max()):.2e}")
    return r
hb.call = _call
E.hb.call = _call
_orig_wg = E.pw_wgrad
def _wg(d, x, N, Cd, Cx, P, **kw):
    if kw.get("per_frame") and kw.get("pro_x") == E.PRO_AFFINE_GELU:
        _CUR["dk"], _CUR["d2"] = kw["dk"], kw["d2"]
    return _orig_wg(d, x, N, Cd, Cx, P, **kw)
E.pw_wgrad = _wg
m = build().train()
xg = x.cuda().requires_grad_(True)
out = m(xg, batch_positions=dates.cuda())
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], y.cuda(), out[:, :, 13:26])
l.backward()
pidx, _ = pool_branch(m, state, x, dates, cfg)
_, lo, dx32, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx)
_, _, dx64, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx)
print(f"loss hip {l.item():.6f} oracle {lo.item():.6f}")
print(f"dx: vs fp32 {rel_err(xg.grad.cpu().numpy(), dx32.numpy()):.2e} vs fp64 {rel_err(xg.grad.cpu().numpy(), dx64.numpy()):.2e}")
rows = []
for k, v in m.named_parameters():
    if is_zero_grad(k, g64):
        continue
    rows.append((rel_err(v.grad.cpu().numpy(), g64[k].numpy()), rel_err(g32[k].numpy(), g64[k].numpy()), k))
for e, ec, k in sorted(rows, reverse=True)[:25]:
    print(f"   {e:.2e} (cpu {ec:.2e}) {k}")
print("-- out_block.4 / out_conv / in_block.0 / in_conv, by name:")
for e, ec, k in sorted(rows, key=lambda r: r[2]):
    if k.startswith(("out_block.4", "out_conv", "in_block.0", "in_conv", "temporal")):
        print(f"   {e:.2e} (cpu {ec:.2e}) {k}")
print("gradients further than 1e-4 from fp64:", sum(1 for e, _, _ in rows if e > 1e-4), "of", len(rows))
