#!/usr/bin/env python
"""Generate golden fixtures by importing the UnCRtainTS reference (read-only) from /root/reference.

Run ONLY in the build container (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Writes `*.npz` next to this file.  The fixtures hold data only (inputs, weights, expected outputs,
gradients, checksums); no reference source travels.  SURVEY.md section 8(c) lists the cases (G1..G7).
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(REF, "model"), REF]

# `src.model_utils` / `base_model` import fvcore only for --profile; stub it.
_fv = types.ModuleType("fvcore"); _fvnn = types.ModuleType("fvcore.nn")
_fvnn.FlopCountAnalysis = object; _fvnn.flop_count_table = lambda *a, **k: ""
_fv.nn = _fvnn
sys.modules.setdefault("fvcore", _fv); sys.modules.setdefault("fvcore.nn", _fvnn)

os.chdir("/tmp")  # model_utils.py:4-5 chdirs into ./model if it exists
from src.backbones import uncrtaints  # noqa: E402
from src.backbones.positional_encoding import PositionalEncoder  # noqa: E402
from src import losses  # noqa: E402
from src.learning.weight_init import weight_init  # noqa: E402

torch.set_num_threads(8)


def np_state(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def proj_vec(n, seed=1234):
    return np.random.default_rng(seed).standard_normal(n).astype(np.float64)


def checksum(a):
    """(sum, abs-sum, random projection) in float64 -- compact stand-in for a full tensor."""
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), float(a @ proj_vec(a.size))])


def build(covmode, seed):
    torch.manual_seed(seed)
    out_c = 13 + (13 if covmode == "diag" else 1)
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[out_c], out_nonlin_mean=True, out_nonlin_var="softplus",
                              covmode=covmode, scale_by=1.0)
    m.apply(weight_init)
    # non-trivial BatchNorm running statistics so eval mode is a real test
    g = torch.Generator().manual_seed(seed + 100)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
            mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
    # non-trivial GroupNorm affine (weight_init leaves them at 1/0)
    for mod in m.modules():
        if isinstance(mod, torch.nn.GroupNorm):
            mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
            mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    m.temporal_aggregator.attn_dropout.p = 0.0     # deterministic train mode
    return m


def synth(B, T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, T, 15, H, W, generator=g)
    y = torch.rand(B, 1, 13, H, W, generator=g)
    dates = torch.sort(torch.randint(1400, 1800, (B, T), generator=g), dim=1).values.float()
    return x, y, dates


def crit(covmode):
    return losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode=covmode, chunk=None)


def case_model(name, covmode, B, T, H, W, seed, full_grads, pad_last=False, save_state=True, taps=True):
    m = build(covmode, seed)
    x, y, dates = synth(B, T, H, W, seed)
    if pad_last:
        x[0, T - 1] = 0.0
    state0 = np_state(m.state_dict())
    out = {"x": x.numpy(), "y": y.numpy(), "dates": dates.numpy(),
           "meta": json.dumps(dict(covmode=covmode, B=B, T=T, H=H, W=W, seed=seed, pad_last=pad_last))}
    if save_state:
        for k, v in state0.items():
            out["state/" + k] = v

    # ---- eval-mode forward (running stats) ----
    m.eval()
    feats = {}
    hooks = []
    if taps:
        hooks.append(m.temporal_encoder.register_forward_hook(lambda mod, i, o: feats.__setitem__("attn", o.detach())))
        hooks.append(m.temporal_aggregator.register_forward_hook(lambda mod, i, o: feats.__setitem__("agg", o.detach())))
        # smart_forward (utae.py:422-450) calls .forward directly, so tap by wrapping the bound method
        def tap(mod, key):
            orig = mod.forward
            def wrapped(inp):
                o = orig(inp); feats[key] = o.detach(); return o
            mod.forward = wrapped
            class _H:
                def remove(self_inner): del mod.forward
            return _H()
        hooks.append(tap(m.in_block[0], "e"))
        hooks.append(tap(m.in_conv, "a0"))
    with torch.no_grad():
        o_eval = m(x, batch_positions=dates)
        vi = m.vars_idx
        l_eval, _ = crit(covmode)(o_eval[:, :, :13], y, o_eval[:, :, 13:vi])
    for h in hooks:
        h.remove()
    out["eval/out"] = o_eval.numpy()
    out["eval/loss"] = np.array(l_eval.item())
    if taps:
        out["eval/attn"] = feats["attn"].contiguous().numpy()          # [16,B,T,32,32]
        e = feats["e"].reshape(B, T, 128, H, W)
        out["eval/e_b0t0"] = e[0, 0, ::16].contiguous().numpy()        # 8 channels of one frame
        out["eval/e_checksum"] = checksum(e.numpy())
        out["eval/agg_b0"] = feats["agg"][0, ::16].contiguous().numpy()
        out["eval/agg_checksum"] = checksum(feats["agg"].numpy())
        out["eval/a0_checksum"] = checksum(feats["a0"].numpy())

    # ---- train-mode forward + MGNLL + backward (dropout p=0, batch-stat BN) ----
    m.train()
    xg = x.clone().requires_grad_(True)
    o_tr = m(xg, batch_positions=dates)
    l_tr, _ = crit(covmode)(o_tr[:, :, :13], y, o_tr[:, :, 13:m.vars_idx])
    l_tr.backward()
    out["train/out"] = o_tr.detach().numpy()
    out["train/loss"] = np.array(l_tr.item())
    out["train/dx_checksum"] = checksum(xg.grad.numpy())
    out["train/dx_b0t0"] = xg.grad[0, 0].numpy()
    for k, v in m.named_parameters():
        g = v.grad.numpy()
        if full_grads:
            out["grad/" + k] = g
        out["gradsum/" + k] = checksum(g)
    state1 = np_state(m.state_dict())
    for k, v in state1.items():
        if "running_" in k:
            out["train/state/" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "eval loss", l_eval.item(), "train loss", l_tr.item())


def case_mgnll():
    out = {}
    g = torch.Generator().manual_seed(7)
    idx = 0
    for mode in ("diag", "iso"):
        for B in (1, 2, 4):
            H, W = 5, 7
            pred = torch.rand(B, 1, 13, H, W, generator=g, requires_grad=True)
            targ = torch.rand(B, 1, 13, H, W, generator=g)
            vc = 13 if mode == "diag" else 1
            var0 = torch.rand(B, 1, vc, H, W, generator=g) * 0.5 + 1e-3
            var0[0, 0, 0, 0, 0] = 1e-10          # below the clamp (eps=1e-8)
            var0[-1, 0, -1, 2, 3] = 0.0           # exactly zero
            var = var0.clone().requires_grad_(True)
            for red in ("none", "mean", "sum"):
                c = losses.MultiGaussianNLLLoss(reduction=red, eps=1e-8, full=True, mode=mode, chunk=None)
                l, v = c(pred, targ, var)
                out[f"k{idx}/loss_{red}"] = l.detach().numpy()
                if red == "mean":
                    out[f"k{idx}/variance"] = v.detach().numpy()
                    gp, gv = torch.autograd.grad(l, (pred, var))
                    out[f"k{idx}/dpred"], out[f"k{idx}/dvar"] = gp.numpy(), gv.numpy()
            out[f"k{idx}/pred"], out[f"k{idx}/target"], out[f"k{idx}/var"] = pred.detach().numpy(), targ.numpy(), var0.numpy()
            out[f"k{idx}/meta"] = json.dumps(dict(mode=mode, B=B))
            idx += 1
    out["n"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "g3_mgnll.npz"), **out)
    print("g3_mgnll", idx, "cases")


def case_eltlosses():
    """GaussianNLLLoss (losses.py:46-128, get_loss 'GNLL') and the l1 / l2 criteria (get_loss 'l1'/'l2')."""
    out = {}
    g = torch.Generator().manual_seed(11)
    idx = 0
    for B in (1, 3):
        H, W = 6, 5
        pred = torch.rand(B, 1, 13, H, W, generator=g, requires_grad=True)
        targ = torch.rand(B, 1, 13, H, W, generator=g)
        var0 = torch.rand(B, 1, 13, H, W, generator=g) * 0.5 + 1e-3
        var0[0, 0, 0, 0, 0] = 1e-10          # below the clamp (eps=1e-8)
        var0[-1, 0, -1, 2, 3] = 0.0           # exactly zero
        var = var0.clone().requires_grad_(True)
        for red in ("none", "mean", "sum"):
            c = losses.GaussianNLLLoss(reduction=red, eps=1e-8, full=True)
            l, v = c(pred, targ, var)
            out[f"k{idx}/gnll_{red}"] = l.detach().numpy()
            if red == "mean":
                out[f"k{idx}/gnll_variance"] = v.detach().numpy()
                gp, gv = torch.autograd.grad(l, (pred, var))
                out[f"k{idx}/gnll_dpred"], out[f"k{idx}/gnll_dvar"] = gp.numpy(), gv.numpy()
        for name, crit in (("l1", torch.nn.L1Loss()), ("l2", torch.nn.MSELoss())):
            l = crit(pred, targ)
            out[f"k{idx}/{name}"] = l.detach().numpy()
            out[f"k{idx}/{name}_dpred"] = torch.autograd.grad(l, pred)[0].numpy()
        out[f"k{idx}/pred"], out[f"k{idx}/target"], out[f"k{idx}/var"] = pred.detach().numpy(), targ.numpy(), var0.numpy()
        idx += 1
    out["n"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "g9_eltlosses.npz"), **out)
    print("g9_eltlosses", idx, "cases")


def _reference_functions(path, names, namespace):
    """Execute ONLY the named top-level functions of a reference file that cannot be imported as a module
    (train_reconstruct.py parses argv at import, data/dataLoader.py needs rasterio): the function definitions are
    taken from the parsed AST at generation time; nothing of the source is stored in the fixture."""
    import ast
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), [n.name for n in body]
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), namespace)
    return namespace


def case_prepare():
    """Input assembly in front of the path: process_MS / process_SAR (data/dataLoader.py:31-59) and
    prepare_data_multi (model/train_reconstruct.py:161-179)."""
    import types
    ns_d = _reference_functions(os.path.join(REF, "data", "dataLoader.py"), ["rescale", "process_MS", "process_SAR"],
                                {"np": np})
    ns_t = _reference_functions(os.path.join(REF, "model", "train_reconstruct.py"),
                                ["prepare_data_multi", "recursive_todevice"], {"torch": torch, "np": np})
    out = {}
    rng = np.random.default_rng(5)
    ms = (rng.random((13, 9, 8)).astype(np.float32) * 14000 - 1500)
    ms[0, 0, 0], ms[1, 1, 1], ms[2, 2, 2] = np.nan, np.inf, -np.inf
    sar = (rng.random((2, 9, 8)).astype(np.float32) * 40 - 36)
    sar[0, 0, 1], sar[1, 3, 3] = np.nan, np.inf
    out["ms_raw"], out["sar_raw"] = ms, sar
    for method in ("default", "resnet"):
        out[f"ms_{method}"] = ns_d["process_MS"](ms.copy(), method)
        out[f"sar_{method}"] = ns_d["process_SAR"](sar.copy(), method)
    # prepare_data_multi: batch as the loader collates it (lists over dates of [B, C, H, W] tensors)
    g = torch.Generator().manual_seed(3)
    B, T, H, W = 2, 3, 6, 5
    batch = {"input": {"S1": [torch.rand(B, 2, H, W, generator=g) for _ in range(T)],
                       "S2": [torch.rand(B, 13, H, W, generator=g) for _ in range(T)],
                       "masks": [torch.rand(B, H, W, generator=g) for _ in range(T)],
                       "S1 TD": [torch.randint(1400, 1800, (B,), generator=g) for _ in range(T)],
                       "S2 TD": [torch.randint(1400, 1800, (B,), generator=g) for _ in range(T)]},
             "target": {"S2": [torch.rand(B, 13, H, W, generator=g)]}}
    for k in ("S1", "S2", "masks", "S1 TD", "S2 TD"):
        for t in range(T):
            out[f"batch/{k.replace(' ', '_')}/{t}"] = batch["input"][k][t].numpy()
    out["batch/target"] = batch["target"]["S2"][0].numpy()
    for use_sar in (True, False):
        cfg = types.SimpleNamespace(batch_size=B, use_sar=use_sar)
        x, y, m, dates = ns_t["prepare_data_multi"](batch, torch.device("cpu"), cfg)
        tag = "sar" if use_sar else "nosar"
        out[f"{tag}/x"], out[f"{tag}/y"], out[f"{tag}/m"], out[f"{tag}/dates"] = x.numpy(), y.numpy(), m.numpy(), dates.numpy()
    np.savez_compressed(os.path.join(HERE, "g10_prepare.npz"), **out)
    print("g10_prepare", {k: v.shape for k, v in out.items() if "/x" in k or "dates" in k})


def case_metrics():
    """img_metrics (model/src/learning/metrics.py:20-63) incl. SSIM (util/pytorch_ssim) and the nan-aware statistics."""
    from src.learning import metrics as ref_metrics
    out = {}
    g = torch.Generator().manual_seed(13)
    idx = 0
    for (B, H, W, nan) in ((1, 24, 20, False), (3, 37, 33, True)):
        targ = torch.rand(B, 13, H, W, generator=g)
        pred = (targ + 0.1 * torch.randn(B, 13, H, W, generator=g)).clamp(0, 1)
        var = torch.rand(B, 13, H, W, generator=g) * 0.1
        if nan:
            var[0, 2, 3, 4] = float("nan")
            var[:, 5, 7, 7] = float("nan")            # NaN for every batch item of one (channel, pixel)
        d = ref_metrics.img_metrics(targ, pred, var)
        for k, v in d.items():
            out[f"k{idx}/m/{k}"] = np.asarray(v)
        out[f"k{idx}/ssim_items"] = ref_metrics.pytorch_ssim.ssim(targ, pred, size_average=False).numpy()
        out[f"k{idx}/target"], out[f"k{idx}/pred"], out[f"k{idx}/var"] = targ.numpy(), pred.numpy(), var.numpy()
        idx += 1
    out["n"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "g11_metrics.npz"), **out)
    print("g11_metrics", idx, "cases", {k: float(v) for k, v in d.items() if np.ndim(v) == 0})


def case_calibration():
    """G14: compute_ece / compute_uce_auce (model/train_reconstruct.py:475-530): calibration of the predicted variance from
    per-sample (variance, error) lists.  The plotting side of compute_uce_auce goes to stubs; the lambda `binarize`
    (train_reconstruct.py:490) is re-stated here because it is an assignment, not a function definition."""
    import types

    class _Anything:                       # absorbs plt.* / writer.* / fig.* / ax.* calls
        def __getattr__(self, k): return self
        def __call__(self, *a, **k): return self
        def __iter__(self): return iter((self, self))
    binarize = lambda arg, n_bins, floor=0, ceil=1: np.digitize(arg, bins=np.linspace(floor, ceil, num=n_bins)[1:])
    ns = _reference_functions(os.path.join(REF, "model", "train_reconstruct.py"), ["compute_ece", "compute_uce_auce"],
                              {"torch": torch, "np": np, "plt": _Anything(), "writer": _Anything(), "binarize": binarize})
    out = {}
    rng = np.random.default_rng(14)
    idx = 0
    for n, nan in ((40, False), (137, True), (1000, False)):
        var = (rng.gamma(2.0, 0.01, n)).astype(np.float64)
        err = np.abs(rng.standard_normal(n)) * np.sqrt(var) * rng.uniform(0.5, 1.5)
        if nan:
            err[[3, 50]] = np.nan
        for l2 in (True, False):
            uce, auce = ns["compute_uce_auce"](list(var), list(err), n, percent=5, l2=l2, mode="test", step=0)
            out[f"k{idx}/uce_{'l2' if l2 else 'l1'}"] = np.array([uce.item(), auce.item()])
        out[f"k{idx}/ece"] = ns["compute_ece"](list(var), list(err ** 2), n, percent=5)
        out[f"k{idx}/var"], out[f"k{idx}/err"] = var, err
        idx += 1
    out["n"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "g14_calibration.npz"), **out)
    print("g14_calibration", {k: v for k, v in out.items() if "uce" in k})


def case_smallmap():
    """G15: a 32 x 32 input -- not larger than the 32 x 32 attention map, so the aggregator takes its AvgPool2d branch
    (identity here) which has NO dropout even in train mode (uncrtaints.py:197-204); weights from g1_diag_t3."""
    base = np.load(os.path.join(HERE, "g1_diag_t3.npz"))
    state = {k[len("state/"):]: torch.from_numpy(base[k]) for k in base.files if k.startswith("state/")}
    torch.manual_seed(0)
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus",
                              covmode="diag", scale_by=1.0)
    m.load_state_dict(state, strict=True)
    assert m.temporal_aggregator.attn_dropout.p > 0
    x, y, dates = synth(1, 3, 32, 32, 15)
    m.train()
    with torch.no_grad():
        o1 = m(x, batch_positions=dates)
        m.load_state_dict(state, strict=True)
        o2 = m(x, batch_positions=dates)
    assert torch.equal(o1, o2), "the reference applied dropout on the small-map branch"
    m.load_state_dict(state, strict=True)
    m.eval()
    with torch.no_grad():
        oe = m(x, batch_positions=dates)
    np.savez_compressed(os.path.join(HERE, "g15_smallmap.npz"), x=x.numpy(), dates=dates.numpy(), train_out=o1.numpy(),
                        eval_out=oe.numpy())
    print("g15_smallmap", tuple(o1.shape))


def case_aggpool():
    """G18: Compact_Temporal_Aggregator 'att_group' on feature maps SMALLER than the 32 x 32 attention map -- the AvgPool2d
    branch (uncrtaints.py:197-204: kernel = w // H, no dropout even in train mode), with and without a padded date."""
    out = {}
    gen = torch.Generator().manual_seed(18)
    for i, (H, padded, nh) in enumerate(((16, False, 4), (8, True, 4), (16, True, 16))):
        B, T, C = 2, 3, 32
        x = torch.randn(B, T, C, H, H, generator=gen, requires_grad=True)
        att = torch.softmax(torch.randn(nh, B, T, 32, 32, generator=gen), dim=2).requires_grad_(True)
        pad = torch.zeros(B, T, dtype=torch.bool)
        if padded:
            pad[1, 0] = True
        gy = torch.randn(B, C, H, H, generator=gen)
        agg = uncrtaints.Compact_Temporal_Aggregator(mode="att_group").train()
        o1 = agg(x, pad_mask=pad, attn_mask=att)
        o2 = agg(x, pad_mask=pad, attn_mask=att)
        assert torch.equal(o1, o2), "the reference applied dropout on the AvgPool branch"
        o1.backward(gy)
        out.update({f"k{i}/x": x.detach().numpy(), f"k{i}/att": att.detach().numpy(), f"k{i}/pad": pad.numpy(),
                    f"k{i}/gy": gy.numpy(), f"k{i}/out": o1.detach().numpy(), f"k{i}/dx": x.grad.numpy(),
                    f"k{i}/datt": att.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "g18_aggpool.npz"), n=3, **out)
    print("g18_aggpool", 3)


def case_attention_rows():
    """G19: the attention classes of model/src/backbones/ltae.py called ON THEIR OWN on pixel-major rows (their native layout):
    ScaledDotProductAttentionSmall (:418-458), ScaledDotProductAttention (:388-416, eval: no dropout), MultiHeadAttentionSmall
    (:313-385), MultiHeadAttention (:244-307) and LTAE2d (:10-141, eval and train) -- forward outputs and every gradient, each case
    with a padded date.  The stand-alone entries of the HIP path (csrc/attn_rows.hip) are held to these."""
    from src.backbones import ltae as R
    out = {}
    gen = torch.Generator().manual_seed(19)
    rn = lambda *s: torch.randn(*s, generator=gen)
    # -- scaled dot-product attention on rows: q [m, dk], k [m, T, dk], v [m, T, dv], pad [m, T]
    for i, (cls, T, dk, dv) in enumerate((("ScaledDotProductAttentionSmall", 3, 4, 16), ("ScaledDotProductAttentionSmall", 6, 8, 8),
                                          ("ScaledDotProductAttention", 4, 4, 16))):
        m = 333
        q, k, v = (t.requires_grad_(True) for t in (rn(m, dk), rn(m, T, dk), rn(m, T, dv)))
        pad = torch.rand(m, T, generator=gen) < 0.3
        pad[:, 0] = False
        ga, go, gc = rn(m, 1, T), rn(m, 1, dv), rn(m, 1, T)
        temp = float(np.power(dk, 0.5))
        if cls.endswith("Small"):
            mod = R.ScaledDotProductAttentionSmall(temperature=temp)
            o, a, c = mod(q, k, v, pad_mask=pad, return_comp=True, weight_v=True)
        else:
            mod = R.ScaledDotProductAttention(temperature=temp, attn_dropout=0.1).eval()
            o, a, c = mod(q, k, v, pad_mask=pad, return_comp=True)
        ((a * ga).sum() + (o * go).sum() + (c * gc).sum()).backward()
        out.update({f"sdpa{i}/cls": np.array(cls), f"sdpa{i}/temperature": np.float64(temp), f"sdpa{i}/q": q.detach().numpy(),
                    f"sdpa{i}/k": k.detach().numpy(), f"sdpa{i}/v": v.detach().numpy(), f"sdpa{i}/pad": pad.numpy(),
                    f"sdpa{i}/ga": ga.numpy(), f"sdpa{i}/go": go.numpy(), f"sdpa{i}/gc": gc.numpy(),
                    f"sdpa{i}/out": o.detach().numpy(), f"sdpa{i}/attn": a.detach().numpy(), f"sdpa{i}/comp": c.detach().numpy(),
                    f"sdpa{i}/dq": q.grad.numpy(), f"sdpa{i}/dk": k.grad.numpy(), f"sdpa{i}/dv": v.grad.numpy()})
    # -- multi-head attention on rows: v [n, T, d_in]
    for i, (cls, weight_v) in enumerate((("MultiHeadAttentionSmall", False), ("MultiHeadAttentionSmall", True), ("MultiHeadAttention", True))):
        nh, dk, d_in, n, T = 16, 4, 256, 70, 3
        torch.manual_seed(190 + i)
        mod = R.MultiHeadAttentionSmall(n_head=nh, d_k=dk, d_in=d_in) if cls.endswith("Small") \
            else R.MultiHeadAttention(n_head=nh, d_k=dk, d_in=d_in, use_dropout=False)
        with torch.no_grad():
            mod.fc1_k.bias.copy_(0.3 * rn(nh * dk))
        mod.eval()
        v = rn(n, T, d_in).requires_grad_(True)
        pad = torch.zeros(n, T, dtype=torch.bool)
        pad[::3, T - 1] = True
        ga, go = rn(nh, n, T), rn(nh, n, d_in // nh)
        if cls.endswith("Small"):
            res = mod(v, pad_mask=pad, weight_v=weight_v)
            o, a = res if weight_v else (None, res)
        else:
            o, a = mod(v, pad_mask=pad)
        ((a * ga).sum() + ((o * go).sum() if o is not None else 0.0)).backward()
        out.update({f"mha{i}/cls": np.array(cls), f"mha{i}/weight_v": np.array(weight_v), f"mha{i}/v": v.detach().numpy(),
                    f"mha{i}/pad": pad.numpy(), f"mha{i}/ga": ga.numpy(), f"mha{i}/go": go.numpy(), f"mha{i}/attn": a.detach().numpy(),
                    f"mha{i}/dv": v.grad.numpy(), f"mha{i}/W": mod.fc1_k.weight.detach().numpy(), f"mha{i}/b": mod.fc1_k.bias.detach().numpy(),
                    f"mha{i}/Q": mod.Q.detach().numpy(), f"mha{i}/dW": mod.fc1_k.weight.grad.numpy(), f"mha{i}/db": mod.fc1_k.bias.grad.numpy(),
                    f"mha{i}/dQ": mod.Q.grad.numpy()})
        if o is not None:
            out[f"mha{i}/out"] = o.detach().numpy()
    # -- LTAE2d on [B, T, C, h, w] (values + attention), eval and train (dropout 0: the reference's streams do not travel).  ONE set of
    #    weights and inputs, run in both modes: the fixture stores them once (the HIP path takes the L-TAE's own 32 x 32 map)
    C, nh, dk, B, T, hw = 128, 16, 4, 1, 3, 32
    torch.manual_seed(195)
    mod = R.LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=[256, C], dropout=0.0, d_model=256, return_att=True, use_dropout=False)
    with torch.no_grad():
        mod.in_norm.weight.copy_(1.0 + 0.3 * rn(C)); mod.in_norm.bias.copy_(0.2 * rn(C))
        mod.out_norm.weight.copy_(1.0 + 0.3 * rn(C)); mod.out_norm.bias.copy_(0.2 * rn(C))
        # (BatchNorm bias well above zero: the ReLU then leaves most of a row's 8-value GroupNorm group alive -- with dead groups
        # the out_norm variance sits at eps and two correct fp32 evaluations of this module differ by 3e-4)
        mod.mlp[1].weight.copy_(1.0 + 0.3 * rn(C)); mod.mlp[1].bias.copy_(1.5 + 0.2 * rn(C))
        mod.mlp[1].running_mean.copy_(0.1 * rn(C)); mod.mlp[1].running_var.copy_(0.5 + torch.rand(C, generator=gen))
        mod.attention_heads.fc1_k.bias.copy_(0.3 * rn(nh * dk))
    state0 = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    x0 = rn(B, T, C, hw, hw)
    dates = torch.sort(torch.randint(1400, 1800, (B, T), generator=gen), dim=1).values.float()
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[B - 1, T - 1] = True
    gv, ga = rn(B, C, hw, hw), rn(nh, B, T, hw, hw)
    out.update({"ltae/x": x0.numpy(), "ltae/dates": dates.numpy(), "ltae/pad": pad.numpy(), "ltae/gv": gv.numpy(), "ltae/ga": ga.numpy()})
    out.update({f"ltae/state/{k}": v.numpy() for k, v in state0.items()})
    for i, training in enumerate((False, True)):
        mod.load_state_dict(state0)
        mod.zero_grad()
        mod.train(training)
        x = x0.clone().requires_grad_(True)
        o, a = mod(x, batch_positions=dates, pad_mask=pad)
        ((o * gv).sum() + (a * ga).sum()).backward()
        out.update({f"ltae{i}/training": np.array(training), f"ltae{i}/out": o.detach().numpy(), f"ltae{i}/attn": a.detach().numpy(),
                    f"ltae{i}/dx": x.grad.numpy()})
        out.update({f"ltae{i}/grad/{k}": p.grad.numpy().copy() for k, p in mod.named_parameters() if p.grad is not None})
        out.update({f"ltae{i}/after/{k}": v.detach().numpy().copy() for k, v in mod.state_dict().items() if "running" in k})
    np.savez_compressed(os.path.join(HERE, "g19_attention_rows.npz"), **out)
    print("g19_attention_rows", len(out), "arrays")


def case_ltae2d_deep():
    """G21: the reference LTAE2d (model/src/backbones/ltae.py:10-141) with a TWO-layer value MLP (mlp=[256, 128, 64], ltae.py:75-84),
    called on its own in eval and train mode with a padded date: outputs, attention, input gradient, every parameter gradient and the
    running statistics of both BatchNorm1d layers."""
    from src.backbones import ltae as R
    out = {}
    gen = torch.Generator().manual_seed(21)
    rn = lambda *s: torch.randn(*s, generator=gen)
    C, nh, dk, B, T, hw, mlp = 128, 16, 4, 1, 3, 32, [256, 128, 64]
    torch.manual_seed(210)
    mod = R.LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=mlp, dropout=0.0, d_model=256, return_att=True, use_dropout=False)
    with torch.no_grad():
        mod.in_norm.weight.copy_(1.0 + 0.3 * rn(C)); mod.in_norm.bias.copy_(0.2 * rn(C))
        mod.out_norm.weight.copy_(1.0 + 0.3 * rn(mlp[-1])); mod.out_norm.bias.copy_(0.2 * rn(mlp[-1]))
        for i, width in ((1, mlp[1]), (4, mlp[2])):        # BatchNorm biases well above zero: see case_attention_rows
            mod.mlp[i].weight.copy_(1.0 + 0.3 * rn(width)); mod.mlp[i].bias.copy_(1.5 + 0.2 * rn(width))
            mod.mlp[i].running_mean.copy_(0.1 * rn(width)); mod.mlp[i].running_var.copy_(0.5 + torch.rand(width, generator=gen))
        mod.attention_heads.fc1_k.bias.copy_(0.3 * rn(nh * dk))
    state0 = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    x0 = rn(B, T, C, hw, hw)
    dates = torch.sort(torch.randint(1400, 1800, (B, T), generator=gen), dim=1).values.float()
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[B - 1, T - 1] = True
    gv, ga = rn(B, mlp[-1], hw, hw), rn(nh, B, T, hw, hw)
    out.update({"mlp": np.array(mlp), "x": x0.numpy(), "dates": dates.numpy(), "pad": pad.numpy(), "gv": gv.numpy(), "ga": ga.numpy()})
    out.update({f"state/{k}": v.numpy() for k, v in state0.items()})
    for i, training in enumerate((False, True)):
        mod.load_state_dict(state0)
        mod.zero_grad()
        mod.train(training)
        x = x0.clone().requires_grad_(True)
        o, a = mod(x, batch_positions=dates, pad_mask=pad)
        ((o * gv).sum() + (a * ga).sum()).backward()
        out.update({f"run{i}/training": np.array(training), f"run{i}/out": o.detach().numpy(), f"run{i}/attn": a.detach().numpy(),
                    f"run{i}/dx": x.grad.numpy()})
        out.update({f"run{i}/grad/{k}": p.grad.numpy().copy() for k, p in mod.named_parameters() if p.grad is not None})
        out.update({f"run{i}/after/{k}": v.detach().numpy().copy() for k, v in mod.state_dict().items() if "running" in k})
    np.savez_compressed(os.path.join(HERE, "g21_ltae2d_deep.npz"), **out)
    print("g21_ltae2d_deep", len(out), "arrays")


def case_ltae2d_nomodel():
    """G22: the reference LTAE2d without its input projection (d_model=None, ltae.py:49-54: the attention and the values work on the
    in_channels themselves), eval and train, a padded date -- small: 64 channels, 8 heads."""
    from src.backbones import ltae as R
    out = {}
    gen = torch.Generator().manual_seed(22)
    rn = lambda *s: torch.randn(*s, generator=gen)
    C, nh, dk, B, T, hw, mlp = 64, 8, 4, 1, 2, 32, [64, 32]
    torch.manual_seed(220)
    mod = R.LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=mlp, dropout=0.0, d_model=None, return_att=True, use_dropout=False)
    with torch.no_grad():
        mod.in_norm.weight.copy_(1.0 + 0.3 * rn(C)); mod.in_norm.bias.copy_(0.2 * rn(C))
        mod.out_norm.weight.copy_(1.0 + 0.3 * rn(mlp[-1])); mod.out_norm.bias.copy_(0.2 * rn(mlp[-1]))
        mod.mlp[1].weight.copy_(1.0 + 0.3 * rn(mlp[1])); mod.mlp[1].bias.copy_(1.5 + 0.2 * rn(mlp[1]))
        mod.mlp[1].running_mean.copy_(0.1 * rn(mlp[1])); mod.mlp[1].running_var.copy_(0.5 + torch.rand(mlp[1], generator=gen))
        mod.attention_heads.fc1_k.bias.copy_(0.3 * rn(nh * dk))
    state0 = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    x0 = rn(B, T, C, hw, hw)
    dates = torch.sort(torch.randint(1400, 1800, (B, T), generator=gen), dim=1).values.float()
    pad = torch.zeros(B, T, dtype=torch.bool)
    gv, ga = rn(B, mlp[-1], hw, hw), rn(nh, B, T, hw, hw)
    out.update({"mlp": np.array(mlp), "x": x0.numpy(), "dates": dates.numpy(), "pad": pad.numpy(), "gv": gv.numpy(), "ga": ga.numpy()})
    out.update({f"state/{k}": v.numpy() for k, v in state0.items()})
    for i, training in enumerate((False, True)):
        mod.load_state_dict(state0)
        mod.zero_grad()
        mod.train(training)
        x = x0.clone().requires_grad_(True)
        o, a = mod(x, batch_positions=dates, pad_mask=pad)
        ((o * gv).sum() + (a * ga).sum()).backward()
        out.update({f"run{i}/training": np.array(training), f"run{i}/out": o.detach().numpy(), f"run{i}/attn": a.detach().numpy(),
                    f"run{i}/dx": x.grad.numpy()})
        out.update({f"run{i}/grad/{k}": p.grad.numpy().copy() for k, p in mod.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, "g22_ltae2d_nomodel.npz"), **out)
    print("g22_ltae2d_nomodel", len(out), "arrays")


def case_posenc():
    pe = PositionalEncoder(256 // 16, T=1000, repeat=16)
    dates = torch.tensor([[1400., 1433., 1799.], [0., 1., 1000.]])
    tab = pe(dates)
    np.savez_compressed(os.path.join(HERE, "g7_posenc.npz"), dates=dates.numpy(), table=tab.numpy())
    print("g7_posenc", tuple(tab.shape))


def case_ensemble():
    # ensemble_reconstruct.py:116-133 has module-level side effects and cannot be imported; known-answer
    # vectors are produced from the published mixture-moment formula on random inputs.
    rng = np.random.default_rng(5)
    mu = rng.random((5, 13, 6, 6)).astype(np.float32)
    var = (rng.random((5, 13, 6, 6)) * 0.1 + 1e-3).astype(np.float32)
    mu64, var64 = mu.astype(np.float64), var.astype(np.float64)
    m = mu64.mean(0)
    np.savez_compressed(os.path.join(HERE, "g8_ensemble.npz"), mu=mu, var=var, mean_ens=m,
                        var_both=(var64 + mu64 ** 2).mean(0) - m ** 2, var_alea=var64.mean(0),
                        var_epi=(mu64 ** 2).mean(0) - m ** 2)
    print("g8_ensemble")


def case_trainseq():
    """G6: BaseModel.optimize_parameters x3 (base_model.py:115-131) with parser defaults + the
    train_reconstruct.py:53-61 fix-ups for covmode=diag."""
    from parse_args import create_parser
    from src.model_utils import get_model
    cfg = create_parser(mode="train").parse_args([])
    cfg.model, cfg.use_sar, cfg.device, cfg.lr, cfg.input_t = "uncrtaints", True, "cpu", 1e-3, 3
    from src.utils import str2list
    cfg = str2list(cfg, ["encoder_widths", "decoder_widths", "out_conv"])
    cfg.out_conv[-1] += 13                      # train_reconstruct.py:53-57 (diag)
    cfg.var_nonLinearity = "softplus"
    torch.manual_seed(1)
    model = get_model(cfg)
    model.netG.apply(weight_init)
    model.netG.temporal_aggregator.attn_dropout.p = 0.0
    model.train()
    state0 = np_state(model.netG.state_dict())
    x, y, dates = synth(2, 3, 64, 64, 11)
    out = {"x": x.numpy(), "y": y.numpy(), "dates": dates.numpy(),
           "meta": json.dumps(dict(lr=cfg.lr, steps=3, scale_by=cfg.scale_by))}
    for k, v in state0.items():
        out["state/" + k] = v
    ls = []
    for _ in range(3):
        model.set_input({"A": x, "B": y, "dates": dates, "masks": torch.zeros(1)})
        model.optimize_parameters()
        ls.append(model.loss_G.item())
    out["losses"] = np.array(ls)
    for k, v in model.netG.state_dict().items():
        if v.dtype.is_floating_point:
            out["final_sum/" + k] = checksum(v.numpy())
    np.savez_compressed(os.path.join(HERE, "g6_trainseq.npz"), **out)
    print("g6_trainseq", ls)


def case_weightinit():
    """G16: `netG.apply(weight_init)` (train_reconstruct.py:627, weight_init.py:4-74) on a freshly built reference model
    under a fixed seed: checksums of every parameter and buffer.  The build's classes + weight_init must reproduce them
    bit for bit (same module types in the same construction / traversal order consume the same RNG stream)."""
    out = {}
    for tag, kw in (("diag", dict(covmode="diag", out_conv=[26])), ("iso_usev", dict(covmode="iso", out_conv=[14], use_v=True))):
        torch.manual_seed(7)
        m = uncrtaints.UNCRTAINTS(**{**dict(input_dim=15, out_nonlin_mean=True, out_nonlin_var="softplus", scale_by=1.0), **kw})
        m.apply(weight_init)
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                out[f"{tag}/sum/{k}"] = checksum(v.numpy())
        out[f"{tag}/Q"] = m.temporal_encoder.attention_heads.Q.detach().numpy().copy()
        out[f"{tag}/in_conv_bias"] = m.in_conv.conv.conv[0].bias.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g16_weightinit.npz"), **out)
    print("g16_weightinit", len(out))


def case_refcheckpoint():
    """G17: a checkpoint FILE written by the reference's own save_model (model_utils.py:117-125) after one
    optimize_parameters step of a narrow model (widths 32, d_model 64: the file layout does not depend on the widths and the
    fixture stays small), plus checksums of what it holds.  The build's load_checkpoint / load_model must read it."""
    import shutil
    import tempfile
    from parse_args import create_parser
    from src.model_utils import get_model, save_model
    from src.utils import str2list
    cfg = create_parser(mode="train").parse_args([])
    cfg.model, cfg.use_sar, cfg.device, cfg.lr, cfg.input_t = "uncrtaints", True, "cpu", 1e-3, 3
    cfg = str2list(cfg, ["encoder_widths", "decoder_widths", "out_conv"])
    cfg.encoder_widths, cfg.decoder_widths, cfg.d_model = [32], [32, 32], 64
    cfg.out_conv[-1] += 13
    cfg.var_nonLinearity = "softplus"
    torch.manual_seed(5)
    model = get_model(cfg)
    model.netG.apply(weight_init)
    model.train()
    x, y, dates = synth(1, 3, 32, 32, 13)
    model.set_input({"A": x, "B": y, "dates": dates, "masks": torch.zeros(1)})
    model.optimize_parameters()
    tmp = tempfile.mkdtemp()
    cfg.res_dir, cfg.experiment_name = tmp, "exp"
    os.makedirs(os.path.join(tmp, "exp"))
    save_model(cfg, 7, model, "model_epoch_7")
    shutil.copy(os.path.join(tmp, "exp", "model_epoch_7.pth.tar"), os.path.join(HERE, "g17_refcheckpoint.pth.tar"))
    out = {"meta": json.dumps(dict(epoch=7, encoder_widths=[32], decoder_widths=[32, 32], d_model=64, out_conv=cfg.out_conv,
                                   lr=cfg.lr, gamma=cfg.gamma, scale_by=cfg.scale_by))}
    for k, v in model.netG.state_dict().items():
        out["sum/" + k] = checksum(v.numpy().astype(np.float64))
    st = model.optimizer_G.state_dict()["state"]
    out["adam_exp_avg_sum"] = np.array([float(v["exp_avg"].double().sum()) for v in st.values()])
    out["adam_step"] = np.array([float(v["step"]) for v in st.values()])
    # the model's eval-mode output on the saved weights, so a loaded model can be run against it
    model.eval()
    with torch.no_grad():
        o = model.netG(x, batch_positions=dates)
    out["x"], out["dates"], out["eval_out"] = x.numpy(), dates.numpy(), o.numpy()
    np.savez_compressed(os.path.join(HERE, "g17_refcheckpoint.npz"), **out)
    shutil.rmtree(tmp)
    print("g17_refcheckpoint", os.path.getsize(os.path.join(HERE, "g17_refcheckpoint.pth.tar")))


VARIANTS = {
    "att_mean": dict(agg_mode="att_mean"),
    "mean": dict(agg_mode="mean"),
    "separate_out": dict(separate_out=True),
    "is_mono": dict(is_mono=True, n_head=1),
    "instance": dict(encoder_norm="instance", decoder_norm="instance"),     # nn.InstanceNorm2d everywhere (uncrtaints.py:19)
    "enc_batch": dict(encoder_norm="batch"),                                # BatchNorm2d in in_conv / in_block as well
    "elu": dict(out_nonlin_var="elu"),                                      # variance = elu(.) + 1 + eps (uncrtaints.py:226)
    "two_enc": dict(encoder_widths=[128, 128]),                             # one MBConv per entry (uncrtaints.py:316-317)
}


def variant_state(state, name):
    """Derive a variant's state_dict from the g1_diag_t3 weights (same rule in tests/test_variants.py)."""
    st = dict(state)
    if name == "separate_out":
        w, b = st.pop("out_conv.conv.conv.0.weight"), st.pop("out_conv.conv.conv.0.bias")
        st["out_conv_mean_1.conv.conv.0.weight"], st["out_conv_mean_1.conv.conv.0.bias"] = w[:13].clone(), b[:13].clone()
        st["out_conv_var_1.conv.conv.0.weight"], st["out_conv_var_1.conv.conv.0.bias"] = w[13:].clone(), b[13:].clone()
    if name == "is_mono":
        st = {k: v for k, v in st.items() if not k.startswith("temporal_encoder")}
    if name == "enc_batch":     # the GroupNorm affine parameters become BatchNorm ones; fresh running statistics
        import re
        for k in [k for k in st if re.match(r"(in_conv\.conv\.conv\.1|in_block\.\d+\.conv\.(norm|fn\.[148]))\.weight", k)]:
            pre = k[:-len("weight")]
            st[pre + "running_mean"], st[pre + "running_var"] = torch.zeros_like(st[k]), torch.ones_like(st[k])
            st[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    if name == "two_enc":       # the second encoder block takes out_block.1's weights (GroupNorm has no running statistics)
        for k in [k for k in st if k.startswith("out_block.1.") and "running" not in k and "num_batches" not in k]:
            st["in_block.1." + k[len("out_block.1."):]] = st[k].clone()
    if name == "instance":      # InstanceNorm2d has neither parameters nor buffers
        import re
        st = {k: v for k, v in st.items()
              if not re.match(r"(in_conv\.conv\.conv\.1|(in|out)_block\.\d+\.conv\.(norm|fn\.[148]))\.", k)}
    return st


def case_usev():
    """G12: UNCRTAINTS(use_v=True): LTAE2d values (attention-weighted, MLP + BatchNorm1d + ReLU + dropout + GroupNorm)
    up-sampled and merged through include_v (uncrtaints.py:324-338,414-417; ltae.py:10-141,244-307,388-416)."""
    torch.manual_seed(7)
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus",
                              covmode="diag", scale_by=1.0, use_v=True)
    m.apply(weight_init)
    g = torch.Generator().manual_seed(107)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
            mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
        if isinstance(mod, torch.nn.GroupNorm):
            mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
            mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    m.temporal_aggregator.attn_dropout.p = 0.0
    m.temporal_encoder.dropout.p = 0.0                      # LTAE2d's dropout on the values (ltae.py:97,125)
    x, y, dates = synth(2, 3, 64, 64, 7)
    x[1, 2] = 0.0                                           # one padded date
    out = {"x": x.numpy(), "y": y.numpy(), "dates": dates.numpy()}
    for k, v in np_state(m.state_dict()).items():
        out["state/" + k] = v
    m.eval()
    with torch.no_grad():
        oe = m(x, batch_positions=dates)
    out["eval/out"] = oe.numpy()
    m.train()
    xg = x.clone().requires_grad_(True)
    ot = m(xg, batch_positions=dates)
    l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
    l.backward()
    out["train/out"] = ot.detach().numpy()
    out["train/loss"] = np.array(l.item())
    out["train/dx_b0t0"] = xg.grad[0, 0].numpy()
    for k, v in m.named_parameters():
        out["grad/" + k] = v.grad.numpy()
    for k, v in np_state(m.state_dict()).items():
        if "running_" in k:
            out["train/state/" + k] = v
    np.savez_compressed(os.path.join(HERE, "g12_usev.npz"), **out)
    print("g12_usev train loss", l.item(), "keys", len(out))


def case_usev_modes():
    """G20: UNCRTAINTS(use_v=True) with agg_mode 'att_mean' and 'mean' (uncrtaints.py:179-192,211-221 with 324-338,414-417): the weights
    and inputs of g12_usev (loaded from that fixture: only results are stored here) -- eval / train outputs on a stride-8 grid with
    checksums, the train loss, and a checksum of every parameter gradient."""
    g12 = np.load(os.path.join(HERE, "g12_usev.npz"))
    state = {k[len("state/"):]: torch.from_numpy(g12[k]).clone() for k in g12.files if k.startswith("state/")}
    x, y, dates = (torch.from_numpy(g12[k]) for k in ("x", "y", "dates"))
    out = {}
    for mode in ("att_mean", "mean"):
        m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus",
                                  covmode="diag", scale_by=1.0, use_v=True, agg_mode=mode)
        m.load_state_dict(state, strict=True)
        m.temporal_aggregator.attn_dropout.p = 0.0
        m.temporal_encoder.dropout.p = 0.0
        m.eval()
        with torch.no_grad():
            oe = m(x, batch_positions=dates)
        out[f"{mode}/eval_slice"] = oe[:, 0, :, ::8, ::8].numpy()
        out[f"{mode}/eval_checksum"] = checksum(oe.numpy())
        m.train()
        ot = m(x, batch_positions=dates)
        l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
        l.backward()
        out[f"{mode}/train_slice"] = ot.detach()[:, 0, :, ::8, ::8].numpy()
        out[f"{mode}/train_checksum"] = checksum(ot.detach().numpy())
        out[f"{mode}/train_loss"] = np.array(l.item())
        for k, v in m.named_parameters():
            if v.grad is not None:
                out[f"{mode}/gradsum/{k}"] = checksum(v.grad.numpy())
        print("g20_usev_modes", mode, "train loss", l.item())
    np.savez_compressed(os.path.join(HERE, "g20_usev_modes.npz"), **out)


def case_residual():
    """G13: UNCRTAINTS(block_type='residual'): ResidualConvBlock = 3 x (dense conv3x3 reflect + norm + ReLU) + skip
    (uncrtaints.py:24-69).  Small network (2 decoder blocks) to keep the fixture small."""
    torch.manual_seed(9)
    m = uncrtaints.UNCRTAINTS(input_dim=15, decoder_widths=[128, 128], out_conv=[26], out_nonlin_mean=True,
                              out_nonlin_var="softplus", covmode="diag", scale_by=1.0, block_type="residual")
    m.apply(weight_init)
    g = torch.Generator().manual_seed(109)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
            mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
        if isinstance(mod, torch.nn.GroupNorm):
            mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
            mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    m.temporal_aggregator.attn_dropout.p = 0.0
    # the dense 3x3 weights (590 KB each) are NOT stored: they are regenerated from a numpy seed by the tests
    # (tests/conftest.py::residual_conv_weights), in sorted key order
    rng = np.random.default_rng(913)
    sd = m.state_dict()
    big = sorted(k for k, v in sd.items() if v.dim() == 4 and v.shape[-1] == 3 and v.shape[1] > 1)
    for k in big:
        sd[k].copy_(torch.from_numpy((rng.standard_normal(tuple(sd[k].shape)) * 0.03).astype(np.float32)))
    x, y, dates = synth(2, 3, 32, 32, 9)
    out = {"x": x.numpy(), "y": y.numpy(), "dates": dates.numpy(), "regenerated": np.array(big), "regen_seed": np.array(913)}
    for k, v in np_state(m.state_dict()).items():
        if k not in big:
            out["state/" + k] = v
    m.eval()
    with torch.no_grad():
        oe = m(x, batch_positions=dates)
    out["eval/out"] = oe.numpy()
    m.train()
    xg = x.clone().requires_grad_(True)
    ot = m(xg, batch_positions=dates)
    l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
    l.backward()
    out["train/out"] = ot.detach().numpy()
    out["train/loss"] = np.array(l.item())
    out["train/dx_b0t0"] = xg.grad[0, 0].numpy()
    for k, v in m.named_parameters():
        if k in big:       # big tensors: (sum, abs-sum, random projection) + one output-channel slice
            out["gradsum/" + k] = checksum(v.grad.numpy())
            out["gradslice/" + k] = v.grad[::16, ::16].numpy()
        else:
            out["grad/" + k] = v.grad.numpy()
    for k, v in np_state(m.state_dict()).items():
        if "running_" in k:
            out["train/state/" + k] = v
    np.savez_compressed(os.path.join(HERE, "g13_residual.npz"), **out)
    print("g13_residual train loss", l.item(), "keys", len(out))


def case_variants():
    """G2: secondary variants of the UNCRTAINTS class; weights come from g1_diag_t3."""
    base = np.load(os.path.join(HERE, "g1_diag_t3.npz"))
    state = {k[len("state/"):]: torch.from_numpy(base[k]) for k in base.files if k.startswith("state/")}
    x, y, dates = (torch.from_numpy(base[k]) for k in ("x", "y", "dates"))
    out = {}
    for name, kw in VARIANTS.items():
        torch.manual_seed(0)
        m = uncrtaints.UNCRTAINTS(**{**dict(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus",
                                            covmode="diag", scale_by=1.0), **kw})
        m.load_state_dict(variant_state(state, name), strict=True)
        if hasattr(m, "temporal_aggregator"):
            m.temporal_aggregator.attn_dropout.p = 0.0
        xin = x[:, :1] if name == "is_mono" else x
        din = dates[:, :1] if name == "is_mono" else dates
        if name == "mean":
            xin = xin.clone(); xin[1, 0] = 0.0           # a padded date exercises the masked mean
        m.eval()
        with torch.no_grad():
            oe = m(xin, batch_positions=din)
        m.train()
        ot = m(xin, batch_positions=din)
        l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
        l.backward()
        out[f"{name}/eval_slice"] = oe[:, 0, :, ::8, ::8].numpy()
        out[f"{name}/eval_checksum"] = checksum(oe.numpy())
        out[f"{name}/train_slice"] = ot.detach()[:, 0, :, ::8, ::8].numpy()
        out[f"{name}/train_loss"] = np.array(l.item())
        for k, v in m.named_parameters():
            if v.grad is not None:
                out[f"{name}/gradsum/{k}"] = checksum(v.grad.numpy())
        print("variant", name, "train loss", l.item())
    np.savez_compressed(os.path.join(HERE, "g2_variants.npz"), **out)


def case_outconv_layers():
    """G23: out_conv with several layers (`--out_conv "[32,26]"`, parse_args.py:30): ConvBlock(norm='none', last_relu=False) puts a ReLU
    behind every 1x1 convolution but the last (utae.py:476-494, uncrtaints.py:381).  Body weights from g1_diag_t3, the two out_conv
    layers freshly drawn; eval output, train output, loss, every gradient."""
    base = np.load(os.path.join(HERE, "g1_diag_t3.npz"))
    state = {k[len("state/"):]: torch.from_numpy(base[k]) for k in base.files if k.startswith("state/")}
    x, y, dates = (torch.from_numpy(base[k]) for k in ("x", "y", "dates"))
    torch.manual_seed(23)
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[32, 26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
    sd = m.state_dict()
    new = {k: v.clone() for k, v in sd.items() if k.startswith("out_conv.")}
    sd.update({k: v for k, v in state.items() if not k.startswith("out_conv.")})
    m.load_state_dict(sd, strict=True)
    m.temporal_aggregator.attn_dropout.p = 0.0
    out = {f"state/{k}": v.numpy() for k, v in new.items()}
    m.eval()
    with torch.no_grad():
        oe = m(x, batch_positions=dates)
    m.train()
    ot = m(x, batch_positions=dates)
    l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
    l.backward()
    out["eval_out"], out["train_out"], out["train_loss"] = oe.numpy(), ot.detach().numpy(), np.array(l.item())
    for k, v in m.named_parameters():
        if v.grad is not None:
            if k.startswith("out_conv.") or k.startswith("out_block.4") or k.startswith("in_conv"):
                out[f"grad/{k}"] = v.grad.numpy()
            out[f"gradsum/{k}"] = checksum(v.grad.numpy())
    print("outconv layers", [k for k in new], "train loss", l.item())
    np.savez_compressed(os.path.join(HERE, "g23_outconv_layers.npz"), **out)


def case_small_input():
    """G24: an input below 32 x 32 (16 x 16): AdaptiveMaxPool2d((32, 32)) pools UP (uncrtaints.py:403-404) and the aggregator takes its
    AvgPool2d(32 // 16) branch, which has no dropout (uncrtaints.py:197-204).  Weights of g1_diag_t3; one padded date in the second sample."""
    base = np.load(os.path.join(HERE, "g1_diag_t3.npz"))
    state = {k[len("state/"):]: torch.from_numpy(base[k]) for k in base.files if k.startswith("state/")}
    x, y, dates = synth(2, 3, 16, 16, seed=24)
    x[1, 2] = 0.0
    torch.manual_seed(0)
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
    m.load_state_dict(state, strict=True)
    out = dict(x=x.numpy(), y=y.numpy(), dates=dates.numpy())
    m.eval()
    with torch.no_grad():
        oe = m(x, batch_positions=dates)
    m.train()                      # (attn_dropout stays at its default 0.1: the small-map branch applies none)
    ot = m(x, batch_positions=dates)
    l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
    l.backward()
    out["eval_out"], out["train_out"], out["train_loss"] = oe.numpy(), ot.detach().numpy(), np.array(l.item())
    for k, v in m.named_parameters():
        if v.grad is not None:
            out[f"gradsum/{k}"] = checksum(v.grad.numpy())
            if k.startswith("temporal_encoder") or k.startswith("in_conv") or k.startswith("out_conv"):
                out[f"grad/{k}"] = v.grad.numpy()
    print("small input train loss", l.item())
    np.savez_compressed(os.path.join(HERE, "g24_small_input.npz"), **out)


def case_instance_attmean():
    """G25: encoder_norm='instance' + agg_mode='att_mean', B = 1, T = 3 with a zero-padded last date, 64 x 64, default initialisation.
    The combination on which (a) ATen's CPU batch-norm backward mis-reads the strided gradient an einsum-form aggregation hands back
    (the oracle's encoder gradients were orthogonal to these before round 6) and (b) a constant frame meets InstanceNorm
    (uncrtaints.py:16-22): every plane of the padded frame is exactly zero behind in_conv's norm."""
    torch.manual_seed(25)
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0,
                              encoder_norm="instance", agg_mode="att_mean", decoder_widths=[128])
    m.temporal_aggregator.attn_dropout.p = 0.0
    x, y, dates = synth(1, 3, 64, 64, seed=25)
    x[0, 2] = 0.0
    out = {f"state/{k}": v.numpy().copy() for k, v in m.state_dict().items()}
    out.update(x=x.numpy(), y=y.numpy(), dates=dates.numpy())
    m.train()
    ot = m(x, batch_positions=dates)
    l, _ = crit("diag")(ot[:, :, :13], y, ot[:, :, 13:26])
    l.backward()
    out["train_out"], out["train_loss"] = ot.detach().numpy(), np.array(l.item())
    for k, v in m.named_parameters():
        if v.grad is not None:
            out[f"gradsum/{k}"] = checksum(v.grad.numpy())
            if k.startswith("in_block") or k.startswith("in_conv"):
                out[f"grad/{k}"] = v.grad.numpy()
    print("instance + att_mean train loss", l.item())
    np.savez_compressed(os.path.join(HERE, "g25_instance_attmean.npz"), **out)


if __name__ == "__main__":
  if "--only-instance-attmean" in sys.argv:
    case_instance_attmean(); sys.exit(0)
  if "--only-small-input" in sys.argv:
    case_small_input(); sys.exit(0)
  if "--only-outconv-layers" in sys.argv:
    case_outconv_layers(); sys.exit(0)
  if "--skip-model" not in sys.argv:
    case_model("g1_diag_t3", "diag", 2, 3, 64, 64, seed=1, full_grads=True)
    case_model("g1_diag_t3_pad", "diag", 2, 3, 64, 64, seed=1, full_grads=False, pad_last=True, save_state=False, taps=False)
    case_model("g1_iso_t6", "iso", 1, 6, 64, 64, seed=2, full_grads=False)
  if "--only-weightinit" in sys.argv:
    case_weightinit(); sys.exit(0)
  if "--only-refcheckpoint" in sys.argv:
    case_refcheckpoint(); sys.exit(0)
  if "--only-variants" in sys.argv:
    case_variants(); sys.exit(0)
  if "--only-eltlosses" in sys.argv:
    case_eltlosses(); sys.exit(0)
  if "--only-prepare" in sys.argv:
    case_prepare(); sys.exit(0)
  if "--only-metrics" in sys.argv:
    case_metrics(); sys.exit(0)
  if "--only-calibration" in sys.argv:
    case_calibration(); sys.exit(0)
  if "--only-smallmap" in sys.argv:
    case_smallmap(); sys.exit(0)
  if "--only-aggpool" in sys.argv:
    case_aggpool(); sys.exit(0)
  if "--only-attention-rows" in sys.argv:
    case_attention_rows(); sys.exit(0)
  if "--only-ltae2d-deep" in sys.argv:
    case_ltae2d_deep(); sys.exit(0)
  if "--only-ltae2d-nomodel" in sys.argv:
    case_ltae2d_nomodel(); sys.exit(0)
  if "--only-usev" in sys.argv:
    case_usev(); sys.exit(0)
  if "--only-usev-modes" in sys.argv:
    case_usev_modes(); sys.exit(0)
  if "--only-residual" in sys.argv:
    case_residual(); sys.exit(0)
  if "--only-trainseq" not in sys.argv:
    case_variants()
    case_mgnll()
    case_eltlosses()
    case_prepare()
    case_metrics()
    case_calibration()
    case_smallmap()
    case_aggpool()
    case_attention_rows()
    case_ltae2d_deep()
    case_ltae2d_nomodel()
    case_usev()
    case_usev_modes()
    case_residual()
    case_posenc()
    case_ensemble()
    case_weightinit()
    case_refcheckpoint()
    case_outconv_layers()
    case_small_input()
    case_instance_attmean()
  case_trainseq()
