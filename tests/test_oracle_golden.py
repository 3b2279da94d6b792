"""Pins oracle/uncrtaints_oracle.py (the CPU restatement) to fixtures captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest
import torch

from conftest import checksum, compare_param_grads, load_golden, rel_err
from oracle import uncrtaints_oracle as orc

TOL = 2e-5   # fp32 re-association between two CPU formulations of the same graph


def _state(g, prefix="state/"):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _cfg(meta):
    cov = meta["covmode"]
    # fixtures were captured with the aggregator's dropout p set to 0 (deterministic train mode)
    return orc.OracleConfig(covmode=cov, out_conv=[13 + (13 if cov == "diag" else 1)], attn_dropout=0.0)


def _run_case(name, state_from=None):
    g = load_golden(name)
    meta = json.loads(str(g["meta"]))
    cfg = _cfg(meta)
    p = _state(load_golden(state_from) if state_from else g)
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))

    # eval mode
    with torch.no_grad():
        out = orc.forward(p, x, dates, cfg, training=False)
        loss = orc.loss_from_output(out, y, cfg)
    assert rel_err(out.numpy(), g["eval/out"]) < TOL
    assert abs(loss.item() - float(g["eval/loss"])) < TOL * abs(float(g["eval/loss"]))

    # train mode (dropout p=0) + backward
    pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
          for k, v in p.items()}
    xg = x.clone().requires_grad_(True)
    out = orc.forward(pt, xg, dates, cfg, training=True)
    loss = orc.loss_from_output(out, y, cfg)
    loss.backward()
    assert rel_err(out.detach().numpy(), g["train/out"]) < TOL
    assert abs(loss.item() - float(g["train/loss"])) < 5e-5 * abs(float(g["train/loss"]))
    compare_param_grads({k: v.grad.numpy() for k, v in pt.items() if v.requires_grad}, g, tol=2e-4)
    for k in g.files:
        if k.startswith("train/state/"):
            name_ = k[len("train/state/"):]
            assert rel_err(pt[name_].numpy(), g[k]) < TOL, name_
    assert rel_err(xg.grad[0, 0].numpy(), g["train/dx_b0t0"]) < 2e-4
    return g, p, cfg, x, dates


def test_g1_diag_t3_with_taps():
    g, p, cfg, x, dates = _run_case("g1_diag_t3")
    taps = {}
    with torch.no_grad():
        orc.forward(p, x, dates, cfg, training=False, taps=taps)
    assert rel_err(taps["attn"].numpy(), g["eval/attn"]) < TOL
    B, T = x.shape[:2]
    e = taps["e"].reshape(B, T, 128, 64, 64)
    assert rel_err(e[0, 0, ::16].numpy(), g["eval/e_b0t0"]) < TOL
    assert rel_err(taps["agg"][0, ::16].numpy(), g["eval/agg_b0"]) < TOL


def test_g1_diag_t3_pad():
    _run_case("g1_diag_t3_pad", state_from="g1_diag_t3")


def test_g1_iso_t6():
    _run_case("g1_iso_t6")


def test_g3_mgnll_kats():
    g = load_golden("g3_mgnll")
    for i in range(int(g["n"])):
        meta = json.loads(str(g[f"k{i}/meta"]))
        pred = torch.from_numpy(g[f"k{i}/pred"]).requires_grad_(True)
        var = torch.from_numpy(g[f"k{i}/var"]).requires_grad_(True)
        targ = torch.from_numpy(g[f"k{i}/target"])
        for red in ("none", "mean", "sum"):
            l, v = orc.mgnll(pred, targ, var, mode=meta["mode"], reduction=red, want_covariance=(red == "mean"))
            assert l.shape == g[f"k{i}/loss_{red}"].shape
            assert rel_err(l.detach().numpy(), g[f"k{i}/loss_{red}"]) < 1e-5
            if red == "mean":
                assert rel_err(v.numpy(), g[f"k{i}/variance"]) < 1e-6
                gp, gv = torch.autograd.grad(l, (pred, var))
                assert rel_err(gp.numpy(), g[f"k{i}/dpred"]) < 1e-5
                assert rel_err(gv.numpy(), g[f"k{i}/dvar"]) < 1e-5


def test_mgnll_rejects_negative_var_and_bad_reduction():
    pred = torch.rand(1, 1, 13, 2, 2); var = torch.rand(1, 1, 13, 2, 2); var[0, 0, 0, 0, 0] = -1.0
    with pytest.raises(ValueError):
        orc.mgnll(pred, pred, var)
    with pytest.raises(ValueError):
        orc.mgnll(pred, pred, var.abs(), reduction="avg")


def test_g7_positional_table():
    g = load_golden("g7_posenc")
    tab = orc.positional_table(torch.from_numpy(g["dates"]), 16, 1000, 16)
    assert rel_err(tab.numpy(), g["table"]) < 1e-6


def test_g8_ensemble_combine():
    g = load_golden("g8_ensemble")
    mu, var = torch.from_numpy(g["mu"]).double(), torch.from_numpy(g["var"]).double()
    for mode, key in (("both", "var_both"), ("aleatoric", "var_alea"), ("epistemic", "var_epi")):
        m, v = orc.ensemble_combine(mu, var, mode)
        assert rel_err(m.numpy(), g["mean_ens"]) < 1e-12
        assert rel_err(v.numpy(), g[key]) < 1e-10


def test_explicit_formulas_agree_with_aten_ops():
    """The oracle's spelled-out formulas (USE_ATEN = False) and the ATen ops the reference modules call agree."""
    g = load_golden("g1_diag_t3")
    p = _state(g)
    cfg = orc.OracleConfig(attn_dropout=0.0)
    x, dates = torch.from_numpy(g["x"]), torch.from_numpy(g["dates"])
    outs = {}
    for flag in (True, False):
        orc.USE_ATEN = flag
        try:
            with torch.no_grad():
                outs[flag] = orc.forward({k: v.clone() for k, v in p.items()}, x, dates, cfg, training=True)
        finally:
            orc.USE_ATEN = True
    assert rel_err(outs[False].numpy(), outs[True].numpy()) < TOL


def test_eltlosses_match_reference():
    """GaussianNLLLoss / l1 / l2 (get_loss 'GNLL', 'l1', 'l2') restated in the oracle vs the reference's values."""
    g = load_golden("g9_eltlosses")
    for i in range(int(g["n"])):
        pred = torch.from_numpy(g[f"k{i}/pred"]).requires_grad_(True)
        targ = torch.from_numpy(g[f"k{i}/target"])
        var = torch.from_numpy(g[f"k{i}/var"]).requires_grad_(True)
        for red in ("none", "mean", "sum"):
            l, v = orc.gnll(pred, targ, var, eps=1e-8, full=True, reduction=red)
            assert rel_err(l.detach().numpy(), g[f"k{i}/gnll_{red}"]) < 2e-6
        l, v = orc.gnll(pred, targ, var, eps=1e-8, full=True, reduction="mean")
        assert rel_err(v.detach().numpy(), g[f"k{i}/gnll_variance"]) < 1e-7
        gp, gv = torch.autograd.grad(l, (pred, var))
        assert rel_err(gp.numpy(), g[f"k{i}/gnll_dpred"]) < 2e-6 and rel_err(gv.numpy(), g[f"k{i}/gnll_dvar"]) < 2e-6
        for name, fn in (("l1", orc.l1_loss), ("l2", orc.l2_loss)):
            l = fn(pred, targ)
            assert rel_err(l.detach().numpy(), g[f"k{i}/{name}"]) < 2e-6
            assert rel_err(torch.autograd.grad(l, pred)[0].numpy(), g[f"k{i}/{name}_dpred"]) < 2e-6


def _g10_batch(g, T=3):
    k = lambda name: [torch.from_numpy(g[f"batch/{name}/{t}"]) for t in range(T)]
    return {"input": {"S1": k("S1"), "S2": k("S2"), "masks": k("masks"), "S1 TD": k("S1_TD"), "S2 TD": k("S2_TD")},
            "target": {"S2": [torch.from_numpy(g["batch/target"])]}}


def test_prepare_matches_reference():
    """process_MS / process_SAR and prepare_data_multi restated in the oracle vs the reference functions' outputs."""
    g = load_golden("g10_prepare")
    for method in ("default", "resnet"):
        assert np.array_equal(orc.process_ms(g["ms_raw"], method), g[f"ms_{method}"])
        assert np.allclose(orc.process_sar(g["sar_raw"], method), g[f"sar_{method}"], rtol=0, atol=1e-7)
    batch = _g10_batch(g)
    for use_sar, tag in ((True, "sar"), (False, "nosar")):
        x, y, m, dates = orc.prepare_data_multi(batch, use_sar, batch_size=2)
        assert np.array_equal(x.numpy(), g[f"{tag}/x"]) and np.array_equal(y.numpy(), g[f"{tag}/y"])
        assert np.array_equal(m.numpy(), g[f"{tag}/m"]) and np.array_equal(dates.numpy(), g[f"{tag}/dates"])


def test_img_metrics_match_reference():
    """img_metrics + SSIM restated in the oracle vs the reference's values (incl. NaN-holed variances)."""
    g = load_golden("g11_metrics")
    for i in range(int(g["n"])):
        targ, pred, var = (torch.from_numpy(g[f"k{i}/{k}"]) for k in ("target", "pred", "var"))
        d = orc.img_metrics(targ, pred, var)
        for k, v in d.items():
            ref = g[f"k{i}/m/{k}"]
            assert np.allclose(np.asarray(v), ref, rtol=2e-5, atol=1e-6, equal_nan=True), k
        assert np.allclose(orc.ssim(targ, pred, size_average=False).numpy(), g[f"k{i}/ssim_items"], rtol=2e-5)


def test_calibration_metrics_match_reference():
    """compute_ece / compute_uce_auce: the oracle restatement AND the package's vectorised host implementation
    (uncrtaints_amd/src/learning/calibration.py) vs the reference functions' outputs (g14, incl. NaN errors)."""
    from uncrtaints_amd.src.learning import calibration as cal
    g = load_golden("g14_calibration")
    for i in range(int(g["n"])):
        var, err = g[f"k{i}/var"], g[f"k{i}/err"]
        n = len(var)
        for impl in (orc, cal):
            for l2 in (True, False):
                uce, auce = impl.compute_uce_auce(list(var), list(err), n, percent=5, l2=l2)
                ref = g[f"k{i}/uce_{'l2' if l2 else 'l1'}"]
                assert np.allclose([uce, auce], ref, rtol=2e-5, atol=1e-8), (impl.__name__, i, l2, uce, auce, ref)
            ece = impl.compute_ece(list(var), list(err ** 2), n, percent=5)
            assert np.allclose(ece, g[f"k{i}/ece"], rtol=2e-5, atol=1e-9, equal_nan=True), (impl.__name__, i)


def test_calibration_bin_ends_follow_torch_integer_linspace():
    from uncrtaints_amd.src.learning import calibration as cal
    rng = np.random.default_rng(0)
    for n in list(range(1, 400)) + [999, 1000, 1001, 4099, 65537]:
        var, err = rng.random(n) + 0.01, rng.random(n)
        got = cal.compute_ece(list(var), list(err), n)
        ref = orc.compute_ece(list(var), list(err), n)
        assert np.allclose(got, ref, rtol=2e-5, atol=1e-9, equal_nan=True), n


def test_small_map_branch_matches_reference():
    """32 x 32 input (g15): AvgPool branch of the aggregator, no dropout in train mode; weights from g1_diag_t3."""
    g, base = load_golden("g15_smallmap"), load_golden("g1_diag_t3")
    state = {k[len("state/"):]: torch.from_numpy(base[k]) for k in base.files if k.startswith("state/")}
    x, dates = torch.from_numpy(g["x"]), torch.from_numpy(g["dates"])
    cfg = orc.OracleConfig()             # attn_dropout 0.1, as in the reference model
    with torch.no_grad():
        ot = orc.forward({k: v.clone() for k, v in state.items()}, x, dates, cfg, training=True)
        oe = orc.forward({k: v.clone() for k, v in state.items()}, x, dates, cfg, training=False)
    assert rel_err(ot.numpy(), g["train_out"]) < 2e-5 and rel_err(oe.numpy(), g["eval_out"]) < 2e-5


def test_aggregator_avgpool_branch_matches_reference():
    """g18: Compact_Temporal_Aggregator on feature maps smaller than the attention map (AvgPool2d(kernel = w // H), no dropout in
    train mode, uncrtaints.py:197-204), outputs and both gradients, with and without a padded date."""
    g = load_golden("g18_aggpool")
    for i in range(int(g["n"])):
        x, att, pad, gy = (torch.from_numpy(g[f"k{i}/{k}"]) for k in ("x", "att", "pad", "gy"))
        x.requires_grad_(True); att.requires_grad_(True)
        out = orc.temporal_aggregate(x, pad, att, orc.OracleConfig(n_head=att.shape[0]), training=True)
        out.backward(gy)
        assert rel_err(out.detach().numpy(), g[f"k{i}/out"]) < 1e-6
        assert rel_err(x.grad.numpy(), g[f"k{i}/dx"]) < 1e-6 and rel_err(att.grad.numpy(), g[f"k{i}/datt"]) < 1e-6


@pytest.mark.parametrize("i", [0, 1])
def test_oracle_ltae2d_vs_reference_rows_fixture(i):
    """G19 (reference LTAE2d called on its own, eval and train): the oracle's value + attention path reproduces the reference's
    outputs, input gradient and running statistics."""
    import torch
    from oracle import uncrtaints_oracle as orc
    g, pre = load_golden("g19_attention_rows"), f"ltae{i}/"
    training = bool(g[pre + "training"])
    p = {}
    for k in g.files:
        if k.startswith("ltae/state/"):
            p["temporal_encoder." + k[len("ltae/state/"):]] = torch.from_numpy(g[k]).clone()
    nh, dk = p["temporal_encoder.attention_heads.Q"].shape
    cfg = orc.OracleConfig(n_head=nh, d_k=dk, d_model=256, ltae_dropout=0.0)
    x = torch.from_numpy(g["ltae/x"]).clone().requires_grad_(True)
    v, a = orc.ltae2d_values_attention(x, torch.from_numpy(g["ltae/dates"]), torch.from_numpy(g["ltae/pad"]), p, cfg, training)
    # (the fixture keeps the value MLP's BatchNorm bias well above zero: with dead post-ReLU GroupNorm groups out_norm's variance sits
    # at eps and two correct fp32 evaluations of this module differ by 3e-4 -- measured while the fixture was designed)
    assert rel_err(v.detach().numpy(), g[pre + "out"]) < 2e-5
    assert rel_err(a.detach().numpy(), g[pre + "attn"]) < 2e-5
    ((v * torch.from_numpy(g["ltae/gv"])).sum() + (a * torch.from_numpy(g["ltae/ga"])).sum()).backward()
    assert rel_err(x.grad.numpy(), g[pre + "dx"]) < 1e-4
    if training:
        assert rel_err(p["temporal_encoder.mlp.1.running_mean"].numpy(), g[pre + "after/mlp.1.running_mean"]) < 2e-5


@pytest.mark.parametrize("i", [0, 1])
def test_oracle_ltae2d_two_layer_mlp_vs_reference_fixture(i):
    """G21 (reference LTAE2d with mlp=[256, 128, 64], ltae.py:75-84, eval and train, a padded date): outputs, attention, input
    gradient, every parameter gradient, both BatchNorm1d layers' running statistics."""
    import torch
    from oracle import uncrtaints_oracle as orc
    g, pre = load_golden("g21_ltae2d_deep"), f"run{i}/"
    training = bool(g[pre + "training"])
    p = {"temporal_encoder." + k[len("state/"):]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("state/")}
    for k, v in p.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    nh, dk = p["temporal_encoder.attention_heads.Q"].shape
    cfg = orc.OracleConfig(n_head=nh, d_k=dk, d_model=256, ltae_dropout=0.0)
    x = torch.from_numpy(g["x"]).clone().requires_grad_(True)
    v, a = orc.ltae2d_values_attention(x, torch.from_numpy(g["dates"]), torch.from_numpy(g["pad"]), p, cfg, training)
    assert tuple(v.shape) == tuple(g[pre + "out"].shape)
    assert rel_err(v.detach().numpy(), g[pre + "out"]) < 2e-5
    assert rel_err(a.detach().numpy(), g[pre + "attn"]) < 2e-5
    ((v * torch.from_numpy(g["gv"])).sum() + (a * torch.from_numpy(g["ga"])).sum()).backward()
    assert rel_err(x.grad.numpy(), g[pre + "dx"]) < 1e-4
    for k in g.files:
        if k.startswith(pre + "grad/"):
            name = k[len(pre + "grad/"):]
            ref = g[k]
            mine = p["temporal_encoder." + name].grad
            if name.endswith(".bias") and np.abs(ref).max() < 1e-3 * np.abs(g[pre + "grad/" + name.replace(".bias", ".weight")]).max():
                continue        # mathematically zero gradients (a bias ahead of a norm / of the softmax): rounding noise on both sides
            assert rel_err(mine.numpy(), ref) < 2e-4, name
        if training and k.startswith(pre + "after/"):
            assert rel_err(p["temporal_encoder." + k[len(pre + "after/"):]].detach().numpy(), g[k]) < 2e-5, k


@pytest.mark.parametrize("i", [0, 1])
def test_oracle_ltae2d_without_input_projection_vs_reference_fixture(i):
    """G22 (reference LTAE2d(d_model=None), ltae.py:49-54: attention and values on the input channels themselves), eval and train."""
    import torch
    from oracle import uncrtaints_oracle as orc
    g, pre = load_golden("g22_ltae2d_nomodel"), f"run{i}/"
    training = bool(g[pre + "training"])
    p = {"temporal_encoder." + k[len("state/"):]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("state/")}
    assert "temporal_encoder.inconv.weight" not in p
    for k, v in p.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    nh, dk = p["temporal_encoder.attention_heads.Q"].shape
    C = p["temporal_encoder.in_norm.weight"].numel()
    cfg = orc.OracleConfig(n_head=nh, d_k=dk, d_model=C, ltae_dropout=0.0)
    x = torch.from_numpy(g["x"]).clone().requires_grad_(True)
    v, a = orc.ltae2d_values_attention(x, torch.from_numpy(g["dates"]), torch.from_numpy(g["pad"]), p, cfg, training)
    assert rel_err(v.detach().numpy(), g[pre + "out"]) < 2e-5
    assert rel_err(a.detach().numpy(), g[pre + "attn"]) < 2e-5
    ((v * torch.from_numpy(g["gv"])).sum() + (a * torch.from_numpy(g["ga"])).sum()).backward()
    assert rel_err(x.grad.numpy(), g[pre + "dx"]) < 1e-4
    for k in g.files:
        if k.startswith(pre + "grad/"):
            name, ref = k[len(pre + "grad/"):], g[k]
            if name.endswith(".bias") and np.abs(ref).max() < 1e-3 * np.abs(g[pre + "grad/" + name.replace(".bias", ".weight")]).max():
                continue        # mathematically zero gradients: rounding noise on both sides
            assert rel_err(p["temporal_encoder." + name].grad.numpy(), ref) < 2e-4, name


def _g23():
    base, g = load_golden("g1_diag_t3"), load_golden("g23_outconv_layers")
    state = {k: v for k, v in _state(base).items() if not k.startswith("out_conv.")}
    state.update(_state(g))
    x, y, dates = (torch.from_numpy(base[k]) for k in ("x", "y", "dates"))
    return g, state, x, y, dates, orc.OracleConfig(out_conv=[32, 26], attn_dropout=0.0)


def test_oracle_multi_layer_out_conv_vs_reference_fixture():
    """g23: `--out_conv "[32,26]"` (parse_args.py:30) = Conv2d 128->32, ReLU, Conv2d 32->26 (utae.py:476-494, uncrtaints.py:381)."""
    g, state, x, y, dates, cfg = _g23()
    with torch.no_grad():
        oe = orc.forward({k: v.clone() for k, v in state.items()}, x, dates, cfg, training=False)
    assert rel_err(oe.numpy(), g["eval_out"]) < TOL
    pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in state.items()}
    ot = orc.forward(pt, x, dates, cfg, training=True)
    loss = orc.loss_from_output(ot, y, cfg)
    loss.backward()
    assert rel_err(ot.detach().numpy(), g["train_out"]) < TOL
    assert abs(loss.item() - float(g["train_loss"])) < TOL * abs(float(g["train_loss"]))
    class _Sub:          # the fixture's gradient entries, split by tolerance
        def __init__(self, keys):
            self.files = keys

        def __getitem__(self, k):
            return g[k]

    got = {k: v.grad.numpy() for k, v in pt.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    gk = [k for k in g.files if k.startswith("grad/") or k.startswith("gradsum/")]
    # in_conv's gradient on these ill-conditioned `weight_init` weights is the noise-dominated one: two CPU fp32 evaluations of the same
    # graph -- the reference's modules on the fixture host, this restatement here -- sit 2.5e-4 apart on it
    noisy = [k for k in gk if k.split("/", 1)[1].startswith("in_conv.")]
    rep = compare_param_grads(got, _Sub([k for k in gk if k not in noisy]), tol=2e-4)
    rep += compare_param_grads(got, _Sub(noisy), tol=5e-4)
    assert len(rep) >= 90


def test_oracle_small_input_vs_reference_fixture():
    """g24: a 16 x 16 input -- AdaptiveMaxPool2d((32, 32)) pools UP (uncrtaints.py:403-404), the aggregator takes AvgPool2d(2) without
    dropout (uncrtaints.py:197-204); written by the reference in train mode with its default attention dropout of 0.1."""
    base, g = load_golden("g1_diag_t3"), load_golden("g24_small_input")
    state = _state(base)
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
    cfg = orc.OracleConfig()
    with torch.no_grad():
        oe = orc.forward({k: v.clone() for k, v in state.items()}, x, dates, cfg, training=False)
    assert rel_err(oe.numpy(), g["eval_out"]) < TOL
    pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in state.items()}
    ot = orc.forward(pt, x, dates, cfg, training=True)
    loss = orc.loss_from_output(ot, y, cfg)
    loss.backward()
    assert rel_err(ot.detach().numpy(), g["train_out"]) < TOL
    assert abs(loss.item() - float(g["train_loss"])) < 1e-4 * abs(float(g["train_loss"]))
    got = {k: v.grad.numpy() for k, v in pt.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    rep = compare_param_grads(got, g, tol=1e-3)
    assert len(rep) >= 90


def test_oracle_instance_norm_att_mean_vs_reference_fixture():
    """g25, written by the reference: encoder_norm='instance' + agg_mode='att_mean', B = 1, a padded date.  Before round 6 the oracle's
    encoder gradients were ORTHOGONAL to the reference's here (ATen's CPU batch-norm backward on the strided gradient of the einsum-form
    aggregation, oracle._ContiguousGrad) while every forward value agreed -- which rounds 4-5 read as a defect of the HIP path."""
    g = load_golden("g25_instance_attmean")
    state = _state(g)
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
    cfg = orc.OracleConfig(encoder_norm="instance", agg_mode="att_mean", decoder_widths=[128], attn_dropout=0.0)
    pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in state.items()}
    ot = orc.forward(pt, x, dates, cfg, training=True)
    loss = orc.loss_from_output(ot, y, cfg)
    loss.backward()
    assert rel_err(ot.detach().numpy(), g["train_out"]) < TOL
    assert abs(loss.item() - float(g["train_loss"])) < TOL * abs(float(g["train_loss"]))
    got = {k: v.grad.numpy() for k, v in pt.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}

    class _Sub:      # without in_conv's bias: a shift in front of an InstanceNorm, mathematically zero, rounding noise x rstd = 316 on both sides
        files = [k for k in g.files if (k.startswith("grad/") or k.startswith("gradsum/")) and not k.endswith("in_conv.conv.conv.0.bias")]

        def __getitem__(self, k):
            return g[k]

    rep = compare_param_grads(got, _Sub(), tol=2e-4)
    assert len(rep) >= 30
