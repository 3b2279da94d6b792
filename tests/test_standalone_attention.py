"""-m gpu: the attention classes of model/src/backbones/ltae.py called ON THEIR OWN (pixel-major rows [B*H*W, T, d], the layout the
reference defines them on): ScaledDotProductAttentionSmall (ltae.py:431-458), ScaledDotProductAttention (:399-416),
MultiHeadAttentionSmall (:341-385), MultiHeadAttention (:266-307), LTAE2d (:84-141) and the d_model=None variants (:49-54, :177-182).
Checked against the same arithmetic in fp32 torch on the CPU (forward and every gradient) AND against outputs of the reference's own
classes (fixture g19_attention_rows, made by tests/golden/make_golden.py::case_attention_rows)."""
import math

import numpy as np

import pytest
import torch

from gpu_util import DEV, close, dev, rand

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import uncrtaints_oracle
    return uncrtaints_oracle


def _sdpa_ref(q, k, v, pad, temperature):
    """score = q.k / temperature, masked_fill(pad, -1e3), softmax over T, attn @ v  (ltae.py:432-452)"""
    score = torch.einsum("md,mtd->mt", q, k) / temperature
    comp = score.masked_fill(pad, -1e3) if pad is not None else score
    attn = torch.softmax(comp, dim=1)
    out = torch.einsum("mt,mtd->md", attn, v) if v is not None else None
    return attn, out, comp


@pytest.mark.parametrize("T,dk,dv,padded", [(3, 4, 16, False), (6, 4, 16, True), (12, 8, 8, True), (1, 4, 4, False),
                                            (100, 4, 8, True)])      # 100 dates: no limit on the sequence length (round 4)
def test_scaled_dot_product_attention_small_rows(T, dk, dv, padded):
    from uncrtaints_amd.src.backbones.ltae import ScaledDotProductAttentionSmall
    m = 700                                                  # not a multiple of the block size
    q, k, v = rand(m, dk, seed=1), rand(m, T, dk, seed=2), rand(m, T, dv, seed=3)
    pad = None
    if padded:
        pad = torch.rand(m, T, generator=torch.Generator().manual_seed(4)) < 0.3
        pad[:, 0] = False
    ga, go, gc = rand(m, 1, T, seed=5), rand(m, 1, dv, seed=6), rand(m, 1, T, seed=7)
    temp = math.sqrt(dk)
    qo, ko, vo = (t.clone().requires_grad_(True) for t in (q, k, v))
    a_ref, o_ref, c_ref = _sdpa_ref(qo, ko, vo, pad, temp)
    ((a_ref * ga[:, 0]).sum() + (o_ref * go[:, 0]).sum() + (c_ref * gc[:, 0]).sum()).backward()
    mod = ScaledDotProductAttentionSmall(temperature=temp)
    qd, kd, vd = (dev(t).requires_grad_(True) for t in (q, k, v))
    out, attn, comp = mod(qd, kd, vd, pad_mask=dev(pad) if pad is not None else None, return_comp=True, weight_v=True)
    assert attn.shape == (m, 1, T) and out.shape == (m, 1, dv) and comp.shape == (m, 1, T)
    close("sdpa_attn", attn[:, 0], a_ref)
    close("sdpa_out", out[:, 0], o_ref)
    close("sdpa_comp", comp[:, 0], c_ref)
    ((attn * dev(ga)).sum() + (out * dev(go)).sum() + (comp * dev(gc)).sum()).backward()
    close("sdpa_dq", qd.grad, qo.grad)
    close("sdpa_dk", kd.grad, ko.grad)
    close("sdpa_dv", vd.grad, vo.grad)
    # attention only (weight_v=False): one tensor, like the reference
    only = mod(dev(q), dev(k), dev(v), pad_mask=dev(pad) if pad is not None else None)
    assert torch.is_tensor(only) and torch.equal(only, attn.detach())


@pytest.mark.parametrize("cls_name,weight_v", [("MultiHeadAttentionSmall", False), ("MultiHeadAttentionSmall", True),
                                               ("MultiHeadAttention", True)])
def test_multi_head_attention_rows(cls_name, weight_v):
    from uncrtaints_amd.src.backbones import ltae
    torch.manual_seed(0)
    nh, dk, d_in, n, T = 16, 4, 256, 300, 3
    mod = getattr(ltae, cls_name)(n_head=nh, d_k=dk, d_in=d_in) if cls_name.endswith("Small") \
        else ltae.MultiHeadAttention(n_head=nh, d_k=dk, d_in=d_in, use_dropout=False)
    with torch.no_grad():
        mod.fc1_k.bias.copy_(0.3 * torch.randn(nh * dk))
    v = rand(n, T, d_in, seed=1)
    pad = torch.zeros(n, T, dtype=torch.bool)
    pad[::3, T - 1] = True
    # reference arithmetic (ltae.py:341-385): keys per head = slices of the projected values, one learned query per head
    W, b, Q = (t.detach().clone().requires_grad_(True) for t in (mod.fc1_k.weight, mod.fc1_k.bias, mod.Q))
    vo = v.clone().requires_grad_(True)
    k = (vo @ W.t() + b).view(n, T, nh, dk)
    score = torch.einsum("hd,nthd->hnt", Q, k) / math.sqrt(dk)
    attn_ref = torch.softmax(score.masked_fill(pad[None], -1e3), dim=2)                  # [nh, n, T]
    out_ref = torch.einsum("hnt,nthd->hnd", attn_ref, vo.view(n, T, nh, d_in // nh))     # [nh, n, d_in / nh]
    ga, go = rand(nh, n, T, seed=2), rand(nh, n, d_in // nh, seed=3)
    ((attn_ref * ga).sum() + ((out_ref * go).sum() if weight_v else 0.0)).backward()
    md = mod.to(DEV).eval()
    vd = dev(v).requires_grad_(True)
    if cls_name.endswith("Small"):
        res = md(vd, pad_mask=dev(pad), weight_v=weight_v)
    else:
        res = md(vd, pad_mask=dev(pad))
    if weight_v:
        out, attn = res
        close("mha_out", out, out_ref)
    else:
        attn, out = res, None
    assert attn.shape == (nh, n, T)
    close("mha_attn", attn, attn_ref)
    ((attn * dev(ga)).sum() + ((out * dev(go)).sum() if weight_v else 0.0)).backward()
    close("mha_dv", vd.grad, vo.grad)
    close("mha_dW", md.fc1_k.weight.grad, W.grad)
    if weight_v:
        close("mha_db", md.fc1_k.bias.grad, b.grad) if float(b.grad.abs().max()) > 1e-4 * float(W.grad.abs().max()) else None
    # a key bias shifts every date's score alike: the softmax does not see it (zero gradient, rounding noise on both sides)
    assert float(md.fc1_k.bias.grad.abs().max()) < 1e-3 * float(W.grad.abs().max())
    close("mha_dQ", md.Q.grad, Q.grad)


def test_attention_dropout_stream_of_the_standalone_classes():
    """ScaledDotProductAttention in train mode: this library's counter-based dropout (statistics, scaling, consistency between the
    returned attention and the weighted values; the backward uses the same mask)."""
    from uncrtaints_amd.src.backbones.ltae import ScaledDotProductAttention
    m, T, dk, dv = 4096, 4, 4, 8
    mod = ScaledDotProductAttention(temperature=2.0, attn_dropout=0.25).train()
    q, k, v = dev(rand(m, dk, seed=1)), dev(rand(m, T, dk, seed=2)), dev(rand(m, T, dv, seed=3)).requires_grad_(True)
    out, attn = mod(q, k, v)
    a_ref, _, _ = _sdpa_ref(q.cpu(), k.cpu(), None, None, 2.0)
    kept = attn[:, 0].cpu() != 0
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - 0.25) < 0.02, frac
    close("dropout_scaled", attn[:, 0].cpu()[kept], (a_ref / 0.75)[kept])
    close("dropout_out", out[:, 0], torch.einsum("mt,mtd->md", attn[:, 0], v.detach()))
    out.sum().backward()
    close("dropout_dv", v.grad, attn[:, 0].detach()[:, :, None].expand(m, T, dv))
    mod.eval()
    out_e, attn_e = mod(q, k, v.detach())
    close("eval_attn", attn_e[:, 0], a_ref)


@pytest.mark.parametrize("d_model,training", [(256, False), (256, True), (None, False)])
def test_ltae2d_standalone(orc, d_model, training):
    """LTAE2d called on its own ([B,T,C,h,w] -> values [B,C,h,w] + attention), and without the input projection (d_model=None)."""
    from uncrtaints_amd.src.backbones.ltae import LTAE2d
    torch.manual_seed(2)
    C, nh, dk, B, T = 128, 16, 4, 2, 3
    dm = d_model if d_model is not None else C
    m = LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=[dm, C], dropout=0.0, d_model=d_model, return_att=True, use_dropout=False)
    with torch.no_grad():      # default initialisation + non-trivial norm parameters, biases and running statistics
        m.in_norm.weight.copy_(1.0 + 0.3 * torch.randn(C)); m.in_norm.bias.copy_(0.2 * torch.randn(C))
        m.out_norm.weight.copy_(1.0 + 0.3 * torch.randn(C)); m.out_norm.bias.copy_(0.2 * torch.randn(C))
        m.mlp[1].weight.copy_(1.0 + 0.3 * torch.randn(C)); m.mlp[1].bias.copy_(0.2 * torch.randn(C))
        m.mlp[1].running_mean.copy_(0.1 * torch.randn(C)); m.mlp[1].running_var.copy_(0.5 + torch.rand(C))
        m.attention_heads.fc1_k.bias.copy_(0.3 * torch.randn(nh * dk))
    m.train(training)
    down = rand(B, T, C, 32, 32, seed=3)
    dates = torch.sort(torch.randint(1400, 1800, (B, T)), dim=1).values.float()
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[1, T - 1] = True
    gv, ga = rand(B, C, 32, 32, seed=4), rand(nh, B, T, 32, 32, seed=5)
    cfg = orc.OracleConfig(n_head=nh, d_k=dk, d_model=dm, ltae_dropout=0.0)
    p = {"temporal_encoder." + k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
                                    else v.detach().clone()) for k, v in m.state_dict().items()}
    if d_model is None:      # no projection: the oracle takes the identity
        p["temporal_encoder.inconv.weight"] = torch.eye(C).view(C, C, 1)
        p["temporal_encoder.inconv.bias"] = torch.zeros(C)
    do = down.clone().requires_grad_(True)
    v_ref, a_ref = orc.ltae2d_values_attention(do, dates, pad, p, cfg, training)
    ((v_ref * gv).sum() + (a_ref * ga).sum()).backward()
    md = m.to(DEV)
    dd = dev(down).requires_grad_(True)
    v, a = md(dd, batch_positions=dev(dates), pad_mask=dev(pad))
    close(f"ltae2d_values[d_model={d_model},train={training}]", v, v_ref)
    close("ltae2d_attn", a, a_ref)
    ((v * dev(gv)).sum() + (a * dev(ga)).sum()).backward()
    close("ltae2d_ddown", dd.grad, do.grad)
    for k, par in md.named_parameters():
        ref = p["temporal_encoder." + k].grad
        if ref is None or ref.abs().max() < 1e-6 * max(1.0, float(par.grad.abs().max())):
            continue
        # a key bias shifts every date's score alike (the softmax does not see it); in train mode the batch-statistics BatchNorm1d
        # behind the value MLP removes every per-channel shift ahead of it: zero gradients, rounding noise on both sides
        if k == "attention_heads.fc1_k.bias" or (training and k in ("inconv.bias", "mlp.0.bias", "in_norm.bias")):
            sib = md.get_parameter(k.replace(".bias", ".weight")).grad
            assert float(par.grad.abs().max()) < 1e-3 * float(sib.abs().max()), k
            continue
        close(f"ltae2d_grad[{k}]", par.grad, ref)
    if training:
        close("ltae2d_running_mean", md.mlp[1].running_mean, p["temporal_encoder.mlp.1.running_mean"])


def test_ltae2dtiny_without_projection(orc):
    from uncrtaints_amd.src.backbones.ltae import LTAE2dtiny
    torch.manual_seed(4)
    C, nh, dk, B, T = 128, 16, 4, 2, 3
    m = LTAE2dtiny(in_channels=C, n_head=nh, d_k=dk, d_model=None)
    down = rand(B, T, C, 32, 32, seed=3)
    dates = torch.sort(torch.randint(1400, 1800, (B, T)), dim=1).values.float()
    pad = torch.zeros(B, T, dtype=torch.bool)
    cfg = orc.OracleConfig(n_head=nh, d_k=dk, d_model=C)
    p = {"temporal_encoder." + k: v.detach().clone() for k, v in m.state_dict().items()}
    p["temporal_encoder.inconv.weight"], p["temporal_encoder.inconv.bias"] = torch.eye(C).view(C, C, 1), torch.zeros(C)
    a_ref = orc.ltae_tiny_attention(down, dates, pad, p, cfg)
    a = m.to(DEV)(dev(down), batch_positions=dev(dates), pad_mask=dev(pad))
    close("ltae_tiny_no_projection", a, a_ref)


# ---- the same classes against outputs of the REFERENCE's own classes (fixture g19_attention_rows, written by
# tests/golden/make_golden.py::case_attention_rows from /root/reference/model/src/backbones/ltae.py) ------------------------------
@pytest.fixture(scope="module")
def g19():
    from conftest import load_golden
    return load_golden("g19_attention_rows")


def _t(g, k):
    return torch.from_numpy(g[k])


@pytest.mark.parametrize("i", [0, 1, 2])
def test_sdpa_rows_vs_reference_fixture(g19, i):
    from uncrtaints_amd.src.backbones import ltae
    g, pre = g19, f"sdpa{i}/"
    cls = str(g[pre + "cls"])
    temp = float(g[pre + "temperature"])
    q, k, v = (dev(_t(g, pre + n)).requires_grad_(True) for n in ("q", "k", "v"))
    pad = dev(_t(g, pre + "pad"))
    if cls.endswith("Small"):
        out, attn, comp = ltae.ScaledDotProductAttentionSmall(temperature=temp)(q, k, v, pad_mask=pad, return_comp=True, weight_v=True)
    else:
        out, attn, comp = ltae.ScaledDotProductAttention(temperature=temp, attn_dropout=0.1).eval()(q, k, v, pad_mask=pad, return_comp=True)
    close(f"g19 {cls} attn", attn, _t(g, pre + "attn"))
    close(f"g19 {cls} out", out, _t(g, pre + "out"))
    close(f"g19 {cls} comp", comp, _t(g, pre + "comp"))
    ((attn * dev(_t(g, pre + "ga"))).sum() + (out * dev(_t(g, pre + "go"))).sum() + (comp * dev(_t(g, pre + "gc"))).sum()).backward()
    for n, t in (("dq", q), ("dk", k), ("dv", v)):
        close(f"g19 {cls} {n}", t.grad, _t(g, pre + n))


@pytest.mark.parametrize("i", [0, 1, 2])
def test_multi_head_attention_rows_vs_reference_fixture(g19, i):
    from uncrtaints_amd.src.backbones import ltae
    g, pre = g19, f"mha{i}/"
    cls, weight_v = str(g[pre + "cls"]), bool(g[pre + "weight_v"])
    nh, dk = g[pre + "Q"].shape
    d_in = g[pre + "W"].shape[1]
    mod = ltae.MultiHeadAttentionSmall(n_head=nh, d_k=dk, d_in=d_in) if cls.endswith("Small") \
        else ltae.MultiHeadAttention(n_head=nh, d_k=dk, d_in=d_in, use_dropout=False)
    with torch.no_grad():
        mod.fc1_k.weight.copy_(_t(g, pre + "W")); mod.fc1_k.bias.copy_(_t(g, pre + "b")); mod.Q.copy_(_t(g, pre + "Q"))
    mod = mod.to(DEV).eval()
    v = dev(_t(g, pre + "v")).requires_grad_(True)
    pad = dev(_t(g, pre + "pad"))
    if cls.endswith("Small"):
        res = mod(v, pad_mask=pad, weight_v=weight_v)
        out, attn = res if weight_v else (None, res)
    else:
        out, attn = mod(v, pad_mask=pad)
    close(f"g19 {cls} attn", attn, _t(g, pre + "attn"))
    if out is not None:
        close(f"g19 {cls} out", out, _t(g, pre + "out"))
    ((attn * dev(_t(g, pre + "ga"))).sum() + ((out * dev(_t(g, pre + "go"))).sum() if out is not None else 0.0)).backward()
    close(f"g19 {cls} dv", v.grad, _t(g, pre + "dv"))
    close(f"g19 {cls} dW", mod.fc1_k.weight.grad, _t(g, pre + "dW"))
    close(f"g19 {cls} dQ", mod.Q.grad, _t(g, pre + "dQ"))
    # the key bias shifts every date's score alike: mathematically zero gradient, rounding noise on both sides
    assert float(mod.fc1_k.bias.grad.abs().max()) < 1e-3 * float(mod.fc1_k.weight.grad.abs().max())
    assert float(np.abs(g[pre + "db"]).max()) < 1e-3 * float(np.abs(g[pre + "dW"]).max())


@pytest.mark.parametrize("i", [0, 1])
def test_ltae2d_vs_reference_fixture(g19, i):
    from uncrtaints_amd.src.backbones.ltae import LTAE2d
    g, pre = g19, f"ltae{i}/"
    training = bool(g[pre + "training"])
    state = {k[len("ltae/state/"):]: _t(g, k) for k in g.files if k.startswith("ltae/state/")}
    C = state["in_norm.weight"].numel()
    nh, dk = state["attention_heads.Q"].shape
    m = LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=[256, C], dropout=0.0, d_model=256, return_att=True, use_dropout=False)
    m.load_state_dict(state, strict=True)      # (a 32 x 32 map, one sample with a padded date; the same weights and inputs in both modes)
    m = m.to(DEV).train(training)
    x = dev(_t(g, "ltae/x")).requires_grad_(True)
    out, attn = m(x, batch_positions=dev(_t(g, "ltae/dates")), pad_mask=dev(_t(g, "ltae/pad")))
    VT = 1e-4      # the plain contract (the fixture keeps out_norm's 8-value groups alive: see make_golden.py::case_attention_rows)
    close(f"g19 LTAE2d[train={training}] values", out, _t(g, pre + "out"), tol=VT)
    close(f"g19 LTAE2d[train={training}] attn", attn, _t(g, pre + "attn"))
    ((out * dev(_t(g, "ltae/gv"))).sum() + (attn * dev(_t(g, "ltae/ga"))).sum()).backward()
    close("g19 LTAE2d dx", x.grad, _t(g, pre + "dx"), tol=VT)
    for k, par in m.named_parameters():
        ref = _t(g, pre + "grad/" + k)
        if k == "attention_heads.fc1_k.bias" or (training and k in ("inconv.bias", "mlp.0.bias", "in_norm.bias")):
            sib = m.get_parameter(k.replace(".bias", ".weight")).grad       # zero gradients (see test_ltae2d_standalone)
            assert float(par.grad.abs().max()) < 1e-3 * float(sib.abs().max()), k
            continue
        close(f"g19 LTAE2d grad[{k}]", par.grad, ref, tol=VT)
    if training:
        for k in ("mlp.1.running_mean", "mlp.1.running_var"):
            close("g19 LTAE2d " + k, m.state_dict()[k], _t(g, pre + "after/" + k))


# ---- a value MLP with more than one layer (ltae.py:75-84): fixture g21_ltae2d_deep written by the reference's LTAE2d(mlp=[256, 128, 64]) ----
@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1])
def test_ltae2d_two_layer_mlp_vs_reference_fixture(i):
    from conftest import load_golden
    from uncrtaints_amd.src.backbones.ltae import LTAE2d
    g, pre = load_golden("g21_ltae2d_deep"), f"run{i}/"
    training = bool(g[pre + "training"])
    state = {k[len("state/"):]: _t(g, k) for k in g.files if k.startswith("state/")}
    C = state["in_norm.weight"].numel()
    nh, dk = state["attention_heads.Q"].shape
    mlp = [int(v) for v in g["mlp"]]
    m = LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=mlp, dropout=0.0, d_model=256, return_att=True, use_dropout=False)
    assert list(m.state_dict().keys()) == [k[len("state/"):] for k in g.files if k.startswith("state/")]      # the reference's keys, in order
    m.load_state_dict(state, strict=True)
    m = m.to(DEV).train(training)
    x = dev(_t(g, "x")).requires_grad_(True)
    out, attn = m(x, batch_positions=dev(_t(g, "dates")), pad_mask=dev(_t(g, "pad")))
    assert tuple(out.shape) == tuple(g[pre + "out"].shape)
    VT = 1e-4
    close(f"g21 LTAE2d[train={training}] values", out, _t(g, pre + "out"), tol=VT)
    close(f"g21 LTAE2d[train={training}] attn", attn, _t(g, pre + "attn"))
    ((out * dev(_t(g, "gv"))).sum() + (attn * dev(_t(g, "ga"))).sum()).backward()
    close("g21 LTAE2d dx", x.grad, _t(g, pre + "dx"), tol=VT)
    for k, par in m.named_parameters():
        ref = _t(g, pre + "grad/" + k)
        if k.endswith(".bias") and k != "out_norm.bias":
            sib_ref = _t(g, pre + "grad/" + k.replace(".bias", ".weight"))
            if float(ref.abs().max()) < 1e-3 * float(sib_ref.abs().max()):      # zero gradients (a bias ahead of a norm / the softmax)
                assert float(par.grad.abs().max()) < 1e-3 * float(m.get_parameter(k.replace(".bias", ".weight")).grad.abs().max()), k
                continue
        close(f"g21 LTAE2d grad[{k}]", par.grad, ref, tol=VT)
    if training:
        for k in g.files:
            if k.startswith(pre + "after/"):
                close("g21 LTAE2d " + k, m.state_dict()[k[len(pre + "after/"):]], _t(g, k))


# ---- LTAE2d without its input projection (d_model=None, ltae.py:49-54): fixture g22_ltae2d_nomodel written by the reference ----
@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1])
def test_ltae2d_without_input_projection_vs_reference_fixture(i):
    from conftest import load_golden
    from uncrtaints_amd.src.backbones.ltae import LTAE2d
    g, pre = load_golden("g22_ltae2d_nomodel"), f"run{i}/"
    training = bool(g[pre + "training"])
    state = {k[len("state/"):]: _t(g, k) for k in g.files if k.startswith("state/")}
    C = state["in_norm.weight"].numel()
    nh, dk = state["attention_heads.Q"].shape
    m = LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=[int(v) for v in g["mlp"]], dropout=0.0, d_model=None, return_att=True, use_dropout=False)
    assert list(m.state_dict().keys()) == [k[len("state/"):] for k in g.files if k.startswith("state/")]
    m.load_state_dict(state, strict=True)
    m = m.to(DEV).train(training)
    x = dev(_t(g, "x")).requires_grad_(True)
    out, attn = m(x, batch_positions=dev(_t(g, "dates")), pad_mask=dev(_t(g, "pad")))
    VT = 1e-4
    close(f"g22 LTAE2d[train={training}] values", out, _t(g, pre + "out"), tol=VT)
    close(f"g22 LTAE2d[train={training}] attn", attn, _t(g, pre + "attn"))
    ((out * dev(_t(g, "gv"))).sum() + (attn * dev(_t(g, "ga"))).sum()).backward()
    close("g22 LTAE2d dx", x.grad, _t(g, pre + "dx"), tol=VT)
    for k, par in m.named_parameters():
        ref = _t(g, pre + "grad/" + k)
        if k.endswith(".bias") and k != "out_norm.bias":
            sib_ref = _t(g, pre + "grad/" + k.replace(".bias", ".weight"))
            if float(ref.abs().max()) < 1e-3 * float(sib_ref.abs().max()):
                assert float(par.grad.abs().max()) < 1e-3 * float(m.get_parameter(k.replace(".bias", ".weight")).grad.abs().max()), k
                continue
        close(f"g22 LTAE2d grad[{k}]", par.grad, ref, tol=VT)
