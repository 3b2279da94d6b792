"""-m gpu: bf16 activation storage with fp32 accumulation (BASELINE config 3, `UNCRTAINTS.set_act_dtype(torch.bfloat16)`).

Tolerance contract of this mode (the fp32 mode keeps the 1e-4 contract of BASELINE.json; bf16 storage cannot: one rounding is
2^-9 = 2e-3 relative, and a step chains ~40 stored tensors):

  kernel level   every bf16-storage kernel against an fp64 evaluation of the same formula ON THE ROUNDED INPUTS:
                 the stored result is within one bf16 rounding of it (|err| <= 2^-8 |value| + tiny), and the statistics a
                 producer emits equal the sums of the values it STORED (fp32 summation error only, <= 2e-5);
  model level    a seeded default-initialised network against the fp32 CPU oracle: outputs max|err| / max|ref| <= 4e-2, loss
                 within 5e-3 relative, every parameter gradient (and the input gradient) relative L2 error <= 1.5e-1 with cosine
                 similarity >= 0.99 (measured 1.9e-2 / 3.8e-4 / 7.8e-2 / 0.9975); against the oracle with the SAME roundings
                 emulated (OracleConfig.act_bf16): 3e-2 / 2e-3 / 1.2e-1 / 0.995 (measured 1.4e-2 / 3.1e-4 / 6.2e-2 / 0.998: two
                 realisations of the same roundings differ in accumulation order and tie breaks, and the network amplifies that
                 like the roundings themselves).  Measured values are printed.  The golden fixture's weight_init weights are ill-conditioned (the
                 emulating oracle itself is 5.8e-2 / 9.6e-2 from fp32): there the HIP path is held to the emulating oracle;
  training       three Adam steps follow the emulated bf16 loss sequence and the fp32 one within 20 %, monotonically decreasing.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_util import DEV, dev

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def rb(t):
    """fp32 tensor rounded to bf16 and back (what a bf16-storage kernel reads)."""
    return t.to(BF).float()


def within_one_rounding(name, got, ref64, extra=0.0, flip_frac=0.0, flip_extra=0.0):
    """got (bf16 tensor) vs the fp64 value of the same formula: |err| <= 2^-8 |ref| (one round-to-nearest; the fp32 evaluation
    ahead of the rounding may tip a tie) + `extra` * max|ref| for fp32 accumulation noise.
    flip_frac / flip_extra (GEMM-type kernels, whose OPERAND is rounded to bf16 inside the kernel): the kernel's fp32 prologue and
    the fp64 reference can round an operand element to different neighbours (probability 2^-16 per element); the few outputs
    that contain such an element may exceed the bound, by at most `flip_extra` * max|ref|."""
    g = got.float().double().cpu()
    r = ref64.double().cpu()
    mx = float(r.abs().max())
    err = (g - r).abs()
    bad = err > 2.0 ** -8 * r.abs() + extra * mx + 1e-30
    frac = float(bad.double().mean())
    worst = float((err / (r.abs() + 1e-3 * mx)).max())
    print(f"[bf16] {name}: fraction beyond one rounding {frac:.2e} (allowed {flip_frac:g}), worst relative {worst:.2e}")
    assert frac <= flip_frac, (name, frac, worst)
    assert bool((err <= 2.0 ** -8 * r.abs() + (extra + flip_extra) * mx + 1e-30).all()), (name, worst)


def part_sums(part, slots):
    return part.view(-1, slots, 2).double().sum(dim=1)


def test_ew_kernels_bf16_storage():
    from uncrtaints_amd import engine as E
    g = torch.Generator().manual_seed(0)
    N, C, P = 2, 8, 4096
    a, b, c, h = (torch.randn(N, C, P, generator=g) for _ in range(4))
    A, B = torch.randn(N * C, generator=g), torch.randn(N * C, generator=g)
    ab, bb, cb, hb_ = (dev(t).to(BF) for t in (a, b, c, h))
    ar, br, cr, hr = (rb(t).double() for t in (a, b, c, h))
    Ad, Bd = A.double().view(N, C, 1), B.double().view(N, C, 1)
    # residual: out = a + A*b + B, stats (sum, sum^2) of the stored values
    out = torch.empty_like(ab)
    _, part = E.ew(E.EW_RESIDUAL, ab, b=bb, out=out, k=(dev(A), dev(B), None, None), want_part=True, planes=N * C, P=P)
    within_one_rounding("ew_residual", out, ar + Ad * br + Bd, extra=1e-6)
    st = part_sums(part.buf, part.slots).cpu()
    o64 = out.float().double().cpu().view(N * C, P)
    assert torch.allclose(st[:, 0], o64.sum(1), rtol=2e-5, atol=1e-3) and torch.allclose(st[:, 1], (o64 ** 2).sum(1), rtol=2e-5)
    # norm-apply + ReLU
    out2 = torch.empty_like(ab)
    _, part = E.ew(E.EW_AFFINE_RELU, ab, out=out2, k=(dev(A), dev(B), None, None), want_part=True, planes=N * C, P=P)
    within_one_rounding("ew_affine_relu", out2, torch.relu(Ad * ar + Bd), extra=1e-6)
    # statistics-only ops read bf16
    part = E.stats_aux(ab, bb, N * C, P)
    st = part_sums(part.buf, part.slots).cpu()
    assert torch.allclose(st[:, 1], (ar * br).view(N * C, P).sum(1), rtol=2e-5, atol=2e-3)
    # ReLU backward mask
    out3 = torch.empty_like(ab)
    E.ew(E.EW_RELU_BWD, ab, b=bb, out=out3, k=(dev(A), dev(B), None, None), want_part=True, planes=N * C, P=P)
    assert torch.equal(out3.float().cpu(), torch.where(Ad * br + Bd > 0, ar, torch.zeros_like(ar)).float())
    # cast round trip
    x = dev(a)
    assert torch.equal(E.cast(E.cast(x, E.BF16), E.F32).cpu(), rb(a))


@pytest.mark.parametrize("H,W", [(64, 256), (32, 64)])      # the row-streaming kernels (W == 256) and the LDS-tiled ones
def test_depthwise_bf16_storage(H, W):
    from uncrtaints_amd import engine as E
    import uncrtaints_amd.hip_backend as hb
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(1)
    N, C = 2, 8
    h1 = torch.randn(N, C, H, W, generator=g)
    A, B = 0.5 + torch.rand(N * C, generator=g), 0.3 * torch.randn(N * C, generator=g)
    w = 0.3 * torch.randn(C, 3, 3, generator=g)
    h1b = dev(h1).to(BF)
    h2 = torch.empty_like(h1b)
    slots = hb.query("uncr_dw_slots_fwd", H)
    part = torch.empty(N * C, slots, 2, device=DEV)
    hb.call("uncr_dw_fwd", h1b, dev(A), dev(B), dev(w.reshape(C, 9)), h2, part, N, C, H, W, 1, 0, E._stream())
    u = A.double().view(N, C, 1, 1) * rb(h1).double() + B.double().view(N, C, 1, 1)
    g1 = 0.5 * u * (1.0 + torch.erf(u / 2 ** 0.5))
    ref = F.conv2d(F.pad(g1, (1, 1, 1, 1), mode="reflect"), w.double().view(C, 1, 3, 3), groups=C)
    within_one_rounding(f"dw_fwd[{H}x{W}]", h2, ref, extra=2e-6)
    st = part_sums(part, slots).cpu()
    o64 = h2.float().double().cpu().view(N * C, -1)
    assert torch.allclose(st[:, 0], o64.sum(1), rtol=2e-5, atol=2e-3) and torch.allclose(st[:, 1], (o64 ** 2).sum(1), rtol=2e-5)
    # backward: du1 = gelu'(u1) * dw^T(C1*du2 + C2*h2 + C3) on the stored tensors
    du2 = torch.randn(N, C, H, W, generator=g)
    c1, c2, c3 = (0.5 * torch.randn(N * C, generator=g) for _ in range(3))
    du2b = dev(du2).to(BF)
    du1 = torch.empty_like(h1b)
    sb = hb.query("uncr_dw_slots_bwd", H)
    partb = torch.empty(N * C, sb, 2, device=DEV)
    dwp = torch.empty(N * C, sb, 9, device=DEV)
    hb.call("uncr_dw_bwd", du2b, h2, h1b, dev(c1), dev(c2), dev(c3), None, dev(A), dev(B), dev(w.reshape(C, 9)), du1, partb, dwp,
            None, 0, N, C, H, W, 1, 0, None, E._stream())
    h1r = rb(h1).double().requires_grad_(True)
    u = A.double().view(N, C, 1, 1) * h1r + B.double().view(N, C, 1, 1)
    g1 = 0.5 * u * (1.0 + torch.erf(u / 2 ** 0.5))
    wd = w.double().view(C, 1, 3, 3).requires_grad_(True)
    out = F.conv2d(F.pad(g1, (1, 1, 1, 1), mode="reflect"), wd, groups=C)
    dh2 = c1.double().view(N, C, 1, 1) * rb(du2).double() + c2.double().view(N, C, 1, 1) * h2.float().double().cpu() \
        + c3.double().view(N, C, 1, 1)
    gh1, gw = torch.autograd.grad(out, (h1r, wd), dh2)
    within_one_rounding(f"dw_bwd[{H}x{W}]", du1, gh1 / A.double().view(N, C, 1, 1), extra=3e-6)
    dwd = torch.empty(C, 9, device=DEV)
    hb.call("uncr_dw_wgrad_reduce", dwp, N, C, sb, dwd, E._stream())
    e = float((dwd.cpu().double() - gw.view(C, 9)).abs().max() / gw.abs().max())
    print(f"[bf16] dw weight gradient: {e:.2e}")
    assert e < 1e-5
    st = part_sums(partb, sb).cpu()
    o64 = du1.float().double().cpu().view(N * C, -1)
    assert torch.allclose(st[:, 0], o64.sum(1), rtol=2e-5, atol=2e-3)
    assert torch.allclose(st[:, 1], (o64 * rb(h1).double().view(N * C, -1)).sum(1), rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("pro,epi", [(1, 1), (2, 1), (3, 3), (0, 1)])
def test_wide_gemm_bf16_storage(pro, epi):
    """The wide 1x1-conv GEMM with bf16 activations: out = W16 . rnd(f(in)) with the prologue in fp32, its result rounded to
    bf16, the weights kept to 16 significant bits (two bf16 parts), fp32 accumulation; epilogue variants."""
    from uncrtaints_amd import engine as E
    g = torch.Generator().manual_seed(2 + pro)
    N, P = 2, 2048
    Cin, Cout = (256, 128) if pro == 2 else (128, 256)
    x = torch.randn(N, Cin, P, generator=g)
    x2 = torch.randn(N, Cin, P, generator=g)
    W = torch.randn(Cout, Cin, generator=g) / Cin ** 0.5
    k0, k1, k2 = 0.5 + torch.rand(N * Cin, generator=g), 0.3 * torch.randn(N * Cin, generator=g), 0.5 + torch.rand(N * Cin, generator=g)
    xb, x2b = dev(x).to(BF), dev(x2).to(BF)
    xr, x2r = rb(x).double(), rb(x2).double()
    K0, K1, K2 = (t.double().view(N, Cin, 1) for t in (k0, k1, k2))
    if pro == 0:
        f = xr
    elif pro == 1:
        f = K0 * xr + K1
    elif pro == 2:
        u = K0 * xr + K1
        f = K2 * 0.5 * u * (1.0 + torch.erf(u / 2 ** 0.5))
    else:
        f = K0 * xr + K1 * x2r + K2
    f = f.float().to(BF).double()                       # the operand as the matrix pipe sees it
    Wt = E.pack_wt(dev(W), transpose=True)
    W16 = (W.view(torch.int32) & ~0xFFFF).view(torch.float32)
    W16 = W16 + ((W - W16).view(torch.int32) & ~0xFFFF).view(torch.float32)      # h + m of the exact split: 16 significant bits
    acc = torch.einsum("oc,ncp->nop", W16.double(), f)
    kw = dict(pro=pro, k=(dev(k0), dev(k1), dev(k2) if pro != 1 else None), x2=x2b if pro == 3 else None)
    if epi == 3:
        aux = torch.randn(N, Cout, P, generator=g)
        e0, e1, e2, e3 = (0.5 + torch.rand(N * Cout, generator=g) for _ in range(4))
        out, part = E.pw_gemm(xb, Wt, N, Cin, Cout, P, epi=3, aux=dev(aux).to(BF), ek=tuple(dev(t) for t in (e0, e1, e2, e3)), **kw)
        E0, E1, E2, E3 = (t.double().view(N, Cout, 1) for t in (e0, e1, e2, e3))
        auxr = rb(aux).double()
        u = E0 * auxr + E1
        gd = 0.5 * (1.0 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * np.pi) ** 0.5
        ref = gd * (E2 * acc + E3)
        second = auxr
    else:
        out, part = E.pw_gemm(xb, Wt, N, Cin, Cout, P, epi=1, **kw)
        ref, second = acc, None
    assert out.dtype == BF
    within_one_rounding(f"pw_gemm_a16[pro{pro},epi{epi}]", out, ref, extra=3e-6, flip_frac=5e-3, flip_extra=3e-3)
    st = part_sums(part.buf, part.slots).cpu()
    o64 = out.float().double().cpu().view(N * Cout, P)
    sec = o64 if second is None else second.view(N * Cout, P)
    assert torch.allclose(st[:, 0], o64.sum(1), rtol=2e-5, atol=2e-3) and torch.allclose(st[:, 1], (o64 * sec).sum(1), rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("shape", [(256, 128), (128, 256)])
def test_wide_weight_gradient_bf16_storage(shape):
    """dW = sum_p rnd(normbwd(d, d2)) * rnd(f(x)) from bf16 operands: one bf16 product per MAC, fp32 accumulation, fp64 reduce."""
    from uncrtaints_amd import engine as E
    g = torch.Generator().manual_seed(7)
    Cd, Cx = shape
    N, P = 2, 4096
    d, d2, x = torch.randn(N, Cd, P, generator=g), torch.randn(N, Cd, P, generator=g), torch.randn(N, Cx, P, generator=g)
    dk = [0.5 + torch.rand(N * Cd, generator=g), 0.2 * torch.randn(N * Cd, generator=g), 0.1 * torch.randn(N * Cd, generator=g)]
    xk = [0.5 + torch.rand(N * Cx, generator=g), 0.2 * torch.randn(N * Cx, generator=g)]
    pro_x = E.PRO_AFFINE if Cd == 256 else E.PRO_AFFINE_GELU
    dW, _ = E.pw_wgrad(dev(d).to(BF), dev(x).to(BF), N, Cd, Cx, P, pro_d=E.PRO_NORMBWD, dk=tuple(dev(t) for t in dk),
                       d2=dev(d2).to(BF), pro_x=pro_x, xk=(dev(xk[0]), dev(xk[1]), None), per_frame=True)
    fd = dk[0].double().view(N, Cd, 1) * rb(d).double() + dk[1].double().view(N, Cd, 1) * rb(d2).double() + dk[2].double().view(N, Cd, 1)
    u = xk[0].double().view(N, Cx, 1) * rb(x).double() + xk[1].double().view(N, Cx, 1)
    fx = u if pro_x == E.PRO_AFFINE else 0.5 * u * (1.0 + torch.erf(u / 2 ** 0.5))
    ref = torch.einsum("nop,ncp->noc", fd.float().to(BF).double(), fx.float().to(BF).double())
    e = float((dW.cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"[bf16] wgrad_a16{shape}: rel_err vs fp64 on the rounded operands {e:.2e}")
    assert e < 2e-5, e


def test_aggregate_and_pool_bf16_storage():
    from uncrtaints_amd import engine as E
    import uncrtaints_amd.hip_backend as hb
    g = torch.Generator().manual_seed(3)
    B, T, C, H, W, NH = 2, 3, 32, 64, 256, 4
    e = torch.randn(B, T, C, H, W, generator=g)
    att = torch.softmax(torch.randn(NH, B, T, 32, 32, generator=g), dim=2)
    eb = dev(e).to(BF)
    gq, sv, gpart = E.aggregate_forward(eb, dev(att), None, False, 0.0, 0, None, True)
    up = torch.nn.functional.interpolate(att.double().view(NH * B, T, 32, 32), size=(H, W), mode="bilinear", align_corners=False)
    up = up.view(NH, B, T, H, W)
    ref = torch.zeros(B, C, H, W, dtype=torch.float64)
    for h in range(NH):
        sl = slice(h * (C // NH), (h + 1) * (C // NH))
        ref[:, sl] = (up[h].unsqueeze(2) * rb(e).double()[:, :, sl]).sum(1)
    assert gq.dtype == BF
    within_one_rounding("aggregate_fwd", gq, ref, extra=2e-6)
    st = part_sums(gpart.buf, gpart.slots).cpu()
    assert torch.allclose(st[:, 0], gq.float().double().cpu().view(B * C, -1).sum(1), rtol=2e-5, atol=2e-3)
    dg = torch.randn(B, C, H, W, generator=g)
    de, datt = E.aggregate_backward(dev(dg).to(BF), sv)
    assert de.dtype == BF and datt.dtype == torch.float32
    ref_de = torch.zeros(B, T, C, H, W, dtype=torch.float64)
    for h in range(NH):
        sl = slice(h * (C // NH), (h + 1) * (C // NH))
        ref_de[:, :, sl] = up[h].unsqueeze(2) * rb(dg).double()[:, None, sl]
    within_one_rounding("aggregate_bwd_de", de, ref_de, extra=1e-6)
    # residual + 8x8 max-pool on bf16: the pooled value is the maximum of the STORED values, the scatter adds in place
    N, Cc = 2, 8
    x, h3 = torch.randn(N, Cc, H, W, generator=g), torch.randn(N, Cc, H, W, generator=g)
    A, Bc = 0.5 + torch.rand(N * Cc, generator=g), 0.1 * torch.randn(N * Cc, generator=g)
    y = torch.empty(N, Cc, H, W, device=DEV, dtype=BF)
    down = torch.empty(N, Cc, H // 8, 32, device=DEV)
    idx = torch.empty(N, Cc, H // 8, 32, device=DEV, dtype=torch.int32)
    hb.call("uncr_residual_pool", dev(x).to(BF), dev(h3).to(BF), dev(A), dev(Bc), y, None, down, idx, N * Cc, H, W, H // 8, 32, 1,
            E._stream())
    ref_y = rb(x).double() + A.double().view(N, Cc, 1, 1) * rb(h3).double() + Bc.double().view(N, Cc, 1, 1)
    within_one_rounding("residual_pool_y", y, ref_y, extra=1e-6)
    pooled = torch.nn.functional.max_pool2d(y.float().cpu(), 8)
    assert torch.equal(down.cpu(), pooled)
    dd = torch.randn(N, Cc, H // 8, 32, generator=g)
    de2 = torch.zeros(N, Cc, H, W, device=DEV, dtype=BF)
    base = torch.randn(N, Cc, H, W, generator=g)
    de2.copy_(dev(base).to(BF))
    E.maxpool_backward_into(dev(dd), idx, de2, H, W, H // 8, 32)
    ref = rb(base).clone().view(N * Cc, -1)
    ref.scatter_add_(1, idx.cpu().long().view(N * Cc, -1), dd.view(N * Cc, -1))
    assert torch.equal(de2.float().cpu().view(N * Cc, -1), rb(ref))


def _state(g, prefix="state/"):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _build(state, act, **widths):
    from uncrtaints_amd.src.backbones import uncrtaints as U
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0,
                     **widths)
    m.load_state_dict(state, strict=True)
    m.temporal_aggregator.attn_dropout.p = 0.0
    return m.to(DEV).set_act_dtype(act)


def _grad_report(tag, grads, ref, zero_ref):
    from gpu_util import is_zero_grad
    worst_l2, worst_cos = 0.0, 1.0
    for k, gv in grads.items():
        if is_zero_grad(k, zero_ref):
            continue        # mathematically-zero gradients: pure round-off on both sides
        a, b = gv.detach().cpu().double().flatten(), ref[k].double().flatten()
        assert torch.isfinite(a).all(), k
        l2 = float((a - b).norm() / b.norm())
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        if l2 > 0.05:
            print(f"[bf16] {tag} grad {k}: rel L2 {l2:.2e}, cos {cos:.5f}")
        worst_l2, worst_cos = max(worst_l2, l2), min(worst_cos, cos)
    print(f"[bf16] {tag}: worst gradient rel L2 {worst_l2:.2e}, cos {worst_cos:.5f}")
    return worst_l2, worst_cos


def _hip_bf16_step(state, x, y, dates, **widths):
    from uncrtaints_amd.src import losses
    m = _build(state, BF, **widths)
    m.train()
    xg = dev(x).requires_grad_(True)
    out = m(xg, batch_positions=dev(dates))
    assert out.dtype == torch.float32
    crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
    l, _ = crit(out[:, :, :13], dev(y), out[:, :, 13:26])
    l.backward()
    assert xg.grad is not None and xg.grad.dtype == torch.float32
    return out.detach().cpu(), l.item(), xg.grad.cpu(), {k: p.grad for k, p in m.named_parameters()}


@pytest.mark.parametrize("widths", [{}, dict(encoder_widths=[64], decoder_widths=[64, 64])], ids=["baseline", "w64"])
def test_model_bf16_vs_fp32_oracle(widths):
    """The model-level contract of the bf16 mode, on a seeded default-initialised network (B=2, T=3, 64x64), forward + MGNLL +
    backward, against (a) the fp32 CPU oracle -- the cost of bf16 storage -- and (b) the oracle with the SAME roundings emulated
    at the same tensors, forward and backward (OracleConfig.act_bf16; oracle.mbconv lists them).
    What is left in (b) is not a difference of rounding PLACES but of rounding TIES: an fp32-level difference (another summation
    order, 1e-6) ahead of a bf16 rounding tips it for about one element in two thousand, that element moves by 2^-8 of itself, and
    the perturbation travels on through 3x3 stencils and batch statistics.  The emulation measures its own sensitivity to exactly
    that -- the same emulated run with the weights perturbed by 1e-6 relative -- and the HIP path has to stay within 2 x that
    self-distance of the emulation (and inside absolute caps).  `w64`: the same contract away from the BASELINE widths (64-wide blocks:
    the narrow GEMM and weight-gradient kernels with bf16 storage on both sides)."""
    from gpu_util import oracle_run
    from oracle import uncrtaints_oracle as orc
    from uncrtaints_amd.src.backbones import uncrtaints as U
    torch.manual_seed(3)
    state = {k: v.clone() for k, v in U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus",
                                                  covmode="diag", scale_by=1.0, **widths).state_dict().items()}
    x, y, dates = orc.synthetic_batch(2, 3, 64, 64, seed=5)
    out_o, loss_o, dx_o, g_o, _ = oracle_run(state, x, y, dates, orc.OracleConfig(attn_dropout=0.0, **widths), torch.float32)
    _, _, _, g64, _ = oracle_run(state, x, y, dates, orc.OracleConfig(attn_dropout=0.0, **widths), torch.float64)
    emu = orc.OracleConfig(attn_dropout=0.0, act_bf16=True, **widths)
    out_e, loss_e, dx_e, g_e, _ = oracle_run(state, x, y, dates, emu, torch.float32)
    out, loss, dx, grads = _hip_bf16_step(state, x, y, dates, **widths)
    from gpu_util import is_zero_grad
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    self_out = self_dx = self_g = 0.0
    for seed in (9, 10, 11):        # the tie-flip distance is a random quantity: the largest of three draws
        gen = torch.Generator().manual_seed(seed)
        state_p = {k: (v * (1.0 + 1e-6 * torch.randn(v.shape, generator=gen)) if v.dtype.is_floating_point and "running" not in k else v.clone())
                   for k, v in state.items()}
        out_p, loss_p, dx_p, g_p, _ = oracle_run(state_p, x, y, dates, emu, torch.float32)
        self_out = max(self_out, float((out_p - out_e).abs().max() / out_e.abs().max()))
        self_dx = max(self_dx, l2(dx_p, dx_e))
        self_g = max(self_g, max(l2(g_p[k], g_e[k]) for k in g_e if not is_zero_grad(k, g64)))
    print(f"[bf16] emulation vs itself under a 1e-6 weight perturbation: out {self_out:.2e}, dx rel L2 {self_dx:.2e}, worst gradient rel L2 {self_g:.2e}")
    for tag, ro, rl, rdx, rg, t_out, t_loss, t_l2, t_cos in (
            ("vs fp32 oracle", out_o, loss_o, dx_o, g_o, 4e-2, 5e-3, 1.5e-1, 0.99),
            ("vs bf16-emulating oracle", out_e, loss_e, dx_e, g_e, 3e-2, 2e-3, 1.0e-1, 0.995)):
        e_out = float((out - ro).abs().max() / ro.abs().max())
        e_loss = abs(loss - rl.item()) / abs(rl.item())
        e_dx = l2(dx, rdx)
        print(f"[bf16] model {tag}: out rel_err {e_out:.2e}, loss rel_err {e_loss:.2e} ({loss:.5f} vs {rl.item():.5f}), dx rel L2 {e_dx:.2e}")
        wl2, cos = _grad_report(tag, grads, rg, g64)
        assert e_out <= t_out and e_loss <= t_loss, (tag, e_out, e_loss)
        assert wl2 <= t_l2 and cos >= t_cos and e_dx <= t_l2, (tag, wl2, cos, e_dx)
        if "emulating" in tag:      # no further from the emulation than two of its own tie-flip distances
            assert e_out <= 2 * self_out + 1e-3 and e_dx <= 2 * self_dx + 1e-3 and wl2 <= 2 * self_g + 1e-3, \
                (e_out, self_out, e_dx, self_dx, wl2, self_g)


def test_model_bf16_on_the_golden_fixture():
    """The golden fixture's weights (weight_init: N(0,1) BatchNorm weights, xavier convolutions; loss 631 dominated by pixels with
    variances at the 1e-8 clamp) amplify ANY 2^-9 perturbation: the bf16-emulating oracle itself is 5.8e-2 (outputs) / 9.6e-2
    (loss) away from the fp32 reference.  The HIP path is therefore held to the emulating oracle (forward), and only loosely to
    the fp32 reference values of the fixture."""
    from gpu_util import oracle_run
    from oracle import uncrtaints_oracle as orc
    g = load_golden("g1_diag_t3")
    state = _state(g)
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
    out_e, loss_e, _, _, _ = oracle_run(state, x, y, dates, orc.OracleConfig(attn_dropout=0.0, act_bf16=True), torch.float32)
    out, loss, _, grads = _hip_bf16_step(state, x, y, dates)
    ref, ref_loss = torch.from_numpy(g["train/out"]), float(g["train/loss"])
    e_emu = float((out - out_e).abs().max() / out_e.abs().max())
    e_ref = float((out - ref).abs().max() / ref.abs().max())
    print(f"[bf16] golden fixture: out vs emulating oracle {e_emu:.2e}, vs fp32 reference {e_ref:.2e}; loss {loss:.3f}, emulated "
          f"{loss_e.item():.3f}, fp32 reference {ref_loss:.3f}")
    # the emulation's own tie-flip distance on this fixture (weights perturbed by 1e-6, as in test_model_bf16_vs_fp32_oracle): the
    # loss of this fixture moves by several per cent under it, so the HIP path is held to the old caps OR two self-distances
    self_out = self_loss = 0.0
    for seed in (9, 10, 11):
        gen = torch.Generator().manual_seed(seed)
        state_p = {k: (v * (1.0 + 1e-6 * torch.randn(v.shape, generator=gen)) if v.dtype.is_floating_point and "running" not in k else v.clone())
                   for k, v in state.items()}
        out_p, loss_p, _, _, _ = oracle_run(state_p, x, y, dates, orc.OracleConfig(attn_dropout=0.0, act_bf16=True), torch.float32)
        self_out = max(self_out, float((out_p - out_e).abs().max() / out_e.abs().max()))
        self_loss = max(self_loss, abs(loss_p.item() - loss_e.item()) / abs(loss_e.item()))
    print(f"[bf16] golden fixture: emulation vs itself under a 1e-6 weight perturbation: out {self_out:.2e}, loss {self_loss:.2e}; "
          f"HIP vs emulation: loss {abs(loss - loss_e.item()) / abs(loss_e.item()):.2e}")
    # measured: out 2.2e-2 ... 2.4e-2, loss 0.7 ... 5.6 % (HIP 648.9, emulation 614.3, fp32 reference 631.3 with in_conv's moment path)
    assert e_emu <= max(4e-2, 2 * self_out) and abs(loss - loss_e.item()) <= max(5e-2, 2 * self_loss) * abs(loss_e.item())
    assert e_ref <= 1e-1 and abs(loss - ref_loss) <= 2e-1 * abs(ref_loss)
    assert all(torch.isfinite(v).all() for v in grads.values())


def test_model_bf16_eval_and_configs():
    """Eval mode (running statistics), T=6 iso (config 4 shapes at fixture size) and the padded-date path in bf16 storage vs the
    reference outputs of the fixtures."""
    from uncrtaints_amd.src.backbones import uncrtaints as U
    for name, cov, tol in (("g1_diag_t3", "diag", 3e-2), ("g1_diag_t3_pad", "diag", 3e-2), ("g1_iso_t6", "iso", 3e-2)):
        g = load_golden(name)
        state = _state(load_golden("g1_diag_t3") if name == "g1_diag_t3_pad" else g)
        m = U.UNCRTAINTS(input_dim=15, out_conv=[26 if cov == "diag" else 14], out_nonlin_mean=True, out_nonlin_var="softplus",
                         covmode=cov, scale_by=1.0)
        m.load_state_dict(state, strict=True)
        m = m.to(DEV).set_act_dtype("bf16").eval()
        with torch.no_grad():
            out = m(dev(torch.from_numpy(g["x"])), batch_positions=dev(torch.from_numpy(g["dates"])))
        ref = torch.from_numpy(g["eval/out"])
        e = float((out.cpu() - ref).abs().max() / ref.abs().max())
        print(f"[bf16] {name} eval out rel_err {e:.2e}")
        assert e <= tol


def test_train_sequence_bf16_tracks_fp32():
    """Three optimize_parameters steps (BaseModel, Adam) with bf16 activations follow the reference's fp32 loss sequence."""
    from types import SimpleNamespace
    from uncrtaints_amd.src.backbones.base_model import BaseModel
    g = load_golden("g6_trainseq")
    meta = json.loads(str(g["meta"]))
    cfg = SimpleNamespace(model="uncrtaints", use_sar=True, encoder_widths=[128], decoder_widths=[128] * 5, out_conv=[26],
                          mean_nonLinearity=True, var_nonLinearity="softplus", agg_mode="att_group", encoder_norm="group",
                          decoder_norm="batch", n_head=16, d_model=256, d_k=4, pad_value=0, padding_mode="reflect",
                          positional_encoding=True, covmode="diag", scale_by=meta["scale_by"], separate_out=False, use_v=False,
                          block_type="mbconv", pretrain=False, loss="MGNLL", lr=meta["lr"], gamma=0.8, device=DEV, chunk_size=None,
                          act_dtype="bf16")
    model = BaseModel(cfg).to(DEV)
    model.netG.load_state_dict(_state(g), strict=True)
    model.netG.temporal_aggregator.attn_dropout.p = 0.0
    assert model.netG.act_dtype == BF
    model.train()
    x, y, dates = (torch.from_numpy(g[k]) for k in ("x", "y", "dates"))
    ls = []
    for _ in range(3):
        model.set_input({"A": x, "B": y, "dates": dates, "masks": None})
        model.optimize_parameters()
        ls.append(model.loss_G.item())
    ref = [float(v) for v in g["losses"]]
    # the oracle with the same roundings emulated (Adam on the CPU) gives [1002.98, 142.89, 87.10] against the fp32 reference
    # [1041.00, 142.12, 97.24] (OracleConfig.act_bf16), the HIP path [973.7, 128.1, 87.0] (measured): the ill-conditioned
    # random-init loss moves by ~10 % under ANY realisation of bf16 storage, so both comparisons carry a 20 % band
    emu = [1002.9759521484375, 142.88760375976562, 87.09696960449219]
    print(f"[bf16] train sequence {ls} vs emulated {emu} vs fp32 reference {ref}")
    for a, e, b in zip(ls, emu, ref):
        assert abs(a - e) <= 2e-1 * abs(e) and abs(a - b) <= 2e-1 * abs(b), (ls, emu, ref)
    assert ls[2] < ls[1] < ls[0]


def test_bf16_mode_refuses_what_is_not_built():
    from uncrtaints_amd.src.backbones import uncrtaints as U
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], covmode="diag", out_nonlin_var="softplus", block_type="residual")
    with pytest.raises(NotImplementedError):
        m.set_act_dtype(torch.bfloat16)
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], covmode="diag", out_nonlin_var="softplus", use_v=True)
    with pytest.raises(NotImplementedError):
        m.set_act_dtype("bf16")
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], covmode="diag", out_nonlin_var="softplus", encoder_widths=[64],
                     decoder_widths=[64, 64])
    assert m.set_act_dtype("bf16").act_dtype == torch.bfloat16        # any width the fp32 path takes
    with pytest.raises(ValueError):
        m.set_act_dtype(torch.float16)
