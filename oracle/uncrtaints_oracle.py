"""CPU oracle for the UnCRtainTS `--model uncrtaints` hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU fp32 restatement of the
reference algorithm (forward of the network + the MGNLL loss; backward through
torch.autograd).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it -- never the product path in `uncrtaints_amd/`.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference from
/root/reference in the build container and records inputs/outputs/gradients; the
fixtures are committed under `tests/golden/` and `tests/test_oracle_golden.py` checks
this restatement against them (the reference itself never travels to the GPU box).

Every function cites the reference file:line (relative to /root/reference) it follows.
Parameters are passed as a flat dict with the reference's `state_dict()` key names
(SURVEY.md section 8(b)), so a reference checkpoint can be fed in unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

S2_BANDS = 13
Tensor = torch.Tensor

# The reference's nn.Conv2d / nn.GroupNorm / nn.BatchNorm2d modules bottom out in these ATen ops; with
# USE_ATEN the oracle calls the same ops (this is also what makes it a fair CPU baseline).  With
# USE_ATEN = False the explicit formulas below are used instead; tests/test_oracle_golden.py checks that
# both agree.
USE_ATEN = True


@dataclass
class OracleConfig:
    """Constructor arguments of UNCRTAINTS (model/src/backbones/uncrtaints.py:231-254)."""
    input_dim: int = 15
    encoder_widths: List[int] = field(default_factory=lambda: [128])
    decoder_widths: List[int] = field(default_factory=lambda: [128] * 5)
    out_conv: List[int] = field(default_factory=lambda: [26])
    out_nonlin_mean: bool = True
    out_nonlin_var: str = "softplus"
    agg_mode: str = "att_group"
    encoder_norm: str = "group"
    decoder_norm: str = "batch"
    n_head: int = 16
    d_model: int = 256
    d_k: int = 4
    pad_value: float = 0.0
    positional_encoding: bool = True
    covmode: str = "diag"
    scale_by: float = 1.0
    T_period: int = 1000          # ltae.py:152
    att_down: int = 32            # uncrtaints.py:403 (hard-coded)
    attn_dropout: float = 0.1     # uncrtaints.py:154
    separate_out: bool = False    # uncrtaints.py:376-379
    is_mono: bool = False         # uncrtaints.py:322,418
    block_type: str = "mbconv"    # 'mbconv' | 'residual' (uncrtaints.py:24-69,315-319,349-353)
    use_v: bool = False           # uncrtaints.py:324-338,414-417 (LTAE2d values + include_v)
    ltae_dropout: float = 0.2     # ltae.py:17,97 (dropout on the MLP-processed values; use_v only)
    # NOT a reference argument: emulate the build's bf16 activation storage (BASELINE config 3).  Every tensor the HIP path
    # stores as bf16 (and every operand it rounds ahead of the matrix pipe) is rounded to bf16 here too, and so is its
    # gradient in backward (straight-through).  The reference itself is fp32 only; this variant exists to separate the cost
    # inherent in bf16 storage from implementation error in the bf16 parity tests.
    act_bf16: bool = False
    inconv_moments: bool = True     # bf16 emulation only: engine.dev_options(inconv_moments=...) of the path under test

    @property
    def covar_dim(self) -> int:   # uncrtaints.py:357-365
        return {"uni": S2_BANDS, "iso": 1, "diag": S2_BANDS}.get(self.covmode, 0)

    @property
    def mean_idx(self) -> int:    # uncrtaints.py:367
        return S2_BANDS

    @property
    def vars_idx(self) -> int:    # uncrtaints.py:368
        return S2_BANDS + self.covar_dim

    @property
    def eps(self) -> float:       # uncrtaints.py:374
        return 1e-9 if self.scale_by == 1.0 else 1e-3


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------

class _RoundBf16(torch.autograd.Function):
    """Round to bf16 and back in the forward (fwd) and / or the backward (bwd) direction: the value as the HIP path stores it,
    and the gradient as the HIP path stores IT -- the two do not always sit at the same tensor (see mbconv)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


class _ContiguousGrad(torch.autograd.Function):
    """Identity whose gradient is made contiguous, placed behind F.instance_norm.  ATen's CPU batch-norm backward (which serves
    instance_norm on the [1, N*C, H, W] view) mis-reads a gradient with expanded / permuted strides -- what the einsum of this file's
    'att_mean' aggregation (one head-averaged map for all channels) hands back through the residual connection for B = 1: the
    encoder's weight gradients then come out ORTHOGONAL to the true ones while the forward values and the gradient's values at the
    encoder output are right.  Found in round 6 (tools/debug_instance_pad.py: finite differences of the HIP forward and the reference's
    own modules agree with the HIP gradients; the reference's `x * attn[:, :, None]` form produces a dense gradient and is not
    affected).  Rounds 4-5 had read the symptom as a defect of the HIP path under `encoder_norm='instance'` with a padded date."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _store(x: Tensor, on: bool, grad: bool = True) -> Tensor:
    """A tensor the HIP path keeps in bf16 (or rounds ahead of the matrix pipe).  grad: its gradient is a bf16 tensor too (a stored
    activation gradient, or a gradient-GEMM operand)."""
    return _RoundBf16.apply(x, True, grad) if on else x


def _ground(x: Tensor, on: bool) -> Tensor:
    """A tensor that never exists in memory on the HIP path (it lives in a prologue) but whose GRADIENT is stored in bf16."""
    return _RoundBf16.apply(x, False, True) if on else x


def gelu_exact(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (uncrtaints.py:128,133,88)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def group_norm(x: Tensor, groups: int, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.GroupNorm over (C/groups channels x all trailing dims) per sample, biased variance.
    Used at utae.py:470-473 (n_groups=4), uncrtaints.py:16-22, ltae.py:191-194."""
    if USE_ATEN:
        return F.group_norm(x, groups, w, b, eps)
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mu = xg.mean(dim=-1, keepdim=True)
    var = xg.var(dim=-1, unbiased=False, keepdim=True)
    xn = ((xg - mu) * torch.rsqrt(var + eps)).reshape(x.shape)
    shape = [1, c] + [1] * (x.dim() - 2)
    return xn * w.view(shape) + b.view(shape)


def batch_norm(x: Tensor, w: Tensor, b: Tensor, running_mean: Tensor, running_var: Tensor,
               training: bool, momentum: float = 0.1, eps: float = 1e-5,
               update_running: bool = True) -> Tensor:
    """nn.BatchNorm2d (uncrtaints.py:17-18): batch statistics (biased var) in train mode, running
    statistics in eval mode; running buffers are updated in place with the unbiased variance."""
    c = x.shape[1]
    if USE_ATEN:
        return F.batch_norm(x, running_mean if (update_running or not training) else None,
                            running_var if (update_running or not training) else None, w, b, training, momentum, eps)
    if training:
        mu = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if update_running:
            with torch.no_grad():
                m = x.numel() / c
                running_mean.mul_(1 - momentum).add_(momentum * mu.detach())
                running_var.mul_(1 - momentum).add_(momentum * var.detach() * m / max(m - 1, 1))
    else:
        mu, var = running_mean, running_var
    xn = (x - mu.view(1, c, 1, 1)) * torch.rsqrt(var.view(1, c, 1, 1) + eps)
    return xn * w.view(1, c, 1, 1) + b.view(1, c, 1, 1)


def conv1x1(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """Conv2d(kernel 1) on [N,C,H,W]; w is [Cout,Cin,1,1] (utae.py:476-484, uncrtaints.py:126,136)."""
    if USE_ATEN:
        return F.conv2d(x, w.reshape(w.shape[0], w.shape[1], 1, 1), b)
    y = torch.einsum("oc,nchw->nohw", w.reshape(w.shape[0], w.shape[1]), x)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def depthwise3x3_reflect(x: Tensor, w: Tensor) -> Tensor:
    """Conv2d(C,C,3,padding=1,padding_mode='reflect',groups=C,bias=False) (uncrtaints.py:130-131).
    w is [C,1,3,3]; cross-correlation (no kernel flip)."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    if USE_ATEN:
        return F.conv2d(xp, w, groups=x.shape[1])
    h, wd = x.shape[-2:]
    out = torch.zeros_like(x)
    for i in range(3):
        for j in range(3):
            out = out + xp[:, :, i:i + h, j:j + wd] * w[:, 0, i, j].view(1, -1, 1, 1)
    return out


def squeeze_excite(x: Tensor, w1: Tensor, w2: Tensor) -> Tensor:
    """SE block (uncrtaints.py:82-97): global avg-pool -> Linear(no bias) -> GELU -> Linear(no bias)
    -> sigmoid -> channel scale."""
    y = x.mean(dim=(2, 3))
    y = gelu_exact(y @ w1.t())
    y = torch.sigmoid(y @ w2.t())
    return x * y[:, :, None, None]


class _NormCtx:
    """Selects GroupNorm(4) (encoder) or BatchNorm2d (decoder) for one MBConv (uncrtaints.py:16-22)."""

    def __init__(self, params: Dict[str, Tensor], kind: str, training: bool, update_running: bool):
        self.p, self.kind, self.training, self.update = params, kind, training, update_running

    def __call__(self, x: Tensor, prefix: str) -> Tensor:
        if self.kind == "instance":        # nn.InstanceNorm2d defaults: no affine, instance statistics in train and eval
            return _ContiguousGrad.apply(F.instance_norm(x, eps=1e-5))      # (a dense gradient for ATen's backward: see the class)
        w, b = self.p[prefix + ".weight"], self.p[prefix + ".bias"]
        if self.kind == "group":
            return group_norm(x, 4, w, b)
        if self.kind == "batch":
            return batch_norm(x, w, b, self.p[prefix + ".running_mean"], self.p[prefix + ".running_var"],
                              self.training, update_running=self.update)
        raise NotImplementedError(self.kind)


def conv3x3_reflect(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Conv2d(k=3, padding=1, padding_mode='reflect') (utae.py:478-487), dense."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def residual_block(x: Tensor, p: Dict[str, Tensor], prefix: str, norm: str, training: bool,
                   update_running: bool = True, relu_masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """ResidualConvBlock (uncrtaints.py:24-69): x + CL3(CL2(CL1(x))), every ConvLayer = conv3x3(reflect, bias) ->
    norm -> ReLU (the third one too: uncrtaints.py:53-62).
    relu_masks (test infrastructure, like `pool_idx` of forward): {f"{prefix}.conv{i}": 0/1 tensor} -- the branch of each ReLU
    that the implementation under test took; where given, the layer multiplies by the mask instead of taking max(u, 0), so that a
    pre-activation within rounding of zero is differentiated on the same side by every party (a ReLU is a kink like the max-pool)."""
    nrm = _NormCtx(p, norm, training, update_running)
    h = x
    for i in (1, 2, 3):
        pre = f"{prefix}.conv{i}.conv"
        u = nrm(conv3x3_reflect(h, p[pre + ".0.weight"], p[pre + ".0.bias"]), pre + ".1")
        mask = relu_masks.get(f"{prefix}.conv{i}") if relu_masks is not None else None
        h = u * mask.to(u.dtype) if mask is not None else torch.relu(u)
    return x + h


def _block(x, p, prefix, norm, training, update_running, taps, cfg, relu_masks=None):
    if cfg.block_type == "residual":
        return residual_block(x, p, prefix, norm, training, update_running, relu_masks)
    return mbconv(x, p, prefix, norm, training, update_running, taps, bf16=cfg.act_bf16)


def mbconv(x: Tensor, p: Dict[str, Tensor], prefix: str, norm: str, training: bool,
           update_running: bool = True, taps: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """MBConv(inp, oup, expansion=2) without down-sampling (uncrtaints.py:100-146):
    x + Norm(pw2(SE(GELU(Norm(dw3x3(GELU(Norm(pw1(PreNorm(x)))))))))).  `prefix` e.g. 'in_block.0'."""
    nrm = _NormCtx(p, norm, training, update_running)
    # bf16 = True: `_store` marks what the HIP path keeps in bf16 (h1, h2, h3, the block output) or rounds ahead of the
    # matrix pipe (the two GEMM operands a and z)
    # Where the HIP path rounds in bf16 mode -- forward: x, h1, h2, h3, the block output (stored) and the two GEMM
    # operands a, z (rounded ahead of the matrix pipe); backward: the stored gradients dx (block input, skip and PreNorm path summed
    # first), du1 and du2 (gradients of the two norm OUTPUTS u1, u2: written by the depthwise backward / the dz GEMM's epilogue) and
    # the gradient-GEMM operands dh1, dh3 (norm backward of (du1, h1) / (dy, h3), rounded ahead of the matrix pipe).  The gradients
    # of a, h2 and z exist in registers only (fp32) -- except that blocks outside 64 < C <= 128 run pw1's backward unfused on the HIP
    # path (uncr_pw_gemm_dx_supported): there the gradient of a is a stored tensor, rounded like the others.
    C, Ch = x.shape[1], p[prefix + ".conv.fn.0.weight"].shape[0]
    da_in_registers = 64 < C <= 128 and Ch <= 256 and C % 32 == 0 and Ch % 8 == 0
    a = _store(nrm(x, prefix + ".conv.norm"), bf16, grad=not da_in_registers)         # PreNorm (uncrtaints.py:72-79,140)
    h1 = _store(conv1x1(a, p[prefix + ".conv.fn.0.weight"]), bf16)      # pw 128->256
    g1 = gelu_exact(_ground(nrm(h1, prefix + ".conv.fn.1"), bf16))
    h2 = _store(depthwise3x3_reflect(g1, p[prefix + ".conv.fn.3.weight"]), bf16, grad=False)      # dw 3x3 reflect
    g2 = gelu_exact(_ground(nrm(h2, prefix + ".conv.fn.4"), bf16))
    z = _store(squeeze_excite(g2, p[prefix + ".conv.fn.6.fc.0.weight"], p[prefix + ".conv.fn.6.fc.2.weight"]), bf16, grad=False)
    h3 = _store(conv1x1(z, p[prefix + ".conv.fn.7.weight"]), bf16)      # pw-linear 256->128
    u3 = nrm(h3, prefix + ".conv.fn.8")
    if taps is not None:
        taps[prefix + ".h1"], taps[prefix + ".h2"], taps[prefix + ".h3"] = h1, h2, h3
    return _store(x + u3, bf16)


def positional_table(dates: Tensor, d: int, T: int, repeat: int) -> Tensor:
    """PositionalEncoder (positional_encoding.py:5-31): [B,T] dates -> [B,T,d*repeat].
    denom_i = T^(2*(i//2)/d); even i -> sin, odd i -> cos; tiled `repeat` times along channels."""
    i = torch.arange(d, dtype=torch.float32)
    denom = torch.pow(torch.tensor(float(T)), 2.0 * torch.div(i, 2, rounding_mode="floor") / d)
    tab = dates[:, :, None] / denom[None, None, :]
    even = (torch.arange(d) % 2 == 0)
    tab = torch.where(even[None, None, :], torch.sin(tab), torch.cos(tab))
    return tab.repeat(1, 1, repeat)


def ltae_tiny_attention(down: Tensor, dates: Tensor, pad_mask: Tensor, p: Dict[str, Tensor],
                        cfg: OracleConfig) -> Tensor:
    """LTAE2dtiny.forward + MultiHeadAttentionSmall + ScaledDotProductAttentionSmall
    (ltae.py:197-239, 341-385, 431-458).  down [B,T,C,h,w] -> attention [n_head,B,T,h,w]."""
    B, T, C, h, w = down.shape
    nh, dk = cfg.n_head, cfg.d_k
    x = down.permute(0, 3, 4, 2, 1).reshape(B * h * w, C, T)            # per low-res pixel: [C,T]
    x = group_norm(x, nh, p["temporal_encoder.in_norm.weight"], p["temporal_encoder.in_norm.bias"])
    wi = p["temporal_encoder.inconv.weight"][:, :, 0]                   # Conv1d k=1: [d_model,C]
    y = torch.einsum("oc,nct->nto", wi, x) + p["temporal_encoder.inconv.bias"]   # [n,T,d_model]
    if cfg.positional_encoding:
        pe = positional_table(dates, cfg.d_model // nh, cfg.T_period, nh)         # [B,T,d_model]
        y = y + pe[:, None, :, :].expand(B, h * w, T, cfg.d_model).reshape(B * h * w, T, cfg.d_model)
    k = y @ p["temporal_encoder.attention_heads.fc1_k.weight"].t() \
        + p["temporal_encoder.attention_heads.fc1_k.bias"]              # [n,T,nh*dk]
    k = k.view(B * h * w, T, nh, dk)
    q = p["temporal_encoder.attention_heads.Q"]                          # [nh,dk]
    score = torch.einsum("hd,nthd->hnt", q, k) / math.sqrt(dk)           # temperature = sqrt(d_k)
    pm = pad_mask[:, None, :].expand(B, h * w, T).reshape(B * h * w, T)
    score = score.masked_fill(pm[None], -1e3)                            # ltae.py:435
    attn = torch.softmax(score, dim=2)                                   # over T
    return attn.view(nh, B, h, w, T).permute(0, 1, 4, 2, 3)


def ltae2d_values_attention(down: Tensor, dates: Tensor, pad_mask: Tensor, p: Dict[str, Tensor], cfg: OracleConfig,
                            training: bool, update_running: bool = True,
                            v_dropout_mask: Optional[Tensor] = None, relu_mask: Optional[Tensor] = None,
                            taps: Optional[Dict[str, Tensor]] = None):
    """LTAE2d.forward + MultiHeadAttention + ScaledDotProductAttention (ltae.py:99-141, 266-307, 399-416), as built
    by UNCRTAINTS(use_v=True) (uncrtaints.py:324-336: mlp=[d_model, C], use_dropout=False, return_att=True).
    down [B,T,C,h,w] -> (values [B,C,h,w], attention [n_head,B,T,h,w]).
    relu_mask (test infrastructure, like `pool_idx` of forward): 0/1 tensor [B*h*w, C] -- the branch of the value MLP's (last) ReLU to
    differentiate.  The GroupNorm behind it normalises C/n_head = 8 values per pixel; where a group is dead (all eight negative) its
    rstd is 1/sqrt(eps) = 316, so ONE pre-activation within rounding of zero in such a group moves the gradients by ~316 / (n*C)."""
    B, T, C, h, w = down.shape
    nh, dk, n = cfg.n_head, cfg.d_k, B * h * w
    pre = "temporal_encoder."
    x = down.permute(0, 3, 4, 2, 1).reshape(n, C, T)
    x = group_norm(x, nh, p[pre + "in_norm.weight"], p[pre + "in_norm.bias"])
    if pre + "inconv.weight" in p:
        y = torch.einsum("oc,nct->nto", p[pre + "inconv.weight"][:, :, 0], x) + p[pre + "inconv.bias"]  # [n,T,d_model]
    else:                                                                  # LTAE2d(d_model=None): no input projection, ltae.py:49-54
        y = x.permute(0, 2, 1)
    if cfg.positional_encoding:
        pe = positional_table(dates, cfg.d_model // nh, cfg.T_period, nh)
        y = y + pe[:, None, :, :].expand(B, h * w, T, cfg.d_model).reshape(n, T, cfg.d_model)
    k = (y @ p[pre + "attention_heads.fc1_k.weight"].t() + p[pre + "attention_heads.fc1_k.bias"]).view(n, T, nh, dk)
    score = torch.einsum("hd,nthd->hnt", p[pre + "attention_heads.Q"], k) / math.sqrt(dk)
    pm = pad_mask[:, None, :].expand(B, h * w, T).reshape(n, T)
    attn = torch.softmax(score.masked_fill(pm[None], -1e3), dim=2)                    # [nh,n,T]; attn dropout p = 0
    dv = cfg.d_model // nh
    vh = y.view(n, T, nh, dv)                                                         # head h = channels h*dv..(h+1)*dv
    out = torch.einsum("hnt,nthd->nhd", attn, vh).reshape(n, cfg.d_model)             # heads concatenated
    vh_cat = out
    # the MLP: Linear + BatchNorm1d + ReLU per layer (ltae.py:75-84; UNCRTAINTS builds one layer, LTAE2d on its own any number);
    # BatchNorm1d over the n = B*h*w samples == BatchNorm2d on [n, C, 1, 1]
    li, m1 = 0, None
    while pre + f"mlp.{3 * li}.weight" in p:
        wk, bk = pre + f"mlp.{3 * li}", pre + f"mlp.{3 * li + 1}"
        last = pre + f"mlp.{3 * li + 3}.weight" not in p
        out = out @ p[wk + ".weight"].t() + p[wk + ".bias"]
        m1 = out
        Cv = out.shape[1]
        out = batch_norm(out.view(n, Cv, 1, 1), p[bk + ".weight"], p[bk + ".bias"], p.get(bk + ".running_mean"),
                         p.get(bk + ".running_var"), training, update_running=update_running).view(n, Cv)
        out = torch.relu(out) if (relu_mask is None or not last) else out * relu_mask.to(out.dtype)
        li += 1
    if taps is not None:        # (debugging: intermediate tensors of the value branch, tools/debug_spike.py)
        taps.update(val_y=y, val_vh=vh_cat, val_m1=m1, val_r=out)
    if training:
        if v_dropout_mask is not None:
            out = out * v_dropout_mask
        elif cfg.ltae_dropout > 0:
            out = F.dropout(out, cfg.ltae_dropout, training=True)
    out = group_norm(out, nh, p[pre + "out_norm.weight"], p[pre + "out_norm.bias"])    # per sample, C/nh channels per group
    v = out.view(B, h, w, out.shape[1]).permute(0, 3, 1, 2)
    return v, attn.view(nh, B, h, w, T).permute(0, 1, 4, 2, 3)


def temporal_aggregate(x: Tensor, pad_mask: Tensor, attn: Tensor, cfg: OracleConfig,
                       training: bool, dropout_mask: Optional[Tensor] = None) -> Tensor:
    """Compact_Temporal_Aggregator, mode 'att_group' (uncrtaints.py:156-221).
    x [B,T,C,H,W], attn [nh,B,T,h,w] -> [B,C,H,W].  `dropout_mask` (if given, shape
    [nh*B,T,H,W], values in {0, 1/(1-p)}) replaces the stochastic nn.Dropout in train mode."""
    nh, B, T, h, w = attn.shape
    H, W = x.shape[-2:]
    if cfg.agg_mode == "mean":              # uncrtaints.py:189-192, 220-221
        keep = (~pad_mask).float()
        out = (x * keep[:, :, None, None, None]).sum(dim=1)
        return out / keep.sum(dim=1)[:, None, None, None]
    if cfg.agg_mode == "att_mean":          # uncrtaints.py:179-188, 211-219: one head-averaged map for all channels
        attn = attn.mean(dim=0, keepdim=True)
        nh = 1
    a = attn.reshape(nh * B, T, h, w)
    if H > w or cfg.agg_mode == "att_mean":     # 'att_mean' always re-samples and applies dropout (uncrtaints.py:179-186, 211-217)
        a = F.interpolate(a, size=(H, W), mode="bilinear", align_corners=False)
        if training:
            if dropout_mask is not None:
                a = a * dropout_mask
            elif cfg.attn_dropout > 0:
                a = F.dropout(a, cfg.attn_dropout, training=True)
    else:
        a = F.avg_pool2d(a, kernel_size=w // H)
    a = a.view(nh, B, T, H, W)
    if bool(pad_mask.any()):
        a = a * (~pad_mask).float()[None, :, :, None, None]
    C = x.shape[2]
    xg = x.reshape(B, T, nh, C // nh, H, W)                               # channel c -> head c // (C/nh)
    out = torch.einsum("hbtyx,bthcyx->bhcyx", a, xg)
    return out.reshape(B, C, H, W)


def forward(p: Dict[str, Tensor], x: Tensor, dates: Tensor, cfg: OracleConfig, training: bool = False,
            dropout_mask: Optional[Tensor] = None, update_running: bool = True,
            taps: Optional[dict] = None, pool_idx: Optional[Tensor] = None,
            relu_masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """UNCRTAINTS.forward (uncrtaints.py:391-447).  x [B,T,Cin,H,W], dates [B,T] -> [B,1,13+covar,H,W].
    relu_masks (test infrastructure): see residual_block; key "temporal_encoder.mlp": see ltae2d_values_attention.
    pool_idx (test infrastructure, not a reference argument): flat in-plane arg-max indices [B*T, C, 32, 32] that the max-pool
    is to take instead of its own.  The max-pool is a kink of the function: where the two largest values of a window differ by
    less than the forward error, two correct fp32 evaluations may select different elements and route the pooled gradient to
    different pixels.  A parity test passes the indices the implementation under test selected (after checking that every
    selected value equals its window's maximum to forward accuracy), so that both sides differentiate the SAME branch."""
    B, T, Cin, H, W = x.shape
    pad_mask = (x == cfg.pad_value).all(dim=-1).all(dim=-1).all(dim=-1)   # [B,T]
    bf = cfg.act_bf16
    f = _store(x.reshape(B * T, Cin, H, W), bf, grad=False)               # smart_forward, utae.py:422-450 (input gradient: fp32)
    c0 = conv1x1(f, p["in_conv.conv.conv.0.weight"], p["in_conv.conv.conv.0.bias"])
    # bf16 emulation: behind a GroupNorm the HIP path never stores the pre-norm tensor of in_conv (statistics and parameter gradients
    # from the frames' second-moment matrices, csrc/inconv.hip: at most 15 input channels, 65 ... 256 output channels): no rounding of
    # c0 or of its gradient there; the other configurations keep it in bf16
    # (the same predicate as uncrtaints_amd.engine._inconv_moments_ok, limits of csrc/inconv.hip included: B*T frames <= 64, H*W % 4,
    # [B*T][Cout / 4] partial pairs of a group within 60 KB; cfg.inconv_moments mirrors the engine's development switch)
    Co, NF = c0.shape[1], B * T
    moments = (cfg.inconv_moments and cfg.encoder_norm == "group" and Cin + 1 <= 16 and 64 < Co <= 256 and Co % 4 == 0 and NF <= 64
               and (H * W) % 4 == 0 and (2 * NF * (Co // 4) + 4 * NF + 288) * 8 <= 60 * 1024)
    if not moments:
        c0 = _store(c0, bf)
    u0 = _NormCtx(p, cfg.encoder_norm, training, update_running)(c0, "in_conv.conv.conv.1")                            # utae.py:463-473
    if relu_masks is not None and relu_masks.get("in_conv") is not None:
        # test infrastructure, like pool_idx: the branch in_conv's ReLU took on the implementation under test ([N, C, H, W] of 0 / 1).
        # One pre-activation within rounding of zero decided differently moves in_conv's weight gradient by 1e-4 ... 6e-3 on small
        # images and under InstanceNorm (tools/fuzz_configs.py cases 247, 258, 331)
        a0 = _store(u0 * relu_masks["in_conv"].to(u0.dtype), bf)
    else:
        a0 = _store(torch.relu(u0), bf)
    e = a0
    for i in range(len(cfg.encoder_widths)):                              # one block per entry, uncrtaints.py:316-319, 399-400
        e = _block(e, p, f"in_block.{i}", cfg.encoder_norm, training, update_running, taps, cfg, relu_masks)
    C = e.shape[1]
    vals = None
    if cfg.is_mono:
        g, down, attn = e.view(B, T, C, H, W).squeeze(dim=1), None, None
    else:
        if pool_idx is not None:
            down = e.flatten(2).gather(2, pool_idx.to(torch.long).reshape(B * T, C, -1))
            down = down.view(B, T, C, cfg.att_down, cfg.att_down)
        else:
            down = F.adaptive_max_pool2d(e, (cfg.att_down, cfg.att_down)).view(B, T, C, cfg.att_down, cfg.att_down)
        if cfg.use_v:
            vals, attn = ltae2d_values_attention(down, dates, pad_mask, p, cfg, training, update_running,
                                                 relu_mask=relu_masks.get("temporal_encoder.mlp") if relu_masks else None, taps=taps)
        else:
            attn = ltae_tiny_attention(down, dates, pad_mask, p, cfg)
        g = _store(temporal_aggregate(e.view(B, T, C, H, W), pad_mask, attn, cfg, training, dropout_mask), bf)
        if cfg.use_v:                                                         # uncrtaints.py:414-417
            up_v = F.interpolate(vals, size=(H, W), mode="bilinear", align_corners=False)
            g = conv1x1(torch.cat((g, up_v), dim=1), p["include_v.weight"], p["include_v.bias"])
    if taps is not None:
        taps.update(c0=c0, a0=a0, e=e, down=down, attn=attn, agg=g)
        if vals is not None:
            taps["vals"] = vals
    out = g
    for i in range(len(cfg.decoder_widths)):
        out = _block(out, p, f"out_block.{i}", cfg.decoder_norm, training, update_running, taps, cfg, relu_masks)
        if taps is not None:
            taps[f"dec{i}"] = out
    if cfg.separate_out:
        o = conv1x1(out, p["out_conv_mean_1.conv.conv.0.weight"], p["out_conv_mean_1.conv.conv.0.bias"])
        if "out_conv_var_1.conv.conv.0.weight" in p:
            o = torch.cat((o, conv1x1(out, p["out_conv_var_1.conv.conv.0.weight"],
                                      p["out_conv_var_1.conv.conv.0.bias"])), dim=1)
        o = o.unsqueeze(1)
    else:
        # out_conv = ConvBlock(norm='none', last_relu=False) over [decoder_widths[0]] + out_conv (uncrtaints.py:381): Conv2d k=1 at the
        # even Sequential indices, a ReLU behind every convolution but the last (utae.py:476-494)
        i = 0
        while f"out_conv.conv.conv.{i + 2}.weight" in p:
            out = _store(torch.relu(conv1x1(out, p[f"out_conv.conv.conv.{i}.weight"], p[f"out_conv.conv.conv.{i}.bias"])), bf)
            i += 2
        o = conv1x1(out, p[f"out_conv.conv.conv.{i}.weight"], p[f"out_conv.conv.conv.{i}.bias"]).unsqueeze(1)
    o = _ground(o, bf)       # the gradient of the head's pre-activation is an activation gradient: stored like the decoder's
    if taps is not None:
        taps["pre_head"] = o
    mean = o[:, :, :cfg.mean_idx]
    if cfg.out_nonlin_mean:
        mean = cfg.scale_by * torch.sigmoid(mean)                          # uncrtaints.py:384
    if not cfg.covmode:
        return mean
    pre = o[:, :, cfg.mean_idx:cfg.vars_idx]
    if cfg.out_nonlin_var == "softplus":
        var = F.softplus(pre, beta=1, threshold=20) + cfg.eps          # uncrtaints.py:225
    elif cfg.out_nonlin_var == "elu":
        var = F.elu(pre) + 1 + cfg.eps                                  # uncrtaints.py:226
    else:
        var = pre                                                       # nn.Identity(), uncrtaints.py:227 (no eps)
    return torch.cat((mean, var), dim=2)


# --------------------------------------------------------------------------------------
# MGNLL (losses.py:131-218)
# --------------------------------------------------------------------------------------

def mgnll_per_pixel(pred: Tensor, target: Tensor, var: Tensor, mode: str = "diag",
                    eps: float = 1e-8) -> Tensor:
    """Un-reduced loss, laid out [W,H,B] like the reference's double vmap over the last two dims.

    L[x,y,b] = k/2 ln(2 pi) + 1/2 sum_{b',c} ln v[b',c,y,x] + 1/2 max(nan_to_num(maha[b,y,x]), 1e-9)
    with v = clamp(var, eps) applied without gradient effect (losses.py:203-205) and the log-det summed
    over the batch AND channels (losses.py:138 -- `var.log().sum()` inside the per-pixel function)."""
    if mode == "iso":
        var = var.expand(-1, -1, S2_BANDS, -1, -1)                         # losses.py:190-192
    if torch.any(var < 0):
        raise ValueError("var has negative entry/entries")                 # losses.py:199-200
    v = var + (var.detach().clamp(min=eps) - var.detach())                 # clamp, identity gradient
    pred, target, v = pred[:, 0], target[:, 0], v[:, 0]                    # T==1
    k = pred.shape[1]
    logdet = v.log().sum(dim=(0, 1))                                       # [H,W], over B and C
    maha = (((pred - target) ** 2) / v).sum(dim=1)                         # [B,H,W]
    maha = torch.nan_to_num(maha).clamp(min=1e-9)
    loss = 0.5 * k * math.log(2 * math.pi) + 0.5 * logdet[None] + 0.5 * maha
    loss = loss.permute(2, 1, 0)                                           # [W,H,B]
    return loss[..., 0] if loss.shape[-1] == 1 else loss                   # losses.py:141 .squeeze() drops B==1


def mgnll(pred: Tensor, target: Tensor, var: Tensor, mode: str = "diag", eps: float = 1e-8,
          reduction: str = "mean", want_covariance: bool = False):
    """multi_gaussian_nll_loss (losses.py:149-218) -> (loss, variance).  The dense covariance
    `diag_embed(v)` [B,1,13,13,H,W] is only materialised on request (it is used for logging only)."""
    if reduction not in ("none", "mean", "sum"):
        raise ValueError(reduction + " is not valid")
    loss = mgnll_per_pixel(pred, target, var, mode, eps)
    variance = None
    if want_covariance:
        v = var.expand(-1, -1, S2_BANDS, -1, -1) if mode == "iso" else var
        v = v.detach().clamp(min=eps)[:, 0]                                 # [B,13,H,W]
        variance = torch.diag_embed(v.permute(0, 2, 3, 1)).permute(0, 3, 4, 1, 2).unsqueeze(1)
    if reduction == "mean":
        return loss.mean(), variance
    if reduction == "sum":
        return loss.sum(), variance
    return loss, variance


def gnll(pred: Tensor, target: Tensor, var: Tensor, eps: float = 1e-8, full: bool = True, reduction: str = "mean"):
    """Element-wise Gaussian NLL (reference losses.py:46-128): var clamped to eps WITHOUT blocking its gradient
    (clone + in-place clamp under no_grad, losses.py:114-116); returns (loss, clamped var)."""
    if var.shape != pred.shape:
        if pred.shape[:-1] == var.shape:
            var = var.unsqueeze(-1)
        elif not (pred.shape[:-1] == var.shape[:-1] and var.shape[-1] == 1):
            raise ValueError("var is of incorrect size")
    if reduction not in ("none", "mean", "sum"):
        raise ValueError(reduction + " is not valid")
    if bool((var < 0).any()):
        raise ValueError("var has negative entry/entries")
    v = var + (var.clamp(min=eps) - var).detach()            # value clamped, identity gradient
    loss = 0.5 * (torch.log(v) + (pred - target) ** 2 / v)
    if full:
        loss = loss + 0.5 * math.log(2 * math.pi)
    if reduction == "mean":
        return loss.mean(), v
    if reduction == "sum":
        return loss.sum(), v
    return loss, v


def l1_loss(pred: Tensor, target: Tensor):
    """nn.L1Loss() of get_loss 'l1' (losses.py:21-23)."""
    return (pred - target).abs().mean()


def l2_loss(pred: Tensor, target: Tensor):
    """nn.MSELoss() of get_loss 'l2' (losses.py:24-26)."""
    return ((pred - target) ** 2).mean()


def loss_from_output(out: Tensor, target: Tensor, cfg: OracleConfig):
    """BaseModel.get_loss_G slicing (base_model.py:80-85)."""
    return mgnll(out[:, :, :cfg.mean_idx], target, out[:, :, cfg.mean_idx:cfg.vars_idx], mode=cfg.covmode)[0]


# --------------------------------------------------------------------------------------
# ensemble combine (ensemble_reconstruct.py:116-133)
# --------------------------------------------------------------------------------------

def ensemble_combine(means: Tensor, variances: Tensor, mode: str = "both"):
    """means [M,...], variances [M,...] -> (mean_ens, var_ens).
    'both': var = mean_i(var_i + mu_i^2) - mu_ens^2; 'aleatoric': mean_i var_i;
    'epistemic': mean_i mu_i^2 - mu_ens^2."""
    mu = means.mean(dim=0)
    if mode == "both":
        var = (variances + means ** 2).mean(dim=0) - mu ** 2
    elif mode == "aleatoric":
        var = variances.mean(dim=0)
    elif mode == "epistemic":
        var = (means ** 2).mean(dim=0) - mu ** 2
    else:
        raise ValueError(mode)
    return mu, var


# --------------------------------------------------------------------------------------
# parameter construction (reference-style init; weight_init.py:4-74, ltae.py:324-337)
# --------------------------------------------------------------------------------------

def init_params(cfg: OracleConfig, seed: int = 1) -> Dict[str, Tensor]:
    """Build a parameter dict with the reference's state_dict key names and shapes, initialised in the
    style of `netG.apply(weight_init)` (Conv2d/Linear xavier-normal + N(0,1) bias, Conv1d N(0,1),
    BatchNorm weight N(0,1) bias 0, GroupNorm default 1/0, Q ~ N(0, sqrt(2/d_k))).  Not bit-identical to
    the reference's RNG stream -- golden fixtures carry the actual reference weights."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}

    def xavier(*shape):
        fan_out, fan_in = shape[0], shape[1]
        rf = 1
        for s in shape[2:]:
            rf *= s
        std = math.sqrt(2.0 / ((fan_in + fan_out) * rf))
        return torch.randn(*shape, generator=g) * std

    def randn(*shape):
        return torch.randn(*shape, generator=g)

    def norm_params(prefix, c, kind):
        if kind == "batch":
            p[prefix + ".weight"], p[prefix + ".bias"] = randn(c), torch.zeros(c)
            p[prefix + ".running_mean"], p[prefix + ".running_var"] = torch.zeros(c), torch.ones(c)
            p[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        elif kind != "instance":
            p[prefix + ".weight"], p[prefix + ".bias"] = torch.ones(c), torch.zeros(c)

    def mb(prefix, c, kind):
        hd = 2 * c
        norm_params(prefix + ".conv.norm", c, kind)
        p[prefix + ".conv.fn.0.weight"] = xavier(hd, c, 1, 1)
        norm_params(prefix + ".conv.fn.1", hd, kind)
        p[prefix + ".conv.fn.3.weight"] = xavier(hd, 1, 3, 3)
        norm_params(prefix + ".conv.fn.4", hd, kind)
        p[prefix + ".conv.fn.6.fc.0.weight"] = xavier(int(c * 0.25), hd)
        p[prefix + ".conv.fn.6.fc.2.weight"] = xavier(hd, int(c * 0.25))
        p[prefix + ".conv.fn.7.weight"] = xavier(c, hd, 1, 1)
        norm_params(prefix + ".conv.fn.8", c, kind)

    if cfg.block_type == "residual":
        def mb(prefix, c, kind):        # noqa: F811 -- ResidualConvBlock parameters instead of MBConv's
            for i in (1, 2, 3):
                p[f"{prefix}.conv{i}.conv.0.weight"], p[f"{prefix}.conv{i}.conv.0.bias"] = xavier(c, c, 3, 3), randn(c)
                norm_params(f"{prefix}.conv{i}.conv.1", c, kind)

    c = cfg.encoder_widths[0]
    p["in_conv.conv.conv.0.weight"], p["in_conv.conv.conv.0.bias"] = xavier(c, cfg.input_dim, 1, 1), randn(c)
    norm_params("in_conv.conv.conv.1", c, cfg.encoder_norm)
    for i, ci in enumerate(cfg.encoder_widths):
        mb(f"in_block.{i}", ci, cfg.encoder_norm)
    p["temporal_encoder.inconv.weight"], p["temporal_encoder.inconv.bias"] = randn(cfg.d_model, c, 1), randn(cfg.d_model)
    p["temporal_encoder.attention_heads.Q"] = randn(cfg.n_head, cfg.d_k) * math.sqrt(2.0 / cfg.d_k)
    p["temporal_encoder.attention_heads.fc1_k.weight"] = xavier(cfg.n_head * cfg.d_k, cfg.d_model)
    p["temporal_encoder.attention_heads.fc1_k.bias"] = randn(cfg.n_head * cfg.d_k)
    p["temporal_encoder.in_norm.weight"], p["temporal_encoder.in_norm.bias"] = torch.ones(c), torch.zeros(c)
    if cfg.use_v:
        p["temporal_encoder.mlp.0.weight"], p["temporal_encoder.mlp.0.bias"] = xavier(c, cfg.d_model), randn(c)
        p["temporal_encoder.mlp.1.weight"], p["temporal_encoder.mlp.1.bias"] = randn(c), torch.zeros(c)
        p["temporal_encoder.mlp.1.running_mean"], p["temporal_encoder.mlp.1.running_var"] = torch.zeros(c), torch.ones(c)
        p["temporal_encoder.mlp.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        p["temporal_encoder.out_norm.weight"], p["temporal_encoder.out_norm.bias"] = torch.ones(c), torch.zeros(c)
        p["include_v.weight"], p["include_v.bias"] = xavier(c, 2 * c, 1, 1), randn(c)
    for i, cw in enumerate(cfg.decoder_widths):
        mb(f"out_block.{i}", cw, cfg.decoder_norm)
    oc = cfg.out_conv[-1]
    p["out_conv.conv.conv.0.weight"], p["out_conv.conv.conv.0.bias"] = xavier(oc, cfg.decoder_widths[0], 1, 1), randn(oc)
    return p


def synthetic_batch(B: int, T: int, H: int, W: int, seed: int = 1, input_dim: int = 15):
    """Synthetic inputs of SURVEY.md section 8(d): x~U[0,1) [B,T,15,H,W], y~U[0,1) [B,1,13,H,W],
    dates = sorted randint(1400,1800) as float (data/dataLoader.py:36-59 value ranges)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, T, input_dim, H, W, generator=g)
    y = torch.rand(B, 1, S2_BANDS, H, W, generator=g)
    dates = torch.sort(torch.randint(1400, 1800, (B, T), generator=g), dim=1).values.float()
    return x, y, dates


# ---- input assembly in front of the path ----
def process_ms(img, method: str = "default"):
    """data/dataLoader.py:38-48 (numpy): clip to [0, 10000]; 'default' -> /10000, 'resnet' -> /2000; nan_to_num."""
    import numpy as np
    img = np.asarray(img, dtype=np.float32).copy()
    if method in ("default", "resnet"):
        img = np.clip(img, 0, 10000)
        img = img / (10000 if method == "default" else 2000)
    return np.nan_to_num(img)


def process_sar(img, method: str = "default"):
    """data/dataLoader.py:50-61 (numpy): 'default' clip to [-25, 0] dB -> [0, 1]; 'resnet' per-channel ranges
    [-25, 0] / [-32.5, 0] -> [0, 2]; nan_to_num."""
    import numpy as np
    img = np.asarray(img, dtype=np.float32).copy()
    if method == "default":
        img = (np.clip(img, -25, 0) + 25) / 25
    elif method == "resnet":
        lo = (-25.0, -32.5)
        img = np.stack([2 * (np.clip(img[c], lo[c], 0) - lo[c]) / (0 - lo[c]) for c in range(2)], axis=0)
    return np.nan_to_num(img)


def prepare_data_multi(batch, use_sar: bool, batch_size: int):
    """model/train_reconstruct.py:161-179: lists over dates of [B,C,H,W] -> x [B,T,(2+)13,H,W] (S1 channels first),
    y [B,1,13,H,W], masks [B,T,H,W], dates [B,T] (mean of the S1 and S2 day offsets when SAR is used)."""
    s2, s2_td = batch["input"]["S2"], batch["input"]["S2 TD"]
    if batch_size > 1:
        s2_td = torch.stack(list(s2_td)).T
    m = torch.stack(list(batch["input"]["masks"])).swapaxes(0, 1)
    y = torch.cat(list(batch["target"]["S2"]), dim=0).unsqueeze(1)
    if use_sar:
        s1, s1_td = batch["input"]["S1"], batch["input"]["S1 TD"]
        if batch_size > 1:
            s1_td = torch.stack(list(s1_td)).T
        x = torch.cat((torch.stack(list(s1), dim=1), torch.stack(list(s2), dim=1)), dim=2)
        dates = torch.stack((torch.as_tensor(s1_td), torch.as_tensor(s2_td))).float().mean(dim=0)
    else:
        x = torch.stack(list(s2), dim=1)
        dates = torch.as_tensor(s2_td).float()
    return x, y, m, dates


# ---- evaluation metrics behind the path ----
def ssim_window(window_size: int = 11, sigma: float = 1.5) -> Tensor:
    """util/pytorch_ssim/__init__.py:7-15: normalised 1-D Gaussian, outer product (fp32)."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float()


def ssim(img1: Tensor, img2: Tensor, window_size: int = 11, size_average: bool = True):
    """util/pytorch_ssim/__init__.py:17-39,65-73: depthwise Gaussian filtering with zero padding."""
    C = img1.shape[1]
    w = ssim_window(window_size).to(img1.dtype).expand(C, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    f = lambda x: F.conv2d(x, w, padding=pad, groups=C)
    mu1, mu2 = f(img1), f(img2)
    s11, s22, s12 = f(img1 * img1) - mu1 * mu1, f(img2 * img2) - mu2 * mu2, f(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def img_metrics(target: Tensor, pred: Tensor, var: Optional[Tensor] = None, pixelwise: bool = True) -> dict:
    """model/src/learning/metrics.py:20-63"""
    rmse = torch.sqrt(torch.mean(torch.square(target - pred)))
    out = {"RMSE": rmse.item(), "MAE": torch.mean(torch.abs(target - pred)).item(),
           "PSNR": (20 * torch.log10(1 / rmse)).item()}
    mat = torch.sum(target * pred, 1)
    mat = mat / torch.sqrt(torch.sum(target * target, 1))
    mat = mat / torch.sqrt(torch.sum(pred * pred, 1))
    out["SAM"] = torch.mean(torch.acos(torch.clamp(mat, -1, 1)) * 180 / math.pi).item()
    out["SSIM"] = ssim(target, pred).item()
    if var is not None:
        e = target - pred
        se, ae = e * e, e.abs()
        out.update({"error": e.nanmean().item(), "mean ae": ae.nanmean().item(), "mean se": se.nanmean().item(),
                    "mean var": var.nanmean().item()})
        if pixelwise:
            pw = lambda x: x.nanmean(0).nanmean(0).flatten().numpy()
            out.update({"pixelwise error": pw(e), "pixelwise ae": pw(ae), "pixelwise se": pw(se), "pixelwise var": pw(var)})
    return out


# ------------------------------------------------------------------------------------------------
# calibration of the predicted variance (evaluation loop, after the path)
# ------------------------------------------------------------------------------------------------

def compute_ece(variances, errors, n_samples: int, percent: int = 5) -> np.ndarray:
    """model/train_reconstruct.py:475-487: rank the per-sample errors by ascending per-sample uncertainty; entry i is
    the nan-mean error of the most certain (i+1)*percent % samples."""
    order = torch.sort(torch.tensor(variances, dtype=torch.float32))[1]
    e = torch.tensor(errors, dtype=torch.float32)[order]
    ends = torch.linspace(0, n_samples, 100 // percent + 1, dtype=int)[1:]
    return np.array([torch.nanmean(e[:int(r)]).item() for r in ends], dtype=np.float32)


def compute_uce_auce(variances, errors, n_samples: int, percent: int = 5, l2: bool = True):
    """model/train_reconstruct.py:492-530 (numerical part): equal-width variance bins between min and max variance;
    per bin |metric(error) - metric(sqrt(var))| with metric = root-mean-square (l2) or mean-abs (l1); UCE is weighted
    by the bin population / n_samples, AUCE is the nan-mean over bins."""
    n_bins = 100 // percent
    v = torch.tensor(variances, dtype=torch.float32)
    e = torch.tensor(errors, dtype=torch.float32)
    edges = np.linspace(v.min().item(), v.max().item(), num=n_bins)[1:]
    which = torch.from_numpy(np.digitize(v.numpy(), bins=edges))
    calib = torch.full((n_bins,), float("nan"))
    for b in range(n_bins):
        sel = which == b
        if l2:
            bv, be = v[sel].sqrt().square().mean().sqrt(), e[sel].square().mean().sqrt()
        else:
            bv, be = v[sel].sqrt().abs().mean(), e[sel].abs().mean()
        calib[b] = (be - bv).abs()
    weight = torch.histogram(which.float(), n_bins)[0] / n_samples
    return torch.nansum(weight * calib).item(), torch.nanmean(calib).item()
