"""L-TAE classes on the `--model uncrtaints` path with the reference's surface
(model/src/backbones/ltae.py:145-239 LTAE2dtiny, :312-385 MultiHeadAttentionSmall,
:420-458 ScaledDotProductAttentionSmall).  Parameters live in stock nn modules under the reference's
attribute paths; compute runs in the HIP engine (csrc/ltae.hip + the MFMA pointwise GEMM)."""
import numpy as np
import torch
import torch.nn as nn

from ... import engine as E
from .positional_encoding import PositionalEncoder


class ScaledDotProductAttentionSmall(nn.Module):
    """Holder for the temperature (ltae.py:420-458); the score/softmax runs fused in the HIP softmax kernel."""

    def __init__(self, temperature):
        super().__init__()
        self.temperature = temperature
        self.softmax = nn.Softmax(dim=2)

    def forward(self, q, k, v, pad_mask=None, return_comp=False, weight_v=False):
        raise NotImplementedError("ScaledDotProductAttentionSmall is fused into LTAE2dtiny on the HIP path; "
                                  "call LTAE2dtiny / MultiHeadAttentionSmall instead")


class MultiHeadAttentionSmall(nn.Module):
    """Parameters Q [n_head,d_k] and fc1_k Linear(d_in, n_head*d_k) (ltae.py:312-339)."""

    def __init__(self, n_head, d_k, d_in):
        super().__init__()
        self.n_head, self.d_k, self.d_in = n_head, d_k, d_in
        self.Q = nn.Parameter(torch.zeros((n_head, d_k))).requires_grad_(True)
        nn.init.normal_(self.Q, mean=0, std=np.sqrt(2.0 / (d_k)))
        self.fc1_k = nn.Linear(d_in, n_head * d_k)
        nn.init.normal_(self.fc1_k.weight, mean=0, std=np.sqrt(2.0 / (d_k)))
        self.attention = ScaledDotProductAttentionSmall(temperature=np.power(d_k, 0.5))

    def forward(self, v, pad_mask=None, return_comp=False, weight_v=False):
        raise NotImplementedError("MultiHeadAttentionSmall is fused into LTAE2dtiny on the HIP path (weight_v / "
                                  "return_comp variants are SURVEY 8(f) 'next')")


def _ltae_params(m):
    return dict(in_norm_w=m.in_norm.weight, in_norm_b=m.in_norm.bias, inconv_w=m.inconv.weight,
                inconv_b=m.inconv.bias, fc_w=m.attention_heads.fc1_k.weight, fc_b=m.attention_heads.fc1_k.bias,
                Q=m.attention_heads.Q)


_LTAE_KEYS = ("in_norm_w", "in_norm_b", "inconv_w", "inconv_b", "fc_w", "fc_b", "Q")
# value branch of LTAE2d (use_v): Linear(d_model -> C), BatchNorm1d(C), out GroupNorm(C)
_LTAEV_KEYS = ("mlp_w", "mlp_b", "bn_w", "bn_b", "on_w", "on_b")


def _ltae_value_params(m):
    return dict(mlp_w=m.mlp[0].weight, mlp_b=m.mlp[0].bias, bn_w=m.mlp[1].weight, bn_b=m.mlp[1].bias,
                on_w=m.out_norm.weight, on_b=m.out_norm.bias)


class _LTAEAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, down, dates, pad, module, *params):
        p = dict(zip(_LTAE_KEYS, params))
        denom = module.positional_encoder.denom_on(down.device) if module.positional_encoder is not None else None
        _, T, C, ah, aw = down.shape
        fwd = E.ltae_attention_forward_fused if E.ltae_fused_ok(T, C, module.n_head, ah * aw) else E.ltae_attention_forward
        att, sv = fwd(down.contiguous(), dates, pad, p, denom, module.n_head, module.attention_heads.d_k)
        ctx.sv, ctx.p, ctx.module = sv, p, module
        return att

    @staticmethod
    def backward(ctx, datt):
        m = ctx.module
        ddown, g = E.ltae_attention_backward(datt, ctx.sv, ctx.p, m.n_head, m.attention_heads.d_k)
        E.join_side()          # the parameter-gradient chain of the fused backward runs on the side stream
        B, T, C, S, _, _ = ctx.sv["dims"]
        ddown = ddown.view(ctx.sv["down"].shape) if ctx.needs_input_grad[0] else None
        return (ddown, None, None, None) + tuple(g[k] for k in _LTAE_KEYS)


class LTAE2dtiny(nn.Module):
    def __init__(self, in_channels=128, n_head=16, d_k=4, d_model=256, T=1000, positional_encoding=True):
        super().__init__()
        self.in_channels = in_channels
        self.n_head = n_head
        if d_model is None:
            raise NotImplementedError("LTAE2dtiny without the input projection (d_model=None) is not built")
        self.d_model = d_model
        self.inconv = nn.Conv1d(in_channels, d_model, 1)
        if positional_encoding:
            self.positional_encoder = PositionalEncoder(self.d_model // n_head, T=T, repeat=n_head)
        else:
            self.positional_encoder = None
        self.attention_heads = MultiHeadAttentionSmall(n_head=n_head, d_k=d_k, d_in=self.d_model)
        self.in_norm = nn.GroupNorm(num_groups=n_head, num_channels=self.in_channels)

    def forward(self, x, batch_positions=None, pad_mask=None):
        """x [B,T,C,h,w], batch_positions [B,T], pad_mask [B,T] bool -> attention [n_head,B,T,h,w]."""
        pad = pad_mask.to(torch.int32).contiguous() if pad_mask is not None else None
        p = _ltae_params(self)
        return _LTAEAttnFn.apply(x, batch_positions, pad, self, *[p[k] for k in _LTAE_KEYS])


class ScaledDotProductAttention(nn.Module):
    """Holder for ltae.py:388-416 (temperature, attention dropout); computed inside the fused L-TAE kernels."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)
        self.softmax = nn.Softmax(dim=2)


class MultiHeadAttention(nn.Module):
    """Parameters Q [n_head,d_k] and fc1_k Linear(d_in, n_head*d_k) (ltae.py:244-264); the attention-weighted values
    are computed by engine.ltae_values_forward."""

    def __init__(self, n_head, d_k, d_in, use_dropout=True):
        super().__init__()
        self.n_head, self.d_k, self.d_in = n_head, d_k, d_in
        self.Q = nn.Parameter(torch.zeros((n_head, d_k))).requires_grad_(True)
        nn.init.normal_(self.Q, mean=0, std=np.sqrt(2.0 / (d_k)))
        self.fc1_k = nn.Linear(d_in, n_head * d_k)
        nn.init.normal_(self.fc1_k.weight, mean=0, std=np.sqrt(2.0 / (d_k)))
        attn_dropout = 0.1 if use_dropout else 0.0
        self.attention = ScaledDotProductAttention(temperature=np.power(d_k, 0.5), attn_dropout=attn_dropout)

    def forward(self, v, pad_mask=None, return_comp=False):
        raise NotImplementedError("MultiHeadAttention is fused into LTAE2d on the HIP path")


class LTAE2d(nn.Module):
    """ltae.py:10-141 as built by UNCRTAINTS(use_v=True): same attention as LTAE2dtiny plus the values
    (attention-weighted projected features -> Linear + BatchNorm1d + ReLU -> dropout -> GroupNorm).  Parameter holder
    with the reference's attribute paths; UNCRTAINTS runs it fused with the max-pool and the aggregation."""

    def __init__(self, in_channels=128, n_head=16, d_k=4, mlp=[256, 128], dropout=0.2, d_model=256, T=1000,
                 return_att=False, positional_encoding=True, use_dropout=True):
        super().__init__()
        import copy
        self.in_channels = in_channels
        mlp = copy.deepcopy(mlp)
        self.return_att = return_att
        self.n_head = n_head
        if d_model is None:
            raise NotImplementedError("LTAE2d without the input projection (d_model=None) is not built")
        if use_dropout:
            raise NotImplementedError("dropout on the low-resolution attention (use_dropout=True) is not built; "
                                      "UNCRTAINTS(use_v=True) builds LTAE2d with use_dropout=False")
        if len(mlp) != 2:
            raise NotImplementedError("LTAE2d is built for a one-layer MLP (mlp=[d_model, C])")
        self.d_model = d_model
        self.inconv = nn.Conv1d(in_channels, d_model, 1)
        assert mlp[0] == self.d_model
        self.positional_encoder = PositionalEncoder(self.d_model // n_head, T=T, repeat=n_head) \
            if positional_encoding else None
        self.attention_heads = MultiHeadAttention(n_head=n_head, d_k=d_k, d_in=self.d_model, use_dropout=use_dropout)
        self.in_norm = nn.GroupNorm(num_groups=n_head, num_channels=self.in_channels)
        self.out_norm = nn.GroupNorm(num_groups=n_head, num_channels=mlp[-1])
        self.mlp = nn.Sequential(nn.Linear(mlp[0], mlp[1]), nn.BatchNorm1d(mlp[1]), nn.ReLU())
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, batch_positions=None, pad_mask=None, return_comp=False):
        raise NotImplementedError("LTAE2d runs fused inside UNCRTAINTS(use_v=True) on the HIP path")
