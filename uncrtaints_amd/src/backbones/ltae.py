"""L-TAE classes on the `--model uncrtaints` path with the reference's surface
(model/src/backbones/ltae.py:145-239 LTAE2dtiny, :312-385 MultiHeadAttentionSmall,
:420-458 ScaledDotProductAttentionSmall).  Parameters live in stock nn modules under the reference's
attribute paths; compute runs in the HIP engine (csrc/ltae.hip + the MFMA pointwise GEMM)."""
import numpy as np
import torch
import torch.nn as nn

from ... import engine as E
from .positional_encoding import PositionalEncoder


_SDPA_CALLS = [0]      # host-side call counter of the stand-alone attention-dropout stream


class _SdpaRowsFn(torch.autograd.Function):
    """softmax(q.k / temperature) (+ dropout) (@ v) on pixel-major rows (csrc/attn_rows.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, pad, temperature, p_drop, want_out, want_comp):
        _SDPA_CALLS[0] += 1
        attn, out, comp, sv = E.sdpa_rows_forward(q, k, v, pad, temperature, p_drop, 0xA77E0000 + _SDPA_CALLS[0], want_out,
                                                  want_comp)
        ctx.sv = sv
        m, T = attn.shape
        empty = attn.new_empty(0)
        return attn.view(m, 1, T), (out.view(m, 1, -1) if out is not None else empty), (comp.view(m, 1, T) if comp is not None else empty)

    @staticmethod
    def backward(ctx, dattn, dout, dcomp):
        sv = ctx.sv
        has_out = dout is not None and dout.numel() > 0
        has_comp = dcomp is not None and dcomp.numel() > 0
        dq, dk, dv = E.sdpa_rows_backward(dattn, dout if has_out else None, dcomp if has_comp else None, sv,
                                          need_dv=ctx.needs_input_grad[2])
        return dq, dk, dv, None, None, None, None, None


def _pad_rows(pad_mask):
    return pad_mask.to(torch.int32).contiguous() if pad_mask is not None else None


class ScaledDotProductAttentionSmall(nn.Module):
    """ltae.py:420-458.  Inside LTAE2dtiny the score / softmax runs fused in the plane-tiled L-TAE kernels; a stand-alone call
    takes the reference's pixel-major rows: q [m, d_k], k [m, T, d_k], v [m, T, d_v], pad_mask [m, T] (bool)."""

    def __init__(self, temperature):
        super().__init__()
        self.temperature = temperature
        self.softmax = nn.Softmax(dim=2)

    def forward(self, q, k, v, pad_mask=None, return_comp=False, weight_v=False):
        attn, output, comp = _SdpaRowsFn.apply(q, k, v if weight_v else None, _pad_rows(pad_mask), float(self.temperature), 0.0,
                                               bool(weight_v), bool(return_comp))
        if weight_v:
            return (output, attn, comp) if return_comp else (output, attn)
        return attn


class _LinearRowsFn(torch.autograd.Function):
    """nn.Linear applied to [..., Din] rows on the MFMA pointwise GEMM (engine.linear_rows_forward)."""

    @staticmethod
    def forward(ctx, x, w, b):
        lead = x.shape[:-1]
        y, sv = E.linear_rows_forward(x.reshape(-1, x.shape[-1]), w, b)
        ctx.sv, ctx.w, ctx.lead, ctx.has_b = sv, w, lead, b is not None
        return y.reshape(*lead, w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        dx, dW, db = E.linear_rows_backward(dy.contiguous(), ctx.sv, ctx.w, ctx.needs_input_grad[0])
        if dx is not None:
            dx = dx.reshape(*ctx.lead, ctx.w.shape[1])
        return dx, dW, (db if ctx.has_b else None)


def _mha_rows(self, v, pad_mask, return_comp, weight_v, p_drop):
    """Common body of MultiHeadAttentionSmall.forward (ltae.py:341-385) and MultiHeadAttention.forward (:266-307)."""
    d_k, d_in, n_head = self.d_k, self.d_in, self.n_head
    if v.dim() != 3 or v.shape[-1] != d_in:
        raise ValueError(f"expected values [B*H*W, T, {d_in}], got {tuple(v.shape)}")
    if d_in > 256 or n_head * d_k > 256:
        raise NotImplementedError("the GEMM kernels are built for at most 256 channels")
    sz_b, seq_len, _ = v.size()
    q = self.Q                                                   # one query per head: shared by that head's sz_b rows
    k = _LinearRowsFn.apply(v, self.fc1_k.weight, self.fc1_k.bias).view(sz_b, seq_len, n_head, d_k)
    k = k.permute(2, 0, 1, 3).contiguous().view(-1, seq_len, d_k)           # (n_head * sz_b) x T x d_k
    pad = _pad_rows(pad_mask.repeat((n_head, 1))) if pad_mask is not None else None
    vv = None
    if weight_v:
        vv = torch.stack(v.split(v.shape[-1] // n_head, dim=-1)).view(n_head * sz_b, seq_len, -1)
    attn, output, comp = _SdpaRowsFn.apply(q, k, vv, pad, float(self.attention.temperature), p_drop, bool(weight_v),
                                           bool(return_comp))
    attn = attn.view(n_head, sz_b, 1, seq_len).squeeze(dim=2)
    if weight_v:
        output = output.view(n_head, sz_b, 1, d_in // n_head).squeeze(dim=2)
        return (output, attn, comp) if return_comp else (output, attn)
    return attn


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
        """v [B*H*W, T, d_in], pad_mask [B*H*W, T] -> attn [n_head, B*H*W, T] (with weight_v: (output [n_head, B*H*W,
        d_in / n_head], attn[, comp])).  Inside LTAE2dtiny this arithmetic runs fused; here on pixel-major rows."""
        return _mha_rows(self, v, pad_mask, return_comp, weight_v, 0.0)


def _ltae_params(m):
    if m.inconv is None:
        # d_model=None (ltae.py:49-54 / :177-182): no input projection.  The kernels take the projection as an operand, so the
        # identity (a constant, no gradient) stands in for it
        dev, C = m.in_norm.weight.device, m.in_channels
        eye = getattr(m, "_uncr_eye", None)
        if eye is None or eye[0].device != dev:
            eye = m._uncr_eye = (torch.eye(C, device=dev).view(C, C, 1), torch.zeros(C, device=dev))
        wi, bi = eye
    else:
        wi, bi = m.inconv.weight, m.inconv.bias
    return dict(in_norm_w=m.in_norm.weight, in_norm_b=m.in_norm.bias, inconv_w=wi,
                inconv_b=bi, fc_w=m.attention_heads.fc1_k.weight, fc_b=m.attention_heads.fc1_k.bias,
                Q=m.attention_heads.Q)


_LTAE_KEYS = ("in_norm_w", "in_norm_b", "inconv_w", "inconv_b", "fc_w", "fc_b", "Q")
# value branch of LTAE2d (use_v): Linear(d_model -> C), BatchNorm1d(C) [, further Linear + BatchNorm1d layers: keys with the layer's
# index appended], out GroupNorm(C)
_LTAEV_KEYS = ("mlp_w", "mlp_b", "bn_w", "bn_b", "on_w", "on_b")          # one layer: what UNCRTAINTS(use_v=True) builds


def _ltaev_keys(n_layers: int):
    ks = ["mlp_w", "mlp_b", "bn_w", "bn_b"]
    for i in range(1, n_layers):
        ks += [f"mlp_w{i}", f"mlp_b{i}", f"bn_w{i}", f"bn_b{i}"]
    return tuple(ks + ["on_w", "on_b"])


def _mlp_layers(m):
    """[(Linear, BatchNorm1d), ...] of the nn.Sequential(Linear, BatchNorm1d, ReLU, ...) the reference builds (ltae.py:75-84)"""
    return [(m.mlp[3 * i], m.mlp[3 * i + 1]) for i in range(len(m.mlp) // 3)]


def _ltae_value_params(m):
    out = {}
    for i, (lin, bn) in enumerate(_mlp_layers(m)):
        sfx = "" if i == 0 else str(i)
        out.update({"mlp_w" + sfx: lin.weight, "mlp_b" + sfx: lin.bias, "bn_w" + sfx: bn.weight, "bn_b" + sfx: bn.bias})
    out.update(on_w=m.out_norm.weight, on_b=m.out_norm.bias)
    return out


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
        if d_model is not None:
            self.d_model = d_model
            self.inconv = nn.Conv1d(in_channels, d_model, 1)
        else:                                  # ltae.py:177-182
            self.d_model = in_channels
            self.inconv = None
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
    """ltae.py:388-416 (temperature, attention dropout).  Inside LTAE2d the arithmetic runs in the L-TAE stage kernels; a
    stand-alone call takes pixel-major rows like ScaledDotProductAttentionSmall.  The dropout stream is this library's counter-based
    one (not torch's Philox): parity with the reference holds in eval mode / p = 0 and statistically in train mode."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(attn_dropout)
        self.softmax = nn.Softmax(dim=2)

    def forward(self, q, k, v, pad_mask=None, return_comp=False):
        p = float(self.dropout.p) if self.training else 0.0
        attn, output, comp = _SdpaRowsFn.apply(q, k, v, _pad_rows(pad_mask), float(self.temperature), p, True, bool(return_comp))
        return (output, attn, comp) if return_comp else (output, attn)


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
        """v [B*H*W, T, d_in] -> (output [n_head, B*H*W, d_in / n_head], attn [n_head, B*H*W, T][, comp])  (ltae.py:266-307)"""
        p = float(self.attention.dropout.p) if self.training else 0.0
        return _mha_rows(self, v, pad_mask, return_comp, True, p)


class _LTAE2dFn(torch.autograd.Function):
    """LTAE2d.forward (ltae.py:84-141) on its own: attention (engine.ltae_attention_forward), optional dropout on the low-resolution
    attention (use_dropout), the attention-weighted values + MLP + BatchNorm1d + ReLU + dropout + GroupNorm
    (engine.ltae_values_forward).  Inside UNCRTAINTS(use_v=True) the same pieces run inside the fused stage."""

    @staticmethod
    def forward(ctx, x, dates, pad, module, *params):
        nk = len(_LTAE_KEYS)
        layers = _mlp_layers(module)
        vkeys = _ltaev_keys(len(layers))
        p, vp = dict(zip(_LTAE_KEYS, params[:nk])), dict(zip(vkeys, params[nk:]))
        te = module
        denom = te.positional_encoder.denom_on(x.device) if te.positional_encoder is not None else None
        B, T, C, h, w = x.shape
        nh, dk = te.n_head, te.attention_heads.d_k
        att, sv_att = E.ltae_attention_forward(x.contiguous().float(), dates, pad, p, denom, nh, dk)
        _SDPA_CALLS[0] += 1
        seed = 0x17AE0000 + _SDPA_CALLS[0]
        pa = float(te.attention_heads.attention.dropout.p) if te.training else 0.0
        att_d = att
        if pa > 0.0:                            # MultiHeadAttention(use_dropout=True): dropout on the attention before the values
            att_d = torch.empty_like(att)
            E.hb.call("uncr_dropout", att, att_d, att.numel(), seed, None, pa, E._stream())
        v, sv_val = E.ltae_values_forward(dict(sv_att, att=att_d), pad, vp, nh, te.training,
                                          [(bn.running_mean, bn.running_var) for _, bn in layers], float(te.dropout.p), seed ^ 0x5bd1e995)
        if te.training:
            for _, bn in layers:
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
        if getattr(te, "keep_relu_branch", False):       # parity tools: the branch the last ReLU took (see _StageFn in uncrtaints.py)
            te._last_relu = (sv_val["m1"], sv_val["nf"].A, sv_val["nf"].B)
        ctx.sv = (sv_att, sv_val, p, vp, nh, dk, pa, seed)
        ctx.vkeys = vkeys
        return v.view(B, -1, h, w), att_d

    @staticmethod
    def backward(ctx, dv, datt):
        sv_att, sv_val, p, vp, nh, dk, pa, seed = ctx.sv
        B = dv.shape[0]
        dy1, datt_v, gv = E.ltae_values_backward(dv.contiguous().reshape(B, dv.shape[1], -1), sv_val, vp, nh)
        dsum = torch.empty_like(datt_v)
        E.hb.call("uncr_add", datt.contiguous(), datt_v, dsum, dsum.numel(), E._stream())
        if pa > 0.0:                            # the same mask (same seed) applied to the gradient
            dd = torch.empty_like(dsum)
            E.hb.call("uncr_dropout", dsum, dd, dsum.numel(), seed, None, pa, E._stream())
            dsum = dd
        ddown, g = E.ltae_attention_backward(dsum, sv_att, p, nh, dk, dy1_extra=dy1)
        g.update(gv)
        ddown = ddown.view(sv_att["down"].shape) if ctx.needs_input_grad[0] else None
        return (ddown, None, None, None) + tuple(g[k] for k in _LTAE_KEYS) + tuple(g[k] for k in ctx.vkeys)


class LTAE2d(nn.Module):
    """ltae.py:10-141: the attention of LTAE2dtiny plus the values (attention-weighted projected features -> (Linear +
    BatchNorm1d + ReLU) per mlp layer -> dropout -> GroupNorm).  UNCRTAINTS(use_v=True) runs it fused with the max-pool and the aggregation; a
    stand-alone call goes through _LTAE2dFn."""

    def __init__(self, in_channels=128, n_head=16, d_k=4, mlp=[256, 128], dropout=0.2, d_model=256, T=1000,
                 return_att=False, positional_encoding=True, use_dropout=True):
        super().__init__()
        import copy
        self.in_channels = in_channels
        mlp = copy.deepcopy(mlp)
        self.return_att = return_att
        self.n_head = n_head
        if len(mlp) < 2:
            raise ValueError("mlp needs at least [d_model, C]")
        if max(mlp) > 256:
            raise NotImplementedError("the GEMM kernels take at most 256 channels per operand: every mlp width <= 256")
        if d_model is not None:
            self.d_model = d_model
            self.inconv = nn.Conv1d(in_channels, d_model, 1)
        else:                                  # ltae.py:49-54
            self.d_model = in_channels
            self.inconv = None
        assert mlp[0] == self.d_model
        self.positional_encoder = PositionalEncoder(self.d_model // n_head, T=T, repeat=n_head) \
            if positional_encoding else None
        self.attention_heads = MultiHeadAttention(n_head=n_head, d_k=d_k, d_in=self.d_model, use_dropout=use_dropout)
        self.in_norm = nn.GroupNorm(num_groups=n_head, num_channels=self.in_channels)
        self.out_norm = nn.GroupNorm(num_groups=n_head, num_channels=mlp[-1])
        layers = []
        for i in range(len(mlp) - 1):                      # ltae.py:75-84
            layers += [nn.Linear(mlp[i], mlp[i + 1]), nn.BatchNorm1d(mlp[i + 1]), nn.ReLU()]
        self.mlp = nn.Sequential(*layers)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, batch_positions=None, pad_mask=None, return_comp=False):
        """x [B,T,C,h,w], batch_positions [B,T], pad_mask [B,T] bool -> out [B, mlp[-1], h, w] (and the attention
        [n_head,B,T,h,w] with return_att).  `return_comp` is accepted and unused, as in the reference (ltae.py:84,120)."""
        if self.positional_encoder is not None and batch_positions is None:
            raise ValueError("batch_positions (dates) are required when positional_encoding=True")
        pad = pad_mask.to(torch.int32).contiguous() if pad_mask is not None else None
        p, vp = _ltae_params(self), _ltae_value_params(self)
        out, attn = _LTAE2dFn.apply(x, batch_positions, pad, self,
                                    *([p[k] for k in _LTAE_KEYS] + [vp[k] for k in _ltaev_keys(len(_mlp_layers(self)))]))
        return (out, attn) if self.return_att else out
