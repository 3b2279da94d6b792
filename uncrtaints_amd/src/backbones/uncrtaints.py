"""UnCRtainTS network with the reference's class surface (model/src/backbones/uncrtaints.py), computed by
hand-written HIP kernels for MI355X.

Same constructor arguments, tensor shapes, attributes (`mean_idx`, `vars_idx`, `variance`) and `state_dict()`
key names as the reference, so it drops into a train_reconstruct.py / test_reconstruct.py style driver and
loads reference checkpoints.  Modules hold parameters in stock `nn.*` layers (weight_init / freeze_layers
dispatch on those types); `forward` routes through autograd Functions over `uncrtaints_amd.engine`.
There is no PyTorch-op fallback: on a machine without the HIP library or a GPU tensor, forward raises.

Built: block_type in {mbconv, residual}, agg_mode in {att_group, att_mean, mean}, encoder_norm/decoder_norm in {group, batch,
instance},
use_v in {False, True}, separate_out, is_mono, out_nonlin_var in {softplus, elu, identity}, covmode in {diag, iso, uni, None}.
(SURVEY 8(a17) / 8(f))."""
import torch
import torch.nn as nn

from ... import engine as E
from .ltae import LTAE2d, LTAE2dtiny, _LTAE_KEYS, _LTAEV_KEYS, _ltae_params, _ltae_value_params
from .utae import ConvBlock, ConvLayer, TemporallySharedBlock, _ConvBiasReluFn

S2_BANDS = 13


def get_norm_layer(out_channels, num_feats, n_groups=4, layer_type='batch'):
    if layer_type == 'batch':
        return nn.BatchNorm2d(out_channels)
    elif layer_type == 'instance':
        return nn.InstanceNorm2d(out_channels)
    elif layer_type == 'group':
        return nn.GroupNorm(num_channels=num_feats, num_groups=n_groups)


class _NormFn(torch.autograd.Function):
    """A norm layer on its own (one statistics pass + one affine pass); inside MBConv it is a prologue of the next kernel."""

    @staticmethod
    def forward(ctx, x, module, gamma, beta):
        kind = "batch" if isinstance(module, nn.BatchNorm2d) else ("instance" if isinstance(module, nn.InstanceNorm2d) else "group")
        spec = E.NormSpec(kind, module.num_groups if kind == "group" else 4)
        rm, rv = (module.running_mean, module.running_var) if kind == "batch" else (None, None)
        mom = 0.1
        if kind == "batch":
            # momentum=None: cumulative moving average, factor 1 / num_batches_tracked counted INCLUDING this batch (torch
            # increments the counter before it computes the factor); apply_norm has incremented it already
            if module.momentum is not None:
                mom = module.momentum
            elif module.training and module.num_batches_tracked is not None:
                mom = 1.0 / max(float(module.num_batches_tracked.item()), 1.0)
        out, sv = E.norm_apply_forward(x, spec, module.training, gamma, beta, rm, rv, mom, module.eps)
        ctx.sv, ctx.gamma = sv, gamma
        return out

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db = E.norm_apply_backward(dy, ctx.sv, ctx.gamma, ctx.needs_input_grad[0])
        return dx, None, (dg if ctx.gamma is not None else None), (db if ctx.gamma is not None else None)


def apply_norm(module, x):
    """`module(x)` for nn.GroupNorm / nn.BatchNorm2d / nn.InstanceNorm2d holders on the HIP path (x [N,C,H,W])."""
    if isinstance(module, nn.BatchNorm2d) and module.training and module.num_batches_tracked is not None:
        module.num_batches_tracked += 1
    return _NormFn.apply(x, module, getattr(module, "weight", None), getattr(module, "bias", None))


class PreNorm(nn.Module):
    """`fn(norm(x))` (uncrtaints.py:72-79).  MBConv evaluates its PreNorm fused (the norm is the prologue of pw1); a stand-alone
    call runs the norm as its own HIP passes and hands the result to `fn`."""

    def __init__(self, dim, fn, norm, n_groups=4):
        super().__init__()
        self.norm = get_norm_layer(dim, dim, n_groups, norm)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(apply_norm(self.norm, x), **kwargs)


class _SEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2):
        out, sv = E.se_forward(x, w1, w2)
        ctx.sv, ctx.w = sv, (w1, w2)
        return out

    @staticmethod
    def backward(ctx, dy):
        dx, dw1, dw2 = E.se_backward(dy, ctx.sv, ctx.w[0], ctx.w[1], ctx.needs_input_grad[0])
        return dx, dw1, dw2


class SE(nn.Module):
    """Squeeze-excite (uncrtaints.py:82-97): x * sigmoid(fc2(gelu(fc1(avgpool(x))))).  Fused into its neighbours inside MBConv;
    a stand-alone call runs pooling statistics + the SE-MLP kernel + one scaling pass."""

    def __init__(self, inp, oup, expansion=0.25):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(
            nn.Linear(oup, int(inp * expansion), bias=False),
            nn.GELU(),
            nn.Linear(int(inp * expansion), oup, bias=False),
            nn.Sigmoid()
        )

    def forward(self, x):
        return _SEFn.apply(x, self.fc[0].weight, self.fc[2].weight)


class _MBConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, inference, *params):
        # inference: the caller ran with autograd off (torch.no_grad(): validation / test loops).  It cannot be read here -- grad mode
        # is always off inside Function.forward, and needs_input_grad reports the parameters' requires_grad whatever the mode
        p = dict(zip(E.MB_KEYS, params))
        x = x.contiguous()
        y, sv, party = E.mbconv_forward(x, p, module._spec, module.training, getattr(x, "_uncr_part", None),
                                        module._bn_buffers(), want_out_stats=True,
                                        x_h3=getattr(x, "_uncr_h3", None), pool=getattr(x, "_uncr_pool", None),
                                        inference=bool(inference))
        sv["x_relu"] = getattr(x, "_uncr_relu", None)     # x is in_conv's relu(norm(c0)): (c0, A, B)
        ctx.sv, ctx.p = sv, p
        ctx.versions = tuple(None if t is None else t._version for t in params)
        y._uncr_part = party        # (sum y, sum y^2) partials for the next PreNorm
        y._uncr_pooled = sv.pop("ypool")   # (max-pooled y, argmax) when the L-TAE stage asked for it
        y._uncr_h3 = sv["h3"]       # lets the consumer of y emit this block's norm-3 backward statistics
        return y

    @staticmethod
    def backward(ctx, dy):
        # the backward re-packs / re-reads the live parameters: like autograd's saved-tensor check, refuse to differentiate
        # through weights that were modified in place since the forward (forward -> optimizer.step() -> backward)
        for t, v, k in zip(ctx.p.values(), ctx.versions, E.MB_KEYS):
            if t is not None and t._version != v:
                raise RuntimeError(f"MBConv parameter '{k}' was modified in place between forward and backward")
        # when dy comes straight from the consumer's backward kernel it carries (sum dy, sum dy*h3) partials; they are
        # honoured only while dy is still the very tensor (address, version) they were computed from
        part = E.claim_part(dy)
        N, C, _, _, H, W = ctx.sv["dims"]
        if part is not None and (part.buf.shape[0] != N * C or part.masked):
            part = None
        dx, g, dx_part = E.mbconv_backward(dy, ctx.sv, ctx.p, need_dx=ctx.needs_input_grad[0], dy_part=part)
        E.tag_part(dx, dx_part)
        # norms without affine parameters (InstanceNorm2d) were passed as None: no gradient slot for them
        return (dx, None, None) + tuple(g[k] if ctx.needs_input_grad[3 + i] else None for i, k in enumerate(E.MB_KEYS))


class MBConv(TemporallySharedBlock):
    def __init__(self, inp, oup, downsample=False, expansion=4, norm='batch', n_groups=4):
        super().__init__()
        if downsample or expansion == 1 or inp != oup:
            raise NotImplementedError("HIP MBConv is built for the UNCRTAINTS configuration: no down-sampling, "
                                      "expansion > 1, inp == oup")
        if norm not in ("group", "batch", "instance"):
            raise NotImplementedError(f"MBConv norm '{norm}' is not built (group | batch | instance)")
        self.downsample = downsample
        hidden_dim = int(inp * expansion)
        self.conv = nn.Sequential(
            nn.Conv2d(inp, hidden_dim, 1, stride=1, padding=0, bias=False),
            get_norm_layer(hidden_dim, hidden_dim, n_groups, norm),
            nn.GELU(),
            nn.Conv2d(hidden_dim, hidden_dim, 3, stride=1, padding=1, padding_mode='reflect',
                      groups=hidden_dim, bias=False),
            get_norm_layer(hidden_dim, hidden_dim, n_groups, norm),
            nn.GELU(),
            SE(inp, hidden_dim),
            nn.Conv2d(hidden_dim, oup, 1, stride=1, padding=0, bias=False),
            get_norm_layer(oup, oup, n_groups, norm),
        )
        self.conv = PreNorm(inp, self.conv, norm, n_groups=4)
        self._spec = E.NormSpec(norm, n_groups)

    def _norms(self):
        f = self.conv.fn
        return (self.conv.norm, f[1], f[4], f[8])

    def _bn_buffers(self):
        out = {}
        for i, m in enumerate(self._norms()):
            if isinstance(m, nn.BatchNorm2d):
                out[f"n{i}rm"], out[f"n{i}rv"] = m.running_mean, m.running_var
        return out

    def _params(self):
        f = self.conv.fn
        n0, n1, n2, n3 = self._norms()
        return (n0.weight, n0.bias, f[0].weight, n1.weight, n1.bias, f[3].weight, n2.weight, n2.bias,
                f[6].fc[0].weight, f[6].fc[2].weight, f[7].weight, n3.weight, n3.bias)

    def forward(self, x):
        y = _MBConvFn.apply(x, self, not torch.is_grad_enabled(), *self._params())
        if self.training and not getattr(self, "_nbt_deferred", False):
            for m in self._norms():
                if isinstance(m, nn.BatchNorm2d):
                    m.num_batches_tracked += 1
        return y


class _ResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        p = dict(zip(E.RES_KEYS, params))
        y, sv = E.residual_forward(x.contiguous(), p, module._spec, module.training, module._buffers_dict())
        ctx.sv, ctx.p = sv, p
        if getattr(module, "keep_relu_branch", False):
            # parity tests differentiate the oracle on the branch of every ReLU this forward took (like `_last_pool_idx`):
            # the three pre-norm conv outputs and the norm coefficients, from which the masks [A*c + B > 0] follow
            module._last_relu = [(c, nf.A, nf.B) for c, nf in zip(sv["c"], sv["nf"])]
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, g = E.residual_backward(dy, ctx.sv, ctx.p, ctx.needs_input_grad[0])
        return (dx, None) + tuple(g[k] if ctx.needs_input_grad[2 + i] else None for i, k in enumerate(E.RES_KEYS))


class ResidualConvBlock(TemporallySharedBlock):
    """uncrtaints.py:24-69: x + CL3(CL2(CL1(x))), every ConvLayer = dense conv3x3 (reflect, bias) -> norm -> ReLU.
    Parameter holder with the reference's attribute paths (conv{1,2,3}.conv.{0,1}); compute in engine.residual_*."""

    def __init__(self, nkernels, pad_value=None, norm="batch", n_groups=4, k=3, s=1, p=1, padding_mode="reflect"):
        super().__init__(pad_value=pad_value)
        if (k, s, p) != (3, 1, 1) or padding_mode != "reflect" or len(nkernels) != 2 or nkernels[0] != nkernels[1]:
            raise NotImplementedError("ResidualConvBlock is built for 3x3 / stride 1 / reflect, equal widths")
        if norm not in ("batch", "group", "instance"):
            raise NotImplementedError(f"ResidualConvBlock norm '{norm}'")
        mk = lambda: ConvLayer(nkernels=nkernels, norm=norm, last_relu=True, k=k, s=s, p=p, n_groups=n_groups,
                               padding_mode=padding_mode)
        self.conv1, self.conv2, self.conv3 = mk(), mk(), mk()
        self._spec = E.NormSpec(norm, n_groups)

    def _layers(self):
        return [(c.conv[0], c.conv[1]) for c in (self.conv1, self.conv2, self.conv3)]

    def _buffers_dict(self):
        out = {}
        for i, (_, n) in enumerate(self._layers(), 1):
            if isinstance(n, nn.BatchNorm2d):
                out[f"rm{i}"], out[f"rv{i}"] = n.running_mean, n.running_var
        return out

    def forward(self, x):
        params = []
        for conv, nrm in self._layers():
            params += [conv.weight, conv.bias, nrm.weight, nrm.bias]
        y = _ResidualFn.apply(x, self, *params)
        if self.training:
            for _, n in self._layers():
                if isinstance(n, nn.BatchNorm2d):
                    n.num_batches_tracked += 1
        return y


class _AggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, att, pad, module, dmask):
        seed = module._next_seed()
        mode = module.mode
        x, att = x.contiguous(), att.contiguous()
        B, T = x.shape[:2]
        nh, ah, aw = att.shape[0], att.shape[-2], att.shape[-1]
        p_drop, shared = module.attn_dropout.p, False
        if mode == "att_mean":
            w_att, shared = E.head_mean_attention(att), True
        elif mode == "mean":
            w_att, p_drop, dmask = E.mean_mode_weights(pad, nh, B, T, ah, aw, x.device), 0.0, None
        else:
            w_att = att
            if x.shape[-2] <= aw:        # AvgPool branch of the reference (identity at equal size): no dropout (uncrtaints.py:197-204)
                p_drop, dmask = 0.0, None
        g, sv, gpart = E.aggregate_forward(x, w_att, pad, module.training, p_drop, seed, dmask, True, shared)
        ctx.sv, ctx.mode = sv, mode
        g._uncr_part = gpart
        return g

    @staticmethod
    def backward(ctx, dg):
        de, datt = E.aggregate_backward(dg, ctx.sv)
        if ctx.mode == "att_mean":
            datt = E.head_mean_attention_backward(datt)
        elif ctx.mode == "mean":
            datt = None
        return de, datt, None, None, None


class Compact_Temporal_Aggregator(nn.Module):
    def __init__(self, mode="mean"):
        super().__init__()
        self.mode = mode
        self.attn_dropout = nn.Dropout(0.1)   # applied after up-sampling, inside the HIP kernel (train only)
        self._seed_base = 0x5EED
        self._calls = 0
        self.dropout_mask = None              # optional explicit mask [n_head*B,T,H,W] (testing / parity)
        self.step_counter = None              # optional device int64[1] step counter (HIP-graph friendly seeding)

    def _next_seed(self):
        """Seed of this call's dropout stream.  With `step_counter` set (a device int64 tensor the training loop
        increments once per step, e.g. inside a captured HIP graph) the per-step variation comes from the device;
        otherwise from a host-side call counter."""
        if self.step_counter is not None:
            return ((self._seed_base * 1000003) & 0xFFFFFFFFFFFF, self.step_counter)
        self._calls += 1
        return (self._seed_base * 1000003 + self._calls) & 0xFFFFFFFFFFFF

    def set_seed(self, seed: int):
        """Seed of the counter-based dropout stream (use seed + rank under data parallelism)."""
        self._seed_base, self._calls = int(seed), 0

    def forward(self, x, pad_mask=None, attn_mask=None):
        if self.mode not in ("att_group", "att_mean", "mean"):
            raise NotImplementedError(f"agg_mode '{self.mode}'")
        pad = None
        if pad_mask is not None:
            pad = pad_mask if pad_mask.dtype == torch.int32 else pad_mask.to(torch.int32)
            pad = pad.contiguous()
        return _AggregateFn.apply(x, attn_mask, pad, self, self.dropout_mask)


def get_nonlinearity(mode, eps):
    if mode == 'softplus':
        fct = lambda vars: nn.Softplus(beta=1, threshold=20)(vars) + eps
    elif mode == 'elu':
        fct = lambda vars: nn.ELU()(vars) + 1 + eps
    elif mode == 'relu':
        raise TypeError("out_nonlin_var='relu' is broken in the reference too (uncrtaints.py:224: nn.ReLU() + eps)")
    else:
        fct = nn.Identity()
    return fct


class _StageFn(torch.autograd.Function):
    """max-pool + L-TAE attention + aggregation as one autograd node (uncrtaints.py:402-412)."""

    @staticmethod
    def forward(ctx, e4, dates, pad, net, dmask, b, t, *params):
        # e4: the encoder output on the folded frames [B*T, C, H, W] (no autograd view between the encoder and this node, so the
        # statistics / pooled values riding on the tensors survive in both directions)
        p = dict(zip(_LTAE_KEYS, params))
        e = e4.contiguous().view(b, t, *e4.shape[1:])
        ctx.e_h3 = getattr(e4, "_uncr_h3", None)
        ctx.bt = (b, t)
        te, agg = net.temporal_encoder, net.temporal_aggregator
        denom = te.positional_encoder.denom_on(e.device) if te.positional_encoder is not None else None
        # (sum, sum^2) partials of the aggregated features: the first decoder block's norm statistics in train mode; in eval mode
        # (running statistics) they still bound the tensor's magnitude for that block's fp16 two-part GEMM (engine.mbconv_forward)
        want_stats = net.out_block[0]._spec.needs_stats(net.training) or (net.block_type == 'mbconv' and e4.dtype == torch.float32)
        net._last_pad = pad
        values = None
        ctx.use_v = getattr(net, "use_v", False)
        if ctx.use_v:            # params = L-TAE keys, value-branch keys, include_v weight and bias
            nk = len(_LTAE_KEYS)
            vp = dict(zip(_LTAEV_KEYS, params[nk:nk + len(_LTAEV_KEYS)]))
            bn = te.mlp[1]
            seed = agg._next_seed()
            vseed = (seed[0] ^ 0x5bd1e995, seed[1]) if isinstance(seed, tuple) else (seed ^ 0x5bd1e995)
            values = dict(p=vp, include_w=params[-2], include_b=params[-1], bn_buffers=(bn.running_mean, bn.running_var),
                          p_drop=te.dropout.p, seed=vseed)
        g, sv, gpart, att = E.ltae_stage_forward(e.contiguous(), dates, pad, p, denom, te.n_head,
                                                 te.attention_heads.d_k, 32, net.training, agg.attn_dropout.p,
                                                 agg._next_seed(), dmask, want_stats, mode=agg.mode, values=values,
                                                 pooled=getattr(e4, "_uncr_pooled", None))
        if ctx.use_v and net.training:
            te.mlp[1].num_batches_tracked += 1
        if ctx.use_v and getattr(te, "keep_relu_branch", False):
            # parity tests differentiate the oracle on the branch the value MLP's ReLU took here (like `_last_pool_idx`): its
            # pre-norm input [B, C, S] and the BatchNorm coefficients per (b, c), from which the mask [A*m + B > 0] follows
            nf = sv["val"]["nf"]
            te._last_relu = (sv["val"]["m1"], nf.A, nf.B)
        ctx.sv, ctx.p, ctx.te = sv, p, te
        net._last_attention = att
        net._last_pool_idx = sv["idx"]        # arg-max of the 32x32 max-pool (flat in-plane index per (frame, channel, cell))
        g._uncr_part = gpart
        return g

    @staticmethod
    def backward(ctx, dg):
        de, g, part = E.ltae_stage_backward(dg, ctx.sv, ctx.p, ctx.te.n_head, ctx.te.attention_heads.d_k, e_h3=ctx.e_h3)
        grads = tuple(g[k] for k in _LTAE_KEYS)
        if ctx.use_v:
            grads += tuple(g[k] for k in _LTAEV_KEYS) + (g["include_w"], g["include_b"])
        b, t = ctx.bt
        de4 = de.view(b * t, *de.shape[2:])
        E.tag_part(de4, part)          # (sum de, sum de*h3) for the last encoder block's norm-3 backward
        return (de4, None, None, None, None, None, None) + grads


class _CastFn(torch.autograd.Function):
    """The model input converted to the activation storage type (bf16 activations); its gradient comes back as fp32."""

    @staticmethod
    def forward(ctx, x, dt):
        return E.cast(x, dt)

    @staticmethod
    def backward(ctx, g):
        return E.cast(g.contiguous(), E.F32), None


class _EmbedFn(torch.autograd.Function):
    """Any H x W (engine.Geom, csrc/anysize.hip): dense [planes..., H, W] -> padded planes [planes..., 1, Pc] with a zero tail."""

    @staticmethod
    def forward(ctx, x, geom):
        ctx.geom = geom
        return E.embed_tail(x, geom)

    @staticmethod
    def backward(ctx, g):
        return E.extract_tail(g.contiguous().float(), ctx.geom), None


class _ExtractFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geom):
        ctx.geom = geom
        return E.extract_tail(x, geom)

    @staticmethod
    def backward(ctx, g):
        return E.embed_tail(g, ctx.geom), None


class _HeadFn(torch.autograd.Function):
    """out_conv (1x1 + bias) + mean/variance nonlinearities (uncrtaints.py:432-445)."""

    @staticmethod
    def forward(ctx, y, w, b, net):
        out, sv = E.head_forward(y.contiguous(), w, b, net.mean_idx, net._mean_sigmoid, float(net.scale_by), net._eps,
                                 getattr(net, "_var_mode", "softplus"))
        ctx.sv, ctx.w = sv, w
        return out

    @staticmethod
    def backward(ctx, dout):
        dy, dW, db, dy_part = E.head_backward(dout, ctx.sv, ctx.w, ctx.needs_input_grad[0])
        E.tag_part(dy, dy_part)
        return dy, dW, db, None


class UNCRTAINTS(nn.Module):
    def __init__(
        self,
        input_dim,
        encoder_widths=[128],
        decoder_widths=[128, 128, 128, 128, 128],
        out_conv=[S2_BANDS],
        out_nonlin_mean=False,
        out_nonlin_var='relu',
        agg_mode="att_group",
        encoder_norm="group",
        decoder_norm="batch",
        n_head=16,
        d_model=256,
        d_k=4,
        pad_value=0,
        padding_mode="reflect",
        positional_encoding=True,
        covmode='diag',
        scale_by=1,
        separate_out=False,
        use_v=False,
        block_type='mbconv',
        is_mono=False
    ):
        super().__init__()
        self.n_stages = len(encoder_widths)
        self.encoder_widths = encoder_widths
        self.decoder_widths = decoder_widths
        self.out_widths = out_conv
        self.is_mono = is_mono
        self.use_v = use_v
        self.block_type = block_type
        self.enc_dim = decoder_widths[0] if decoder_widths is not None else encoder_widths[0]
        self.stack_dim = sum(decoder_widths) if decoder_widths is not None else sum(encoder_widths)
        self.pad_value = pad_value
        self.padding_mode = padding_mode
        self.scale_by = scale_by
        self.separate_out = separate_out

        if decoder_widths is not None:
            assert encoder_widths[-1] == decoder_widths[-1]
        else:
            decoder_widths = encoder_widths
        if not is_mono and (encoder_widths[-1] % n_head or encoder_widths[-1] // n_head > 32):
            # (2, 4, 6, 8, 16 or 32 channels per head take the streaming aggregation kernels, other counts the scalar ones)
            raise NotImplementedError(f"encoder width {encoder_widths[-1]} with n_head={n_head}: the L-TAE kernels take at most 32 channels "
                                      "per head, and the width must divide into the heads (uncrtaints.py:206)")
        # d_model beyond 256 runs wherever the fused L-TAE kernels apply (use_v off: the d_model-wide projections are folded into one
        # [n_head, C] functional and never exist as activations, csrc/ltae_fused.hip); the unfused path's GEMMs end at 256 channels
        # MBConv blocks wider than 128 channels (hidden width > 256) run with the hidden axis in groups (engine._mbconv_forward_wide)
        if (d_model is not None and d_model > 256 and use_v) or max(list(encoder_widths) + list(decoder_widths)) > 256:
            raise NotImplementedError("the GEMM kernels take at most 256 channels per operand: block widths <= 256, d_model <= 256 with use_v")
        if block_type not in ('mbconv', 'residual'):
            raise NotImplementedError(block_type)
        # use_v with is_mono: the reference builds neither a temporal encoder nor include_v then (uncrtaints.py:322-348) -- the flag is
        # stored and has no effect, as there
        if agg_mode not in ("att_group", "att_mean", "mean"):
            raise NotImplementedError(f"agg_mode '{agg_mode}'")
        # padding_mode is stored and never used by the reference's UNCRTAINTS either (uncrtaints.py:299 is its only use: MBConv
        # hard-codes 'reflect' at :115/:130, ResidualConvBlock is built with its default 'reflect' at :319/:353, in_conv / out_conv
        # are 1x1 without padding) -- any value is accepted and the convolutions pad by reflection, exactly like the reference
        if any(w != encoder_widths[0] for w in encoder_widths):
            # the reference builds MBConv(w, w) per entry behind an in_conv of width encoder_widths[0] (uncrtaints.py:309-319): its
            # forward only type-checks when all entries are equal
            raise ValueError(f"encoder_widths {list(encoder_widths)}: every encoder block maps w -> w, so all entries must be equal")

        self.encoder_norm = encoder_norm
        self.in_conv = ConvBlock(nkernels=[input_dim] + [encoder_widths[0]], k=1, s=1, p=0, norm=encoder_norm)
        if block_type == 'residual':      # uncrtaints.py:318-319
            self.in_block = nn.ModuleList([ResidualConvBlock(nkernels=[layer] + [layer], k=3, s=1, p=1,
                                                             norm=encoder_norm, n_groups=4) for layer in encoder_widths])
        else:
            self.in_block = nn.ModuleList([MBConv(layer, layer, downsample=False, expansion=2, norm=encoder_norm)
                                           for layer in encoder_widths])
        if not self.is_mono:     # uncrtaints.py:322-348
            if use_v:            # uncrtaints.py:324-338
                self.temporal_encoder = LTAE2d(in_channels=encoder_widths[0], d_model=d_model, n_head=n_head,
                                               mlp=[d_model, encoder_widths[0]], return_att=True, d_k=d_k,
                                               positional_encoding=positional_encoding, use_dropout=False)
                self.include_v = nn.Conv2d(encoder_widths[0] + encoder_widths[0], encoder_widths[0], 1)
            else:
                self.temporal_encoder = LTAE2dtiny(in_channels=encoder_widths[0], d_model=d_model, n_head=n_head,
                                                   d_k=d_k, positional_encoding=positional_encoding)
            self.temporal_aggregator = Compact_Temporal_Aggregator(mode=agg_mode)
        if block_type == 'residual':      # uncrtaints.py:352-353
            self.out_block = nn.ModuleList([ResidualConvBlock(nkernels=[layer] + [layer], k=3, s=1, p=1,
                                                              norm=decoder_norm, n_groups=4) for layer in decoder_widths])
        else:
            self.out_block = nn.ModuleList([MBConv(layer, layer, downsample=False, expansion=2, norm=decoder_norm)
                                            for layer in decoder_widths])

        self.covmode = covmode
        if covmode == 'uni':
            covar_dim = S2_BANDS
        elif covmode == 'iso':
            covar_dim = 1
        elif covmode == 'diag':
            covar_dim = S2_BANDS
        else:
            covar_dim = 0
        self.mean_idx = S2_BANDS
        self.vars_idx = self.mean_idx + covar_dim
        self.out_dims = out_conv[-1]
        eps = 1e-9 if self.scale_by == 1.0 else 1e-3
        self._eps = eps
        self._mean_sigmoid = bool(out_nonlin_mean)
        if self.separate_out:    # two 1x1 streams for mean and variance (uncrtaints.py:376-379)
            self.out_conv_mean_1 = ConvBlock(nkernels=[decoder_widths[0]] + [S2_BANDS], k=1, s=1, p=0, norm='none',
                                             last_relu=False)
            if self.out_dims - self.mean_idx > 0:
                self.out_conv_var_1 = ConvBlock(nkernels=[decoder_widths[0]] + [self.out_dims - S2_BANDS], k=1, s=1,
                                                p=0, norm='none', last_relu=False)
        else:
            self.out_conv = ConvBlock(nkernels=[decoder_widths[0]] + out_conv, k=1, s=1, p=0, norm='none',
                                      last_relu=False)

        if out_nonlin_mean:
            self.out_mean = lambda vars: self.scale_by * nn.Sigmoid()(vars)
        else:
            self.out_mean = nn.Identity()
        if self.covmode in ['uni', 'iso', 'diag']:
            # 'softplus' is what the reference's diag/iso/uni fix-up always selects (train_reconstruct.py:53-61); 'elu' and
            # the identity fall-through of get_nonlinearity are built too ('relu' raises like the reference, above)
            self._var_mode = out_nonlin_var if out_nonlin_var in ('softplus', 'elu') else 'identity'
            self.diag_var = get_nonlinearity(out_nonlin_var, eps)
            if self.out_dims < self.vars_idx:
                raise ValueError(f"out_conv[-1]={self.out_dims} < 13 + covar_dim={self.vars_idx}")
        self.variance = None
        self._last_attention = None
        self._last_pool_idx = None
        # keep_boundaries = True: forward records the two tensors at which a training loop may cut the backward pass into the
        # three gradient-bucket segments of uncrtaints_amd.parallel (decoder + head | temporal encoder | encoder):
        # _boundary_enc (encoder output [B,T,C,H,W]) and _boundary_agg (aggregated features [B,C,H,W])
        self.keep_boundaries = False
        self._boundary_enc = self._boundary_agg = None
        self.act_dtype = torch.float32      # storage of the activations: see set_act_dtype

    def set_act_dtype(self, dtype):
        """Storage type of every full-resolution activation and activation gradient of the path: torch.float32 (the
        reference's arithmetic; parity contract 1e-4) or torch.bfloat16 ("bf16 activations, fp32 accumulate", BASELINE config 3:
        half the HBM bytes per step; statistics, accumulators, weights, weight gradients, the 32x32 attention branch, the outputs
        and the loss stay fp32; parity contract stated in tests/test_bf16.py).  Parameters and the module API are unchanged."""
        if isinstance(dtype, str):
            dtype = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}[dtype]
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("act_dtype must be torch.float32 or torch.bfloat16")
        if dtype == torch.bfloat16:
            if self.block_type != 'mbconv' or self.use_v:
                raise NotImplementedError("bf16 activations are built for block_type='mbconv' without use_v")
            if self.out_dims > 64:
                raise NotImplementedError("bf16 activations: out_conv wider than 64 channels is not built")
            if len(self.out_widths) > 1:
                raise NotImplementedError("bf16 activations: a multi-layer out_conv is built for fp32 storage")
        self.act_dtype = dtype
        return self

    def _pack_list(self):
        """(weight as [Cout][Cin], transpose) for every pointwise GEMM of forward and backward (engine.pack_wt calls)."""
        out = []

        def both(w):
            w2 = w.reshape(w.shape[0], -1)
            out.append((w2, True))
            if self.training or torch.is_grad_enabled():
                out.append((w2, False))
        for mod in self.modules():
            if isinstance(mod, MBConv):
                f = mod.conv.fn
                if f[0].weight.shape[0] <= 256:     # (wider hidden axis: engine's grouped path packs its own slices)
                    both(f[0].weight)
                    both(f[7].weight)
        both(self.in_conv.conv.conv[0].weight)
        if not self.is_mono:
            te = self.temporal_encoder
            p = _ltae_params(te)
            if p["inconv_w"].shape[0] <= 256:       # (wider: only the fused kernels run it, and they read the raw weights)
                both(p["inconv_w"])
                both(p["fc_w"])
        if not self.separate_out:
            for mod in self.out_conv.conv.conv:
                if isinstance(mod, nn.Conv2d):
                    both(mod.weight)
        return out

    def forward(self, input, batch_positions=None):
        if not input.is_cuda:
            raise RuntimeError("uncrtaints_amd.UNCRTAINTS runs on the GPU only (HIP kernels); move the model and "
                               "the inputs to the cuda device")
        conv0 = self.in_conv.conv.conv[0]
        if input.dim() != 5 or input.shape[2] != conv0.in_channels:
            raise ValueError(f"expected input [B, T, {conv0.in_channels}, H, W], got {tuple(input.shape)}")
        if batch_positions is not None and tuple(batch_positions.shape) != tuple(input.shape[:2]):
            raise ValueError(f"batch_positions {tuple(batch_positions.shape)} must be [B, T] = {tuple(input.shape[:2])}")
        input = input.contiguous().float()
        E.prepack(self._pack_list(), owner=self)                           # every 1x1-conv weight, one launch
        if self.training:
            # BatchNorm bookkeeping of all MBConv blocks in one multi-tensor launch (20 scalar-add kernels otherwise)
            nbt = []       # rebuilt per call: buffers are re-created by .to() / load_state_dict(assign=True)
            for mod in self.modules():
                if isinstance(mod, MBConv):
                    mod._nbt_deferred = True
                    nbt += [n.num_batches_tracked for n in mod._norms() if isinstance(n, nn.BatchNorm2d)]
            if nbt:
                torch._foreach_add_(nbt, 1)
        pad = E.pad_mask_of(input, float(self.pad_value))                  # [B,T] int32, uncrtaints.py:392-394
        # the encoder runs on the folded [B*T, C, H, W] frames (smart_forward, utae.py:422-450) without autograd views
        # between its blocks, so the statistics / masks that ride on the tensors survive in both directions
        b, t, _, h, w = input.shape
        x4 = input.view(b * t, input.shape[2], h, w)
        # any H x W (uncrtaints.py:391-447): a size outside the tuned tilings runs on padded planes [frames, C, 1, Pc] (dense H*W pixels
        # + a zero tail) inside a geometry scope; the output is cut back to [B, 1, C_out, H, W] at the end
        geom = E.plan_geom(h, w)
        if geom is not None:
            if self.act_dtype == torch.bfloat16:
                raise NotImplementedError(f"spatial size {h}x{w} (H*W not a multiple of 1024 or W not of 4) is built for fp32 storage")
            x4 = _EmbedFn.apply(x4, geom)
        with E.geom_scope(geom):
            return self._forward_frames(x4, batch_positions, pad, b, t, h, w, geom)

    def _forward_frames(self, x4, batch_positions, pad, b, t, h, w, geom):
        if self.act_dtype == torch.bfloat16:           # everything downstream allocates in the storage type of its input
            x4 = _CastFn.apply(x4, E.BF16)
        x4 = self.in_conv(x4)
        if self.keep_boundaries:
            self._boundary_a0 = x4          # in_conv's relu(norm(c0)): parity tests read the branch its ReLU took from it (a0 > 0)
        pooled = None
        for li, layer in enumerate(self.in_block):
            if li == len(self.in_block) - 1 and not self.is_mono and h % 32 == 0 and w % 32 == 0:
                x4._uncr_pool = 32          # the stage's 32x32 max-pool rides on the last encoder block's residual kernel
            x4 = layer(x4)
            pooled = getattr(x4, "_uncr_pooled", None)
        part = getattr(x4, "_uncr_part", None)
        out = x4
        if not self.is_mono:
            if self.temporal_encoder.positional_encoder is not None and batch_positions is None:
                raise ValueError("batch_positions (dates) are required when positional_encoding=True")
            p = _ltae_params(self.temporal_encoder)
            extra = []
            if self.use_v:
                vp = _ltae_value_params(self.temporal_encoder)
                extra = [vp[k] for k in _LTAEV_KEYS] + [self.include_v.weight, self.include_v.bias]
            if self.keep_boundaries:
                self._boundary_enc = out
            out = _StageFn.apply(out, batch_positions, pad, self, self.temporal_aggregator.dropout_mask, b, t,
                                 *([p[k] for k in _LTAE_KEYS] + extra))
            if self.keep_boundaries:
                self._boundary_agg = out
        else:                                                              # uncrtaints.py:418
            if t != 1:
                raise ValueError("is_mono expects a single input date (T == 1)")
            # B*1 folded frames ARE the [B, C, H, W] decoder input (out.squeeze(dim=1) of the 5-D view in the reference)
        for layer in self.out_block:
            out = layer.smart_forward(out)
        if self.separate_out:
            # two 1x1 convolutions on the same input == one convolution with concatenated kernels
            cm = self.out_conv_mean_1.conv.conv[0]
            if self.out_dims - self.mean_idx > 0:
                cv = self.out_conv_var_1.conv.conv[0]
                w_all, b_all = torch.cat((cm.weight, cv.weight), dim=0), torch.cat((cm.bias, cv.bias), dim=0)
            else:
                w_all, b_all = cm.weight, cm.bias
        else:
            # out_conv with several layers (`--out_conv "[32,13]"`, parse_args.py:30): Conv2d + ReLU for every layer but the last
            # (utae.py:476-494 with norm='none', last_relu=False); the last one carries the output nonlinearities in its epilogue
            convs = [mod for mod in self.out_conv.conv.conv if isinstance(mod, nn.Conv2d)]
            for cl in convs[:-1]:
                out = _ConvBiasReluFn.apply(out, cl.weight, cl.bias)
            conv = convs[-1]
            w_all, b_all = conv.weight, conv.bias
        if not self.covmode:
            # mean only: plain conv + mean nonlinearity on all out_dims channels
            o = _HeadFnMeanOnly.apply(out, w_all, b_all, self)
            if geom is not None:
                o = _ExtractFn.apply(o, geom)
            return o.unsqueeze(1)[:, :, :self.mean_idx, ...]
        o = _HeadFn.apply(out, w_all, b_all, self)                         # [B, out_dims, H, W]
        if geom is not None:
            o = _ExtractFn.apply(o, geom)
        o = o.unsqueeze(1)
        if self.out_dims != self.vars_idx:
            o = o[:, :, :self.vars_idx, ...]
        return o


class _HeadFnMeanOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, w, b, net):
        out, sv = E.head_forward(y.contiguous(), w, b, w.shape[0], net._mean_sigmoid, float(net.scale_by), 0.0)
        ctx.sv, ctx.w = sv, w
        return out

    @staticmethod
    def backward(ctx, dout):
        dy, dW, db, dy_part = E.head_backward(dout, ctx.sv, ctx.w, ctx.needs_input_grad[0])
        E.tag_part(dy, dy_part)
        return dy, dW, db, None
