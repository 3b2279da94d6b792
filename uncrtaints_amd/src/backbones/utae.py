"""The three utae.py classes that sit on the `--model uncrtaints` path
(model/src/backbones/utae.py:409-520): TemporallySharedBlock, ConvLayer, ConvBlock.

They keep stock nn.Conv2d / nn.GroupNorm / nn.BatchNorm2d modules as PARAMETER HOLDERS (so `weight_init`,
`state_dict()` keys and `freeze_layers` behave as in the reference) and route compute to the HIP engine.
Only the configurations UNCRTAINTS builds are supported: 1x1 convolutions, one conv per block, norm in
{group, batch, none}; anything else raises NotImplementedError (no silent PyTorch fallback)."""
import torch
import torch.nn as nn

from ... import engine as E


class TemporallySharedBlock(nn.Module):
    """smart_forward folds T into the batch for 5-D inputs (utae.py:422-450)."""

    def __init__(self, pad_value=None):
        super().__init__()
        self.out_shape = None
        self.pad_value = pad_value

    def smart_forward(self, input):
        if len(input.shape) == 4:
            return self.forward(input)
        b, t, c, h, w = input.shape
        if self.pad_value is not None:
            # utae.py:433-446: frames that consist of pad_value only are not run through the block; their output is pad_value.
            # (UNCRTAINTS builds its blocks without pad_value; this is the class surface.)  The mask comes from the HIP
            # frame-scan kernel; selecting / re-inserting the frames is data movement (torch indexing, differentiable); the
            # `.any()` is the host sync the reference has at the same place.
            pm = E.pad_mask_of(input.contiguous().float(), float(self.pad_value)).view(-1).bool()
            if bool(pm.any()):
                keep = (~pm).nonzero().squeeze(1)
                if keep.numel() == 0:
                    raise ValueError("every frame of the sequence equals pad_value")
                y = self.forward(input.reshape(b * t, c, h, w).index_select(0, keep))
                self.out_shape = (b * t,) + tuple(y.shape[1:])
                full = torch.full(self.out_shape, float(self.pad_value), device=y.device, dtype=y.dtype)
                out4 = full.index_copy(0, keep, y)
                return out4.view(b, t, *out4.shape[1:])
        out4 = self.forward(input.reshape(b * t, c, h, w))
        _, c, h, w = out4.shape
        out = out4.view(b, t, c, h, w)
        part = getattr(out4, "_uncr_part", None)      # partial statistics ride along for the next PreNorm
        if part is not None:
            out._uncr_part = part
        return out


class _ConvNormActFn(torch.autograd.Function):
    """Conv2d(k=1,bias) [+ GroupNorm/BatchNorm] [+ ReLU] through the HIP engine."""

    @staticmethod
    def forward(ctx, x, w, b, gw, gb, layer):
        x = x.contiguous()
        spec = layer._spec
        if spec is None:
            raise NotImplementedError
        buffers = layer._bn_buffers()
        a0, sv, part = E.inconv_forward(x, w, b, gw, gb, spec, layer.training, buffers)
        ctx.sv, ctx.layer = sv, layer
        ctx.params = (w, gw)
        a0._uncr_part = part
        # lets the consumer's backward apply this ReLU's mask: from c0 (A*c0 + B > 0) or, where c0 was never stored, from a0 itself
        a0._uncr_relu = (sv["c0"], sv["nf"].A, sv["nf"].B if sv["c0"] is not None else None, sv.get("mu"))
        return a0

    @staticmethod
    def backward(ctx, da):
        w, gw = ctx.params
        part = E.claim_part(da)        # set when the consumer's backward already applied the ReLU mask (and da is untouched since)
        if part is not None and not (part.masked and part.buf.shape[0] == da.shape[0] * da.shape[1]):
            part = None
        dx, dW, db, dgw, dgb = E.inconv_backward(da, ctx.sv, w, gw, ctx.needs_input_grad[0], masked_part=part)
        if gw is None:          # InstanceNorm2d: no affine parameters
            dgw = dgb = None
        return dx, dW, db, dgw, dgb, None


class _ConvBiasFn(torch.autograd.Function):
    """Plain Conv2d(k=1,bias), no norm / activation (out_conv without the output nonlinearities)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        N, C, H, W = E._check4(x)
        Co = w.shape[0]
        Wt = E.pack_wt(w.reshape(Co, C), transpose=True)
        o, _ = E.pw_gemm(x, Wt, N, C, Co, H * W, bias=b.contiguous())
        ctx.save_for_backward(x, w)
        return o.view(N, Co, H, W)

    @staticmethod
    def backward(ctx, do):
        x, w = ctx.saved_tensors
        N, C, H, W = x.shape
        Co = w.shape[0]
        do = do.contiguous()
        dW, db = E.pw_wgrad(do, x, N, Co, C, H * W, rowsum=True)
        dx = None
        if ctx.needs_input_grad[0]:
            Wk = E.pack_wt(w.reshape(Co, C), transpose=False)
            dx, _ = E.pw_gemm(do, Wk, N, Co, C, H * W)
            dx = dx.view(N, C, H, W)
        return dx, dW.view_as(w), db


class _ConvBiasReluFn(torch.autograd.Function):
    """Conv2d(k=1, bias) + ReLU without a norm: the inner layers of a multi-layer out_conv (`--out_conv "[32,13]"`, utae.py:453-497 with
    norm='none', last_relu=False: a ReLU behind every convolution but the last)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        N, C, H, W = E._check4(x)
        Co = w.shape[0]
        pre, _ = E.pw_gemm(x, E.pack_wt(w.reshape(Co, C), transpose=True), N, C, Co, H * W, bias=b.contiguous())
        one, zero = E._const_planes(x.device, N * Co)
        out = torch.empty_like(pre)
        E.ew(E.EW_AFFINE_RELU, pre, out=out, k=(one, zero, None, None), planes=N * Co, P=H * W)
        ctx.save_for_backward(x, w, pre)
        return out.view(N, Co, H, W)

    @staticmethod
    def backward(ctx, do):
        x, w, pre = ctx.saved_tensors
        N, C, H, W = x.shape
        Co = w.shape[0]
        one, zero = E._const_planes(x.device, N * Co)
        do = E.cast(do.contiguous(), E._dt(pre))
        dpre = torch.empty_like(pre)
        E.ew(E.EW_RELU_BWD, do, b=pre, out=dpre, k=(one, zero, None, None), planes=N * Co, P=H * W)
        dW, db = E.pw_wgrad(dpre, x, N, Co, C, H * W, rowsum=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _ = E.pw_gemm(dpre, E.pack_wt(w.reshape(Co, C), transpose=False), N, Co, C, H * W)
            dx = dx.view(N, C, H, W)
        return dx, dW.view_as(w), db


class ConvLayer(nn.Module):
    def __init__(self, nkernels, norm="batch", k=3, s=1, p=1, n_groups=4, last_relu=True, padding_mode="reflect"):
        super().__init__()
        layers = []
        if norm == "batch":
            nl = nn.BatchNorm2d
        elif norm == "instance":
            nl = nn.InstanceNorm2d
        elif norm == "group":
            nl = lambda num_feats: nn.GroupNorm(num_channels=num_feats, num_groups=n_groups)
        else:
            nl = None
        for i in range(len(nkernels) - 1):
            layers.append(nn.Conv2d(in_channels=nkernels[i], out_channels=nkernels[i + 1], kernel_size=k, padding=p,
                                    stride=s, padding_mode=padding_mode))
            if nl is not None:
                layers.append(nl(nkernels[i + 1]))
            if last_relu:
                layers.append(nn.ReLU())
            elif i < len(nkernels) - 2:
                layers.append(nn.ReLU())
        self.conv = nn.Sequential(*layers)
        self._k, self._s, self._p = k, s, p
        self._norm, self._n_groups, self._last_relu = norm, n_groups, last_relu
        self._nconv = len(nkernels) - 1
        if norm == "group":
            self._spec = E.NormSpec("group", n_groups)
        elif norm == "batch":
            self._spec = E.NormSpec("batch")
        elif norm == "instance":
            self._spec = E.NormSpec("instance")
        else:
            self._spec = None

    def _bn_buffers(self):
        m = self.conv[1] if len(self.conv) > 1 else None
        if isinstance(m, nn.BatchNorm2d):
            return dict(rm=m.running_mean, rv=m.running_var)
        return {}

    def forward(self, input):
        if self._k != 1 or self._s != 1 or self._p != 0:
            raise NotImplementedError("HIP ConvLayer is built for 1x1 convolutions (the UNCRTAINTS in_conv / out_conv configuration)")
        mods = list(self.conv)
        out, i = input, 0
        while i < len(mods):                     # Conv2d [norm] [ReLU] groups, as the constructor laid them out (utae.py:476-494)
            conv = mods[i]
            nrm = mods[i + 1] if (i + 1 < len(mods) and not isinstance(mods[i + 1], (nn.ReLU, nn.Conv2d))) else None
            j = i + 1 + (nrm is not None)
            relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
            i = j + (1 if relu else 0)
            if nrm is not None and relu:
                if self._nconv != 1:
                    raise NotImplementedError("HIP ConvLayer: a normalised 1x1 convolution is built as a single layer (in_conv)")
                out = _ConvNormActFn.apply(out, conv.weight, conv.bias, nrm.weight, nrm.bias, self)
                if isinstance(nrm, nn.BatchNorm2d) and self.training:
                    nrm.num_batches_tracked += 1
            elif nrm is None and relu:
                out = _ConvBiasReluFn.apply(out, conv.weight, conv.bias)
            elif nrm is None:
                out = _ConvBiasFn.apply(out, conv.weight, conv.bias)
            else:
                raise NotImplementedError(f"ConvLayer(norm={self._norm}, last_relu={self._last_relu}): a norm without a ReLU is not built")
        return out


class ConvBlock(TemporallySharedBlock):
    def __init__(self, nkernels, pad_value=None, norm="batch", last_relu=True, k=3, s=1, p=1, padding_mode="reflect"):
        super().__init__(pad_value=pad_value)
        self.conv = ConvLayer(nkernels=nkernels, norm=norm, last_relu=last_relu, k=k, s=s, p=p,
                              padding_mode=padding_mode)

    def forward(self, input):
        return self.conv(input)
