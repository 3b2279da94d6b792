"""Losses with the reference's surface (model/src/losses.py): get_loss, calc_loss, MultiGaussianNLLLoss,
multi_gaussian_nll_loss, GaussianNLLLoss, gaussian_nll_loss and the l1 / l2 criteria -- each one HIP streaming kernel
per direction (csrc/mgnll.hip).

Differences from the reference, all opt-in / documented:
  * the second result of the MGNLL is the clamped per-band variance [B,1,13,H,W] on the device; its dense form
    `diag_embed(var)` [B,1,13,13,H,W] (moved to the host inside the reference loss, losses.py:145,211; logging only) is
    produced only with `want_covariance=True`, on the device;
  * `torch.any(var < 0)` (a host sync, losses.py:110,199) is checked only with `check_negative=True`."""
import torch
import torch.nn as nn
from torch.nn.modules.loss import _Loss

from .. import engine as E

S2_BANDS = 13
Tensor = torch.Tensor


class _MGNLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, var, eps, reduction, check_negative):
        # channel slices of the head's output are read in place (engine._batch_strided); anything else is made dense
        pred = pred.float() if E._batch_strided(pred.float()) else pred.contiguous().float()
        var = var.float() if E._batch_strided(var.float()) else var.contiguous().float()
        target = target.contiguous().float()
        loss, vclamp = E.mgnll_forward(pred, target, var, eps, reduction, check_negative, want_variance=True)
        ctx.save_for_backward(pred, target, var)
        ctx.eps, ctx.reduction = eps, reduction
        ctx.mark_non_differentiable(vclamp)
        return loss, vclamp

    @staticmethod
    def backward(ctx, gout, _gv=None):
        pred, target, var = ctx.saved_tensors
        dpred, dvar = E.mgnll_backward(gout, pred, target, var, ctx.eps, ctx.reduction,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[2])
        return dpred, None, dvar, None, None, None


class _SplitFn(torch.autograd.Function):
    """(out[:, :, :k0], out[:, :, k0:k1]) as views.  Backward: when the two incoming gradients are the channel slices of one buffer
    (what the MGNLL backward kernel writes), that buffer IS the gradient of `out` -- no zero fills, embedding copies or add."""

    @staticmethod
    def forward(ctx, out, k0, k1):
        ctx.k0, ctx.k1, ctx.shape = k0, k1, tuple(out.shape)
        return out[:, :, :k0], out[:, :, k0:k1]

    @staticmethod
    def backward(ctx, dm, dv):
        k0, k1, shape = ctx.k0, ctx.k1, ctx.shape
        if dm is not None and dv is not None and shape[2] == k1:
            base = dm._base
            if (base is not None and base is dv._base and tuple(base.shape) == shape and base.is_contiguous()
                    and dm.stride() == base.stride() and dv.stride() == base.stride()
                    and dm.storage_offset() == base.storage_offset()
                    and dv.storage_offset() == base.storage_offset() + k0 * base.stride(2)):
                return base, None, None
        g = torch.zeros(shape, device=(dm if dm is not None else dv).device, dtype=(dm if dm is not None else dv).dtype)
        if dm is not None:
            g[:, :, :k0] = dm
        if dv is not None:
            g[:, :, k0:k1] = dv
        return g, None, None


def split_prediction(out: Tensor, mean_idx: int, vars_idx: int):
    """Mean and variance channels of the model output [B,1,13+cov,H,W] -- what base_model.py:72-85 slices out -- for the loss."""
    return _SplitFn.apply(out, int(mean_idx), int(vars_idx))


def multi_gaussian_nll_loss(input: Tensor, target: Tensor, var: Tensor, full: bool = False, eps: float = 1e-8,
                            reduction: str = "mean", mode: str = "diag", chunk=None, want_covariance: bool = False,
                            check_negative: bool = False):
    """(loss, variance) like losses.py:149-218.  input/target [B,1,13,H,W], var [B,1,13|1,H,W]."""
    if reduction != 'none' and reduction != 'mean' and reduction != 'sum':
        raise ValueError(reduction + " is not valid")
    if mode not in ("diag", "iso"):
        raise NotImplementedError(f"MGNLL mode '{mode}' is not built (diag | iso)")
    if input.dim() != 5 or input.shape[1] != 1:
        raise ValueError("expected [B,1,C,H,W] tensors")
    if target.shape != input.shape:
        raise ValueError(f"target {tuple(target.shape)} and input {tuple(input.shape)} differ in shape")
    if var.dim() != 5 or var.shape[:2] != input.shape[:2] or var.shape[3:] != input.shape[3:] \
            or (var.shape[2] != input.shape[2] and (mode != "iso" or var.shape[2] < 1)):
        raise ValueError(f"var {tuple(var.shape)} does not match input {tuple(input.shape)} (mode '{mode}')")
    if mode == "iso" and var.shape[2] != 1:
        var = var[:, :, :1]
    # second result: the clamped per-band variance [B,1,13,H,W] on the device (iso: the channel broadcast to the 13 bands),
    # written by the loss kernel itself.  The reference returns its dense form diag_embed(var) [B,1,13,13,H,W] on the HOST
    # (losses.py:145,211); the validation / test loops (train_reconstruct.py:302-329) and img_metrics accept the 5-D
    # per-band form, which carries the same numbers.  The dense layout is opt-in (`want_covariance=True`).
    loss, variance = _MGNLLFn.apply(input, target, var, float(eps), reduction, bool(check_negative))
    if want_covariance:   # logging-only export, losses.py:145,211 (layout [B,1,13,13,H,W])
        variance = torch.diag_embed(variance[:, 0].permute(0, 2, 3, 1)).permute(0, 3, 4, 1, 2).unsqueeze(1)
    return loss, variance


class MultiGaussianNLLLoss(_Loss):
    __constants__ = ['full', 'eps', 'reduction']

    def __init__(self, *, full: bool = False, eps: float = 1e-8, reduction: str = 'mean', mode: str = 'diag',
                 chunk=None, want_covariance: bool = False, check_negative: bool = False) -> None:
        super().__init__(None, None, reduction)
        self.full, self.eps, self.mode, self.chunk = full, eps, mode, chunk
        self.want_covariance, self.check_negative = want_covariance, check_negative

    def forward(self, input: Tensor, target: Tensor, var: Tensor):
        return multi_gaussian_nll_loss(input, target, var, full=self.full, eps=self.eps, reduction=self.reduction,
                                       mode=self.mode, chunk=self.chunk, want_covariance=self.want_covariance,
                                       check_negative=self.check_negative)


class _EltLossFn(torch.autograd.Function):
    """GaussianNLL / L1 / L2 as one streaming HIP pass per direction (engine.eltloss_forward/backward)."""

    @staticmethod
    def forward(ctx, kind, pred, target, var, eps, full, reduction, check_negative):
        if target.shape != pred.shape or (var is not None and var.shape != pred.shape):
            raise ValueError(f"loss operands differ in shape: input {tuple(pred.shape)}, target {tuple(target.shape)}"
                             + (f", var {tuple(var.shape)}" if var is not None else ""))
        pred_c, targ_c = pred.contiguous().float(), target.contiguous().float()
        var_c = var.contiguous().float() if var is not None else None
        loss, vclamp = E.eltloss_forward(kind, pred_c, targ_c, var_c, eps, full, reduction, check_negative)
        ctx.save_for_backward(pred_c, targ_c, var_c if var_c is not None else pred_c)
        ctx.meta = (kind, eps, reduction, var is not None)
        if vclamp is not None:
            ctx.mark_non_differentiable(vclamp)
            return loss, vclamp
        return loss

    @staticmethod
    def backward(ctx, gout, *_):
        kind, eps, reduction, has_var = ctx.meta
        pred, targ, var = ctx.saved_tensors
        dpred, dvar = E.eltloss_backward(kind, gout, pred, targ, var if has_var else None, eps, reduction,
                                         ctx.needs_input_grad[1], has_var and ctx.needs_input_grad[3])
        return None, dpred, None, dvar, None, None, None, None


def gaussian_nll_loss(input: Tensor, target: Tensor, var: Tensor, full: bool = False, eps: float = 1e-8,
                      reduction: str = "mean", check_negative: bool = False):
    """losses.py:46-128.  Returns (loss, clamped var).  The reference's `torch.any(var < 0)` host sync is opt-in
    (`check_negative=True`), like MultiGaussianNLLLoss here."""
    if var.size() != input.size():
        if input.size()[:-1] == var.size():
            var = torch.unsqueeze(var, dim=-1)
        elif not (input.size()[:-1] == var.size()[:-1] and var.size(-1) == 1):
            raise ValueError("var is of incorrect size")
        var = var.expand_as(input)      # homoscedastic form: autograd sums the gradient back over the broadcast axis
    if reduction not in ("none", "mean", "sum"):
        raise ValueError(reduction + " is not valid")
    loss, v = _EltLossFn.apply(E.ELT_GNLL, input, target, var, eps, full, reduction, check_negative)
    return loss, v


class GaussianNLLLoss(_Loss):
    """losses.py:131-ish class surface: GaussianNLLLoss(full=, eps=, reduction=)(input, target, var) -> (loss, var)."""

    def __init__(self, *, full: bool = False, eps: float = 1e-8, reduction: str = 'mean', check_negative: bool = False):
        super().__init__(None, None, reduction)
        self.full, self.eps, self.check_negative = full, eps, check_negative

    def forward(self, input: Tensor, target: Tensor, var: Tensor):
        return gaussian_nll_loss(input, target, var, full=self.full, eps=self.eps, reduction=self.reduction,
                                 check_negative=self.check_negative)


class L1Loss(_Loss):
    """nn.L1Loss() stand-in of get_loss 'l1' (mean reduction)."""

    def __init__(self, reduction: str = 'mean'):
        super().__init__(None, None, reduction)

    def forward(self, input: Tensor, target: Tensor):
        return _EltLossFn.apply(E.ELT_L1, input, target, None, 0.0, False, self.reduction, False)


class MSELoss(_Loss):
    """nn.MSELoss() stand-in of get_loss 'l2' (mean reduction)."""

    def __init__(self, reduction: str = 'mean'):
        super().__init__(None, None, reduction)

    def forward(self, input: Tensor, target: Tensor):
        return _EltLossFn.apply(E.ELT_L2, input, target, None, 0.0, False, self.reduction, False)


def get_loss(config):
    """losses.py:14-32"""
    if config.loss == "GNLL":
        criterion1 = GaussianNLLLoss(reduction='mean', eps=1e-8, full=True)
        return lambda pred, targ, var: criterion1(pred, targ, var)
    if config.loss == "MGNLL":
        criterion1 = MultiGaussianNLLLoss(reduction='mean', eps=1e-8, full=True, mode=config.covmode,
                                          chunk=getattr(config, "chunk_size", None),
                                          want_covariance=getattr(config, "want_covariance", False))
        return lambda pred, targ, var: criterion1(pred, targ, var)
    if config.loss == "l1":
        criterion1 = L1Loss()
        return lambda pred, targ: criterion1(pred, targ)
    if config.loss == "l2":
        criterion1 = MSELoss()
        return lambda pred, targ: criterion1(pred, targ)
    raise NotImplementedError


def calc_loss(criterion, config, out, y, var=None):
    """losses.py:35-43"""
    if config.loss in ['GNLL', 'MGNLL']:
        loss, variance = criterion(out, y, var)
    else:
        loss, variance = criterion(out, y), None
    return loss, variance
