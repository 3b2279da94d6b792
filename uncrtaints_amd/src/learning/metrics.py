"""Evaluation metrics with the reference's surface (model/src/learning/metrics.py: img_metrics, avg_img_metrics, Metric;
util/pytorch_ssim: ssim, create_window, gaussian), computed by uncr_img_metrics on the device: RMSE, MAE, PSNR, spectral
angle, SSIM and the nan-aware error / variance statistics in one call on tensors already in HBM."""
from math import exp

import numpy as np
import torch

from ... import engine as E
from ... import hip_backend as hb


class Metric(object):
    """Base class for all metrics (metrics.py:10-18)."""

    def reset(self): pass
    def add(self): pass
    def value(self): pass


def gaussian(window_size, sigma):
    """util/pytorch_ssim/__init__.py:7-9"""
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):
    """util/pytorch_ssim/__init__.py:11-15 (the same 2-D window for every channel)."""
    _1D_window = gaussian(window_size, 1.5).unsqueeze(1)
    _2D_window = _1D_window.mm(_1D_window.t()).float().unsqueeze(0).unsqueeze(0)
    return _2D_window.expand(channel, 1, window_size, window_size).contiguous()


_WIN = {}


def _run(target, pred, var, want_pixelwise):
    if not (target.is_cuda and pred.is_cuda):
        raise RuntimeError("uncrtaints_amd metrics run on the GPU only (HIP kernels)")
    if target.dim() != 4 or target.shape != pred.shape:
        raise ValueError("img_metrics expects target and pred of the same [B, C, H, W] shape")
    t, p = target.contiguous().float(), pred.contiguous().float()
    v = None
    if var is not None:
        # the kernels index var with target's [B][C][P] offsets: anything else would read out of bounds
        if not var.is_cuda:
            raise RuntimeError("uncrtaints_amd metrics run on the GPU only (HIP kernels)")
        if var.dim() == 5 and var.shape[1] == 1:          # [B,1,C,H,W] as returned by the MGNLL loss
            var = var[:, 0]
        if var.dim() == 4 and var.shape[1] == 1 and target.shape[1] > 1 and var.shape[0] == target.shape[0] \
                and var.shape[2:] == target.shape[2:]:
            var = var.expand_as(target)                   # isotropic variance: one channel for all bands
        if tuple(var.shape) != tuple(target.shape) or var.numel() == 0:
            raise ValueError(f"img_metrics: var {tuple(var.shape)} does not match target {tuple(target.shape)}")
        v = var.contiguous().float()
    B, C, H, W = t.shape
    dev = t.device
    win = _WIN.get(dev)
    if win is None:
        win = _WIN[dev] = create_window(11, 1).reshape(121).to(dev)
    out = torch.empty(16 + B, device=dev, dtype=torch.float32)
    pix = torch.empty(4, H * W, device=dev, dtype=torch.float32) if want_pixelwise else None
    work = torch.empty(hb.query("uncr_img_metrics_work", B, C, H, W), device=dev, dtype=torch.float32)
    hb.call("uncr_img_metrics", t, p, v, win, out, pix, work, B, C, H, W, E._stream())
    return out, pix


def ssim(img1, img2, window_size=11, size_average=True):
    """util/pytorch_ssim/__init__.py:65-73 (window_size 11 only)."""
    if window_size != 11:
        raise NotImplementedError("the HIP SSIM kernel is built for the reference's 11x11 window")
    out, _ = _run(img1, img2, None, False)
    return out[4] if size_average else out[16:16 + img1.shape[0]]


def img_metrics(target, pred, var=None, pixelwise=True):
    """metrics.py:20-63"""
    out, pix = _run(target, pred, var, var is not None and pixelwise)
    o = out.cpu().numpy()
    metric_dict = {'RMSE': o[0].item(), 'MAE': o[1].item(), 'PSNR': o[2].item(), 'SAM': o[3].item(), 'SSIM': o[4].item()}
    if var is not None:
        errvar_samplewise = {'error': o[5].item(), 'mean ae': o[6].item(), 'mean se': o[7].item(), 'mean var': o[8].item()}
        if pixelwise:
            pw = pix.cpu().numpy()
            errvar_samplewise = {**errvar_samplewise, **{'pixelwise error': pw[0], 'pixelwise ae': pw[1],
                                                        'pixelwise se': pw[2], 'pixelwise var': pw[3]}}
        metric_dict = {**metric_dict, **errvar_samplewise}
    return metric_dict


class avg_img_metrics(Metric):
    """Running nan-skipping means of the scalar metrics (metrics.py:65-104; host bookkeeping only)."""

    def __init__(self):
        super().__init__()
        self.n_samples = 0
        self.metrics = ['RMSE', 'MAE', 'PSNR', 'SAM', 'SSIM']
        self.metrics += ['error', 'mean se', 'mean ae', 'mean var']
        self.running_img_metrics = {}
        self.running_nonan_count = {}
        self.reset()

    def reset(self):
        for metric in self.metrics:
            self.running_nonan_count[metric] = 0
            self.running_img_metrics[metric] = np.nan

    def add(self, metrics_dict):
        for key, val in metrics_dict.items():
            if key not in self.metrics:
                continue
            if torch.is_tensor(val):
                continue
            if isinstance(val, tuple):
                val = val[0]
            if np.isnan(val):
                continue
            if not self.running_nonan_count[key]:
                self.running_nonan_count[key] = 1
                self.running_img_metrics[key] = val
            else:
                self.running_nonan_count[key] += 1
                n = self.running_nonan_count[key]
                self.running_img_metrics[key] = (n - 1) / n * self.running_img_metrics[key] + 1 / n * val

    def value(self):
        return self.running_img_metrics
