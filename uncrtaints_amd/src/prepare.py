"""Input assembly in front of the hot path, on the device: `prepare_data_multi`
(model/train_reconstruct.py:161-179) and the loader's `process_MS` / `process_SAR` (data/dataLoader.py:38-61) as ONE
gather kernel (uncr_assemble_input): the per-date S1 / S2 tensors are read once and written once into
x [B,T,C,H,W], clipped and rescaled on the way when raw intensities are passed."""
import torch

from .. import engine as E
from .. import hip_backend as hb

_KIND = {None: 0, "none": 0, ("ms", "default"): 1, ("ms", "resnet"): 2, ("sar", "default"): 3, ("sar", "resnet"): 4}


def recursive_todevice(x, device):
    """train_reconstruct.py:533-539"""
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, dict):
        return {k: recursive_todevice(v, device) for k, v in x.items()}
    return [recursive_todevice(c, device) for c in x]


def _assemble(groups, B, T, C, H, W, device):
    """groups: list (per channel group) of (list over dates of [B,Cg,H,W] tensors, channel offset, kind)."""
    x = torch.empty(B, T, C, H, W, device=device, dtype=torch.float32)
    rows, keep = [], []
    for t in range(T):
        for tensors, c0, kind in groups:
            src = tensors[t].to(device=device, dtype=torch.float32).contiguous()
            keep.append(src)
            rows.append([src.data_ptr(), src.shape[1], c0, kind])
    desc = torch.tensor(rows, dtype=torch.int64).to(device)
    hb.call("uncr_assemble_input", desc, x, B, T, C, H * W, len(groups), E._stream())
    for s in keep:
        s.record_stream(torch.cuda.current_stream())
    return x


def process_MS(img: torch.Tensor, method: str = "default") -> torch.Tensor:
    """data/dataLoader.py:38-48 on a device tensor [13,H,W] (or [B,13,H,W])."""
    v = img if img.dim() == 4 else img.unsqueeze(0)
    B, C, H, W = v.shape
    out = _assemble([([v], 0, _KIND[("ms", method)] if method in ("default", "resnet") else 0)], B, 1, C, H, W, v.device)
    return out[:, 0] if img.dim() == 4 else out[0, 0]


def process_SAR(img: torch.Tensor, method: str = "default") -> torch.Tensor:
    """data/dataLoader.py:50-61 on a device tensor [2,H,W] (or [B,2,H,W])."""
    v = img if img.dim() == 4 else img.unsqueeze(0)
    B, C, H, W = v.shape
    out = _assemble([([v], 0, _KIND[("sar", method)] if method in ("default", "resnet") else 0)], B, 1, C, H, W, v.device)
    return out[:, 0] if img.dim() == 4 else out[0, 0]


def prepare_data_multi(batch, device, config, process=None):
    """train_reconstruct.py:161-179.  `process`: None (the loader already rescaled, as in the reference) or
    'default' / 'resnet' to apply process_MS / process_SAR to raw intensities inside the same pass."""
    in_S2 = batch['input']['S2']
    in_S2_td = recursive_todevice(batch['input']['S2 TD'], device)
    if config.batch_size > 1:
        in_S2_td = torch.stack(list(in_S2_td)).T
    masks = batch['input']['masks']
    T, (B, _, H, W) = len(in_S2), in_S2[0].shape
    in_m = _assemble([([m.unsqueeze(1) for m in masks], 0, 0)], B, T, 1, H, W, device)[:, :, 0]
    y = torch.cat(recursive_todevice(batch['target']['S2'], device), dim=0).unsqueeze(1)
    k_ms = _KIND[("ms", process)] if process else 0
    k_sar = _KIND[("sar", process)] if process else 0
    if config.use_sar:
        in_S1 = batch['input']['S1']
        in_S1_td = recursive_todevice(batch['input']['S1 TD'], device)
        if config.batch_size > 1:
            in_S1_td = torch.stack(list(in_S1_td)).T
        x = _assemble([(in_S1, 0, k_sar), (in_S2, 2, k_ms)], B, T, 15, H, W, device)
        dates = torch.stack((torch.as_tensor(in_S1_td), torch.as_tensor(in_S2_td))).float().mean(dim=0).to(device)
    else:
        x = _assemble([(in_S2, 0, k_ms)], B, T, 13, H, W, device)
        dates = torch.as_tensor(in_S2_td).float().to(device)
    return x, y, in_m, dates
