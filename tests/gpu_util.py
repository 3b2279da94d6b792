"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch

from conftest import rel_err

DEV = "cuda"
TOL = 1e-4   # BASELINE.json: max|a-b| / max|b| <= 1e-4, fp32


def dev(t):
    return t.to(DEV).contiguous()


def close(name, got, ref, tol=TOL, report=None):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().float().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{name}: non-finite values"
    e = rel_err(got, ref)
    if report is not None:
        report.append((name, e))
    print(f"[parity] {name}: rel_err={e:.3e}")
    assert e < tol, f"{name}: rel_err {e:.3e} >= {tol}"
    return e


def rand(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale + shift
