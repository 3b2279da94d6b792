import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def proj_vec(n, seed=1234):
    return np.random.default_rng(seed).standard_normal(n).astype(np.float64)


def checksum(a):
    """Same definition as tests/golden/make_golden.py::checksum."""
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), float(a @ proj_vec(a.size))])


def rel_err(a, b):
    """max|a-b| / max|b| -- the metric of BASELINE.json / SURVEY 8(d)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / denom


def compare_param_grads(got, golden, tol=2e-4, zero_ratio=1e-5):
    """Compare parameter gradients `got` (name -> ndarray) with a golden fixture holding `grad/<name>`
    (full tensors) and/or `gradsum/<name>` (checksums).  A few gradients are mathematically zero
    (date-independent shifts ahead of the temporal softmax; a per-channel shift ahead of a bias-free
    1x1 conv + batch-stat BatchNorm): there both sides are pure round-off, detected by the golden value
    being < zero_ratio x its sibling `.weight` gradient, and only smallness is required.
    Returns the list of (name, error) pairs checked."""
    report = []
    for k in golden.files:
        if not (k.startswith("grad/") or k.startswith("gradsum/")):
            continue
        kind, name = k.split("/", 1)
        gv = np.asarray(got[name], dtype=np.float64)
        sib = name.replace(".bias", ".weight")
        sib_scale = float(np.abs(np.asarray(got[sib])).max()) if sib in got else float(np.abs(gv).max())
        ref = golden[k]
        ref_abs = float(np.abs(ref).max()) if kind == "grad" else float(ref[1]) / max(gv.size, 1)
        if name.endswith(".bias") and ref_abs < zero_ratio * sib_scale:
            assert float(np.abs(gv).max()) < 1e-3 * sib_scale, (name, "expected ~0")
            report.append((name, 0.0))
            continue
        if kind == "grad":
            err = rel_err(gv, ref)
        else:
            err = abs(checksum(gv)[1] - ref[1]) / max(abs(ref[1]), 1e-30)
        print(f"[parity] grad {kind} {name}: {err:.3e}")
        assert err < tol, (name, kind, err)
        report.append((name, err))
    return report


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def residual_state(g):
    """state_dict of the g13_residual fixture: stored tensors + the dense 3x3 weights regenerated from the numpy seed
    (the fixture keeps them out to stay small; same generator and order as tests/golden/make_golden.py::case_residual)."""
    import torch
    state = {k[len("state/"):]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("state/")}
    rng = np.random.default_rng(int(g["regen_seed"]))
    shapes = {}
    for k in state:
        if k.endswith(".conv.1.weight") and ".conv1." in k or ".conv2." in k and k.endswith(".conv.1.weight") \
                or ".conv3." in k and k.endswith(".conv.1.weight"):
            shapes[k.replace(".conv.1.weight", ".conv.0.weight")] = (state[k].shape[0], state[k].shape[0], 3, 3)
    for k in [str(s) for s in g["regenerated"]]:
        state[k] = torch.from_numpy((rng.standard_normal(shapes[k]) * 0.03).astype(np.float32))
    return state
