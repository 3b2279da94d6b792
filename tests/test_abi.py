"""CPU-only: the C-ABI library builds/loads and exports every symbol include/uncr_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os

import pytest

from uncrtaints_amd import hip_backend as hb


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(hb.LIB_PATH):
        from uncrtaints_amd.build import build_lib
        build_lib()
    return hb.LIB_PATH


def test_header_parses_and_has_all_stages():
    protos = hb.parse_header()
    for name in ("uncr_norm_finalize_fwd", "uncr_norm_finalize_bwd", "uncr_ew", "uncr_pw_gemm", "uncr_pw_wgrad",
                 "uncr_dw_fwd", "uncr_dw_bwd", "uncr_se_mlp_fwd", "uncr_se_mlp_bwd", "uncr_maxpool_fwd",
                 "uncr_ltae_gn_fwd", "uncr_ltae_softmax_fwd", "uncr_aggregate_fwd", "uncr_aggregate_bwd",
                 "uncr_mgnll_fwd", "uncr_mgnll_bwd", "uncr_ensemble_combine", "uncr_pad_mask", "uncr_version"):
        assert name in protos, name
    # every pointer/size signature is plain C: no torch types
    for name, args in protos.items():
        for typ, _ in args:
            assert typ.replace("const ", "").replace("*", "").strip() in (
                "float", "double", "int", "unsigned long long", "long long", "hipStream_t", "void"), (name, typ)


def test_library_exports_every_declared_symbol(built):
    cdll = ctypes.CDLL(built)
    for name in hb.parse_header():
        assert hasattr(cdll, name), f"{name} declared in include/uncr_hip.h but not exported"


def test_library_keeps_no_switches(built):
    """the entry points are re-entrant and hold no mutable behaviour state -- no process-wide setters in the ABI, no
    environment reads inside the library (kernel variants follow from the arguments of each call)."""
    import subprocess
    assert not [n for n in hb.parse_header() if "_set_" in n]
    syms = subprocess.run(["nm", "-D", built], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_python_layer_reads_no_environment_switches():
    """The arithmetic path of the Python layer above the library is fixed as well: engine.py and the module classes never read the
    process environment (development switches are module constants flipped through `engine.dev_options(...)` by tools and tests)."""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uncrtaints_amd")
    files = [os.path.join(root, "engine.py"), os.path.join(root, "optim.py"), os.path.join(root, "parallel.py")] + \
        glob.glob(os.path.join(root, "src", "**", "*.py"), recursive=True)
    for f in files:
        src = open(f).read()
        assert "os.environ" not in src and "getenv" not in src, f
    from uncrtaints_amd import engine
    with engine.dev_options(h2_fwd=False):
        assert engine._H2_FWD is False
    assert engine._H2_FWD is True
    with pytest.raises(KeyError):
        engine.dev_options(no_such_switch=1)


def test_size_queries(built):
    assert hb.query("uncr_version") >= 1
    assert hb.query("uncr_pw_coutp", 26) == 32 and hb.query("uncr_pw_coutp", 128) == 128
    assert hb.query("uncr_pw_coutp", 256) == 256 and hb.query("uncr_pw_coutp", 64) == 64
    assert hb.query("uncr_pw_kpad", 15) == 32 and hb.query("uncr_pw_kpad", 256) == 256
    assert hb.query("uncr_ew_slots", 65536) == 64
    assert hb.query("uncr_dw_slots_fwd", 256) == 8 and hb.query("uncr_dw_slots_bwd", 256) == 16


def test_product_path_refuses_cpu_tensors(built):
    import torch
    from uncrtaints_amd.src.backbones import uncrtaints
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus")
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 3, 15, 64, 64), batch_positions=torch.tensor([[1., 2., 3.]]))


def test_state_dict_keys_match_reference_fixture():
    import torch
    from conftest import load_golden
    from uncrtaints_amd.src.backbones import uncrtaints
    g = load_golden("g1_diag_t3")
    sd = {k[len("state/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus")
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert sum(p.numel() for p in m.parameters()) == 570010


def test_split_prediction_backward_paths():
    """losses.split_prediction (pure host logic): two gradients that are the channel slices of ONE buffer come back as that buffer
    without a copy; anything else is assembled; a missing gradient counts as zeros."""
    import torch
    from uncrtaints_amd.src import losses
    out = torch.randn(2, 1, 26, 4, 8, requires_grad=True)
    mid = out * 1.0                                     # a non-leaf, like the head's output
    seen = []
    mid.register_hook(lambda g: seen.append(g))
    m, v = losses.split_prediction(mid, 13, 26)
    assert m.shape == (2, 1, 13, 4, 8) and v.shape == (2, 1, 13, 4, 8) and m.data_ptr() == mid.data_ptr()
    full = torch.randn(2, 1, 26, 4, 8)
    torch.autograd.backward([m, v], [full[:, :, :13], full[:, :, 13:]])
    assert seen[0].data_ptr() == full.data_ptr() and torch.equal(out.grad, full)          # handed on, not copied
    out.grad = None
    m, v = losses.split_prediction(out, 13, 26)
    gm, gv = torch.randn_like(m), torch.randn_like(v)
    torch.autograd.backward([m, v], [gm, gv])
    assert torch.equal(out.grad[:, :, :13], gm) and torch.equal(out.grad[:, :, 13:], gv)
    out.grad = None
    m, v = losses.split_prediction(out, 13, 26)
    (v * 3.0).sum().backward()
    assert float(out.grad[:, :, :13].abs().max()) == 0.0 and torch.equal(out.grad[:, :, 13:], torch.full_like(v, 3.0))
