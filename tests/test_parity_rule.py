"""The gradient parity rule itself (tests/gpu_util.py::close_grad).

CPU part: a guard -- the contract's tolerance and the second clause's allowed multiple cannot be raised without this file failing,
and the rule accepts / rejects what its docstring says.
GPU part: the statistical statement behind the second clause -- over fresh inputs the HIP path's gradient noise is distributed like
the CPU fp32 path's (profiles/r05_parity_attribution.json)."""
import inspect
import re

import numpy as np
import pytest
import torch

import gpu_util
from gpu_util import Fp32Draws, close_grad


def test_tolerance_and_noise_multiple_are_not_raised():
    """SURVEY 8(d) / BASELINE.json: 1e-4 relative, per output tensor and per gradient.  The second clause (fp64 tie-breaker for
    cancellation-dominated sums) allows at most 3 x the CPU fp32 evaluations' own distance from the fp64 truth (round 4: 8)."""
    assert gpu_util.TOL == 1e-4
    assert gpu_util.NOISE <= 3.0
    sig = inspect.signature(close_grad)
    assert sig.parameters["tol"].default == gpu_util.TOL and sig.parameters["noise"].default == gpu_util.NOISE
    # no call site overrides them
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for f in os.listdir(here):
        if f.endswith(".py") and f != "test_parity_rule.py":
            src = open(os.path.join(here, f)).read()
            for m in re.finditer(r"close_grad\(([^\n]*(?:\n[^\n]*){0,2}?)\)\n", src):
                assert "noise=" not in m.group(1) and "tol=" not in m.group(1), (f, m.group(0))


def test_the_set_of_fp32_draws_is_frozen():
    """close_grad's scale is the LARGEST distance among the fp32 evaluations at hand, so every further evaluation widens the allowance
    monotonically.  The set is therefore fixed: this host's first fp32 run (`ref32`), two more by `Fp32Draws.get` (spelled-out
    formulas, one thread) and at most two pre-existing ones in `extra` (the fixture's golden gradient = the reference host's run, and
    the first fp32 run itself); nothing else, at no call site."""
    import ast
    import os
    assert Fp32Draws.N_RUNS == 2 and Fp32Draws.MAX_EXTRA == 2
    src = inspect.getsource(Fp32Draws.get)
    assert src.count("self.run()") == Fp32Draws.N_RUNS, "Fp32Draws.get performs exactly two further evaluations"
    with pytest.raises(AssertionError):
        Fp32Draws(lambda: {}, extra=[{}, {}, {}])
    # close_grad reads nothing but ref32 and draws.get(key) into its scale
    csrc = inspect.getsource(close_grad)
    assert csrc.count("e_cpu = max(") == 1 and "draws.get(key)" in csrc
    here = os.path.dirname(os.path.abspath(__file__))
    sites = 0
    for f in sorted(os.listdir(here)):
        if not f.endswith(".py") or f in ("test_parity_rule.py", "gpu_util.py"):
            continue
        tree = ast.parse(open(os.path.join(here, f)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and getattr(node.func, "id", getattr(node.func, "attr", None)) == "Fp32Draws":
                sites += 1
                assert len(node.args) == 1, (f, node.lineno)
                for kw in node.keywords:
                    assert kw.arg == "extra", (f, node.lineno, kw.arg)
                    # `extra` is a literal list of at most two entries (optionally `[...] if cond else []`)
                    val = kw.value.body if isinstance(kw.value, ast.IfExp) else kw.value
                    assert isinstance(val, ast.List) and len(val.elts) <= Fp32Draws.MAX_EXTRA, (f, node.lineno)
    assert sites >= 5


def _t(v):
    return torch.tensor(v, dtype=torch.float64)


def test_rule_accepts_and_rejects():
    truth = _t([1.0, -2.0, 4.0])
    ref32 = truth + _t([0.0, 0.0, 4e-5])          # a CPU fp32 evaluation 1e-5 (relative to max) from the truth
    # clause 1: within 1e-4 of the fp32 reference
    close_grad("c1", (ref32 + _t([3e-4, 0, 0])).float(), ref32.float(), truth)
    # clause 2 does not apply below tol: up to 1e-4 of the truth passes whatever the CPU distance
    close_grad("c2a", truth + _t([0, 0, 3.9e-4]), truth + _t([0, 0, -3.9e-4]), truth)
    # clause 2: 3 x the CPU distance when that exceeds tol
    ref_far = truth + _t([0, 0, 4e-4])            # 1e-4 from the truth
    close_grad("c2b", truth + _t([0, 0, -1.19e-3]), ref_far, truth)      # 2.98e-4 <= 3 x 1e-4
    with pytest.raises(AssertionError):
        close_grad("c2c", truth + _t([0, 0, -1.3e-3]), ref_far, truth)    # 3.25e-4 >  3 x 1e-4
    with pytest.raises(AssertionError):
        close_grad("nan", _t([1.0, float("nan"), 4.0]), ref32, truth)


def test_further_fp32_evaluations_widen_the_scale_lazily():
    truth = _t([1.0, -2.0, 4.0])
    ref32 = truth + _t([0, 0, 8e-5])              # 2e-5 from the truth
    got = truth + _t([0, 0, -6e-4])               # 1.5e-4: fails against max(1e-4, 3 x 2e-5) ...
    calls = []

    def run():
        calls.append(1)
        return {"w": (truth + _t([0, 0, 2.4e-4])).float()}   # ... another correct evaluation sits 6e-5 away: 3 x 6e-5 = 1.8e-4 admits it

    d = Fp32Draws(run)
    with pytest.raises(AssertionError):
        close_grad("single", got, ref32, truth)
    close_grad("draws", got, ref32, truth, draws=d, key="w")
    assert len(calls) == 2                        # spelled-out formulas + one thread, evaluated once
    close_grad("draws again", got, ref32, truth, draws=d, key="w")
    assert len(calls) == 2
    lazy = Fp32Draws(lambda: calls.append(2) or {})
    close_grad("no need", ref32, ref32, truth, draws=lazy, key="w")
    assert 2 not in calls                         # a gradient inside clause 1 never pays for further evaluations
    from oracle import uncrtaints_oracle as orc
    assert orc.USE_ATEN is True


@pytest.mark.gpu
def test_gradient_noise_is_distributed_like_the_cpu_paths_over_fresh_inputs():
    """The ill-conditioned `weight_init` weights of the g1 fixture on four fresh inputs: per input the LARGEST gradient distance
    from the fp64 oracle, HIP path vs the CPU fp32 oracle.  One input is a lottery ticket (close_grad's docstring); the medians are
    not: the HIP path's stays under the contract's 1e-4 and within 2 x the CPU path's."""
    import json
    from conftest import load_golden, rel_err
    from gpu_util import dev, is_zero_grad, oracle_run, pool_branch
    from oracle import uncrtaints_oracle as orc
    from uncrtaints_amd.src import losses
    from uncrtaints_amd.src.backbones import uncrtaints as U
    g = load_golden("g1_diag_t3")
    state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    cfg = orc.OracleConfig(attn_dropout=0.0)
    hip, cpu = [], []
    for seed in (11, 12, 13, 14):
        x, y, dates = orc.synthetic_batch(2, 3, 64, 64, seed=seed)
        m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
        m.load_state_dict(state, strict=True)
        m.temporal_aggregator.attn_dropout.p = 0.0
        m = m.to("cuda").train()
        out = m(dev(x), batch_positions=dev(dates))
        l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
        l.backward()
        pidx, _ = pool_branch(m, state, x, dates, cfg)
        _, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx)
        _, _, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx)
        keys = [k for k, _ in m.named_parameters() if not is_zero_grad(k, g64)]
        grads = dict(m.named_parameters())
        hip.append(max(rel_err(grads[k].grad.double().cpu().numpy(), g64[k].numpy()) for k in keys))
        cpu.append(max(rel_err(g32[k].double().numpy(), g64[k].numpy()) for k in keys))
        print(f"[parity] noise distribution seed {seed}: worst gradient distance from fp64: hip {hip[-1]:.3e} cpu-fp32 {cpu[-1]:.3e}")
    mh, mc = float(np.median(hip)), float(np.median(cpu))
    print(f"[parity] noise distribution medians: hip {mh:.3e} cpu-fp32 {mc:.3e}")
    assert mh <= gpu_util.TOL and mh <= 2.0 * mc, (hip, cpu)
