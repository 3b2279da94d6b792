"""Where does an eval-mode forward lose its digits?  (GPU box)   python tools/debug_eval_taps.py <case> [--wide]
Rebuilds one tools/fuzz_configs.py case, runs the eval forward on the HIP path and on the oracle (fp64 and fp32) and compares the stage
boundaries (a0, encoder output, aggregate, every decoder block fed with the fp64 path's input), with mean/std of the planes beside."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
case = int(sys.argv[1])
import torch

import fuzz_configs as F        # its case generator is run_case's first half: re-derived here through the same seed
from conftest import rel_err
from gpu_util import dev
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E_
from uncrtaints_amd.src.backbones import uncrtaints as U


def draw(case, wide):
    """the constructor arguments / shapes tools/fuzz_configs.py draws for this seed (same order of draws)"""
    F.WIDE = wide
    got = {}
    real = U.UNCRTAINTS

    class Stop(Exception):
        pass

    def grab(**mk):
        got["mk"] = mk
        raise Stop
    U.UNCRTAINTS = grab
    sb = orc.synthetic_batch

    def grab_batch(B, T, H, W, seed):
        got["shape"] = (B, T, H, W)
        r = sb(B, T, H, W, seed=seed)
        got["xyz"] = r
        return r
    orc.synthetic_batch = grab_batch
    cfgc = orc.OracleConfig

    def grab_cfg(**kw):
        got["cfg"] = cfgc(**kw)
        return got["cfg"]
    orc.OracleConfig = grab_cfg
    try:
        F.run_case(case)
    finally:
        U.UNCRTAINTS, orc.synthetic_batch, orc.OracleConfig = real, sb, cfgc
    return got


g = draw(case, "--wide" in sys.argv)
mk, cfg, (x, y, dates) = g["mk"], g["cfg"], g["xyz"]
print("case", case, {k: v for k, v in mk.items()}, g["shape"])
torch.manual_seed(case)
m = U.UNCRTAINTS(**mk)
g_ = torch.Generator().manual_seed(1000 + case)
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g_))
        mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g_))
    if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)) and mod.weight is not None:
        mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g_))
        mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g_))
state = {k: v.detach().clone() for k, v in m.state_dict().items()}
if hasattr(m, "temporal_aggregator"):
    m.temporal_aggregator.attn_dropout.p = 0.0
m = m.to("cuda").eval()
m.keep_boundaries = True
t64, t32 = {}, {}
with torch.no_grad():
    oe = m(dev(x), batch_positions=dev(dates))
    r64 = orc.forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}, x.double(), dates.double(), cfg,
                      training=False, taps=t64)
    r32 = orc.forward({k: v.clone() for k, v in state.items()}, x, dates, cfg, training=False, taps=t32)


def line(name, h, k):
    a, b, c = h.float().cpu().reshape(t64[k].shape).numpy(), t32[k].numpy(), t64[k].numpy()
    pl = t64[k].reshape(-1, t64[k].shape[-2] * t64[k].shape[-1])
    ratio = (pl.mean(1).abs() / pl.std(1).clamp_min(1e-30))
    print(f"{name:10s} hip {rel_err(a, c):.2e}  cpu32 {rel_err(b, c):.2e}   |mean|/std of planes: median {ratio.median():.1f} max {ratio.max():.1f}")


B, T = g["shape"][:2]
line("a0", m._boundary_a0, "a0")
if not mk.get("is_mono"):
    line("enc", m._boundary_enc, "e")
    line("agg", m._boundary_agg, "agg")
print(f"{'out':10s} hip {rel_err(oe.cpu().numpy(), r64.numpy()):.2e}  cpu32 {rel_err(r32.numpy(), r64.numpy()):.2e}")
with torch.no_grad():
    prev = t64["agg"] if "agg" in t64 else t64["e"]
    if E_.plan_geom(*g["shape"][2:]) is not None:
        m.out_block = []          # (any-size planes: the blocks take embedded tensors; the stage boundaries above are what this tool reports)
    for i, layer in enumerate(m.out_block):
        o = layer.smart_forward(dev(prev.float().contiguous()))
        k = f"dec{i}"
        a, c = o.float().cpu().numpy(), t64[k].numpy()
        pl = prev.reshape(-1, prev.shape[-2] * prev.shape[-1])
        ratio = pl.mean(1).abs() / pl.std(1).clamp_min(1e-30)
        print(f"block {i} on the fp64 path's input: hip {rel_err(a, c):.2e}  cpu32 {rel_err(t32[k].numpy(), c):.2e}   input planes |mean|/std median "
              f"{ratio.median():.1f} max {ratio.max():.1f}")
        for nm in ("h1", "h2", "h3"):
            tt = t64[f"out_block.{i}.{nm}"]
            pl2 = tt.reshape(-1, tt.shape[-2] * tt.shape[-1])
            r2 = pl2.mean(1).abs() / pl2.std(1).clamp_min(1e-30)
            print(f"      {nm}: |mean|/std median {r2.median():.2f} max {r2.max():.1f}   std min {pl2.std(1).min():.2e} max {pl2.std(1).max():.2e}")
        prev = t64[k]

if "--train" in sys.argv and mk.get("use_v"):
    # the value branch in training: pre-BatchNorm tensor m1 [B, C, S] and the BatchNorm1d coefficients the HIP path derived from its statistics
    m.train()
    m.temporal_encoder.dropout.p = 0.0
    m.temporal_encoder.keep_relu_branch = True
    tt = {}
    ot = m(dev(x), batch_positions=dev(dates))
    r64t = orc.forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}, x.double(), dates.double(), cfg,
                       training=True, taps=tt, update_running=False)
    m1, A, Bc = m.temporal_encoder._last_relu
    b_, c_, s_ = m1.shape
    m1h = m1.float().permute(0, 2, 1).reshape(b_ * s_, c_).cpu().double()
    m1r = tt["val_m1"]
    print(f"train out  hip {rel_err(ot.detach().cpu().numpy(), r64t.numpy()):.2e}")
    print(f"val m1     hip {rel_err(m1h.numpy(), m1r.numpy()):.2e}   per-channel |mean|/std: median {(m1r.mean(0).abs() / m1r.std(0)).median():.1f} "
          f"max {(m1r.mean(0).abs() / m1r.std(0)).max():.1f}")
    p64 = {k: v.double() for k, v in state.items() if v.is_floating_point()}
    gam, bet = p64["temporal_encoder.mlp.1.weight"], p64["temporal_encoder.mlp.1.bias"]
    mu, var = m1r.mean(0), m1r.var(0, unbiased=False)
    A64 = gam / torch.sqrt(var + 1e-5)
    B64 = bet - mu * A64
    Ah, Bh = A.double().cpu().view(-1, c_)[0], Bc.double().cpu().view(-1, c_)[0]
    print(f"BatchNorm1d A: {float(((Ah - A64).abs() / A64.abs()).max()):.2e} relative   B: {float((Bh - B64).abs().max()):.2e} absolute (|B| up to "
          f"{float(B64.abs().max()):.1f})")
    for nm in ("val_y", "val_vh"):
        t_ = tt[nm].reshape(-1, tt[nm].shape[-1])
        print(f"      {nm}: per-channel |mean|/std median {(t_.mean(0).abs() / t_.std(0)).median():.1f} max {(t_.mean(0).abs() / t_.std(0)).max():.1f}")
