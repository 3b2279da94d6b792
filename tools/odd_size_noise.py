"""Distribution of the worst gradient error of a small odd-size model over input seeds: HIP against the CPU fp32 paths (GPU box).
    python tools/odd_size_noise.py [n_seeds]
Prints, per seed, the largest distance from fp64 among the gradients for HIP and for the CPU fp32 evaluations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import rel_err
from gpu_util import dev, is_zero_grad, oracle_run, pool_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

kw = dict(encoder_norm="batch", encoder_widths=[128, 128])
B, T, H, W = 2, 2, 34, 70
RANDOM_AFFINES = "--affines" in sys.argv
NSHOW = 12 if "--more" in sys.argv else 3
for a in sys.argv[1:]:                      # --kw="use_v=True" --shape=1,2,50,46
    if a.startswith("--kw="):
        kw = eval("dict(" + a[5:] + ")")
    if a.startswith("--shape="):
        B, T, H, W = (int(v) for v in a[8:].split(","))
cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **kw)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    x, y, dates = orc.synthetic_batch(B, T, H, W, seed=7 + seed)
    if "--nopad" not in sys.argv:
        x[B - 1, T - 1] = 0.0
    torch.manual_seed(6 + seed)
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
    if RANDOM_AFFINES:                     # as tests/test_anysize.py::test_model_variants_at_odd_sizes
        g_ = torch.Generator().manual_seed(16)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g_))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g_))
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)) and mod.weight is not None:
                mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g_))
                mod.bias.data.copy_(0.2 * torch.randn(mod.bias.shape, generator=g_))
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.temporal_aggregator.attn_dropout.p = 0.0
    if kw.get("use_v"):
        m.temporal_encoder.dropout.p = 0.0
        m.temporal_encoder.keep_relu_branch = "--free-relu" not in sys.argv
    m = m.to("cuda").train()
    out = m(dev(x), batch_positions=dev(dates))
    l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
    l.backward()
    pidx, _ = pool_branch(m, state, x, dates, cfg)
    vm = None
    if kw.get("use_v") and "--free-relu" not in sys.argv:
        from gpu_util import value_relu_mask
        vm = value_relu_mask(m)
    _, _, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx, relu_masks=vm)
    orc.USE_ATEN = False
    _, _, _, g32b, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx, relu_masks=vm)
    orc.USE_ATEN = True
    _, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx, relu_masks=vm)
    rows = []
    for k, v in m.named_parameters():
        if v.grad is None or g64.get(k) is None or is_zero_grad(k, g64):
            continue
        t = g64[k].numpy()
        rows.append((rel_err(v.grad.cpu().numpy(), t), max(rel_err(g32[k].numpy(), t), rel_err(g32b[k].numpy(), t)), k))
    for a, b, k in sorted(rows, reverse=True)[:NSHOW]:
        print(f"      {a:.2e} cpu {b:.2e} ratio {a / b:.2f} {k}")
    eh, ec, kh = max(rows)
    ec_worst = max(r[1] for r in rows)
    beyond = [(a, b, k) for a, b, k in rows if a > 1e-4]
    print(f"seed {seed}: hip worst {eh:.2e} ({kh}; cpu there {ec:.2e}); cpu worst anywhere {ec_worst:.2e}; hip lines beyond 1e-4: "
          + ", ".join(f"{k} {a:.1e}/{b:.1e}" for a, b, k in beyond), flush=True)
