"""Random constructor arguments x input shapes against the CPU oracle (GPU box): a sweep for plumbing bugs in combinations no test names.
    python tools/fuzz_configs.py [n_cases] [first_seed]
Per case: eval output, train output, loss and every gradient; prints the cases whose outputs leave 1e-4 or whose gradients leave
max(1e-4, 3 x the CPU fp32 path's distance from fp64)."""
import os
import random
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import rel_err
from gpu_util import dev, inconv_relu_mask, is_zero_grad, oracle_run, pool_branch, value_relu_mask
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

_MAIN = __name__ == "__main__"
WIDE = _MAIN and "--wide" in sys.argv
n_cases = int(sys.argv[1]) if _MAIN and len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if _MAIN and len(sys.argv) > 2 else 0
for a in (sys.argv[3:] if _MAIN else ()):                      # --dev=h2_fwd=0,eval_tail=0 : engine.dev_options for an A/B of one case
    if a.startswith("--dev="):
        from uncrtaints_amd import engine as _E
        _E.dev_options(**{kv.split("=")[0]: bool(int(kv.split("=")[1])) for kv in a[6:].split(",")}).__enter__()
def run_case(case, overrides=()):
    """-> (outside the contract?, refused?): one random case, printed as a line"""
    bad = 0
    refused = False
    rnd = random.Random(case)
    kw = {}
    if rnd.random() < 0.4:
        kw["agg_mode"] = rnd.choice(["att_mean", "mean"])
    if rnd.random() < 0.3:
        kw["encoder_norm"] = rnd.choice(["batch", "instance"])
    if rnd.random() < 0.3:
        kw["decoder_norm"] = rnd.choice(["group", "instance"])
    if rnd.random() < 0.3:
        kw["encoder_widths"] = rnd.choice([[128, 128], [64], [96]])
    if rnd.random() < 0.5:
        kw["decoder_widths"] = rnd.choice([[128], [64, 64], [128, 128, 128], [96, 96]])
    w = kw.get("encoder_widths", [128])[0]
    if kw.get("decoder_widths", [128])[-1] != w:                          # uncrtaints.py asserts encoder_widths[-1] == decoder_widths[-1]
        kw["decoder_widths"] = [w] * rnd.choice([1, 2])
    if rnd.random() < 0.25:
        kw["covmode"], kw["out_conv"] = "iso", [14]
    if rnd.random() < 0.2:
        kw["separate_out"] = True
    if rnd.random() < 0.2:
        kw["use_v"] = True
    if rnd.random() < 0.15:
        kw["n_head"], kw["d_k"] = rnd.choice([(8, 4), (4, 8), (32, 4)])
    if rnd.random() < 0.15:
        kw["scale_by"] = 10.0
    if rnd.random() < 0.15:
        kw["out_nonlin_var"] = "elu"
    if WIDE:                                   # second sweep: more of the constructor surface
        if rnd.random() < 0.15:
            kw["block_type"] = "residual"
            kw["decoder_widths"] = [kw.get("encoder_widths", [128])[0]] * rnd.choice([1, 2])
        if rnd.random() < 0.15:
            kw["positional_encoding"] = False
        if rnd.random() < 0.15:
            kw["out_nonlin_mean"] = False
        if rnd.random() < 0.2 and "n_head" not in kw:
            kw["d_model"] = rnd.choice([128, 512]) if not kw.get("use_v") else 128
        if rnd.random() < 0.2 and "encoder_widths" not in kw:
            wdt = rnd.choice([32, 192, 256])
            kw["encoder_widths"] = [wdt]
            kw["decoder_widths"] = [wdt] * rnd.choice([1, 2])
    mono = rnd.random() < 0.1 and not kw.get("use_v")
    if mono:
        kw["is_mono"] = True
        kw.pop("agg_mode", None)
    B = rnd.choice([1, 2, 3])
    T = 1 if mono else rnd.choice([1, 2, 3, 5] + ([8] if WIDE else []))
    H, W = rnd.choice([(64, 64), (96, 96), (33, 47), (50, 46), (72, 60), (40, 100), (128, 32), (37, 37)]
                      + ([(65, 33), (32, 32), (64, 128), (97, 129), (34, 257)] if WIDE else [(32, 64)]))
    for a in overrides:                     # --kw="..." / --shape=B,T,H,W / --pad=0|1 override what the seed drew (A/B runs of one case)
        if a.startswith("--kw="):
            kw = eval("dict(" + a[5:] + ")")
        if a.startswith("--shape="):
            B, T, H, W = (int(v) for v in a[8:].split(","))
    tag = f"case {case}: {kw} B={B} T={T} {H}x{W}"
    try:
        okw = {k: v for k, v in kw.items()}
        cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **okw)
        x, y, dates = orc.synthetic_batch(B, T, H, W, seed=100 + case)
        if T > 1 and rnd.random() < 0.5:
            x[B - 1, T - 1] = 0.0
        torch.manual_seed(case)
        mk = dict(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
        mk.update(kw)
        if kw.get("block_type") == "residual":
            raise NotImplementedError("(fuzz) residual blocks need their ReLU masks pinned: covered by tests/test_variants.py, test_anysize.py")
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
        if kw.get("use_v"):
            m.temporal_encoder.dropout.p = 0.0
            m.temporal_encoder.keep_relu_branch = True
        m = m.to("cuda").eval()
        m.keep_boundaries = True
        with torch.no_grad():
            oe = m(dev(x), batch_positions=dev(dates))
            re_ = orc.forward({k: v.clone() for k, v in state.items()}, x, dates, cfg, training=False)
        e_eval = rel_err(oe.cpu().numpy(), re_.numpy())
        m.train()
        out = m(dev(x), batch_positions=dev(dates))
        cov = kw.get("covmode", "diag")
        l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode=cov)(out[:, :, :13], dev(y), out[:, :, 13:m.vars_idx])
        l.backward()
        pidx = None if mono else pool_branch(m, state, x, dates, cfg)[0]
        vm = value_relu_mask(m) if kw.get("use_v") else None
        if "--pin-inconv" in sys.argv:        # in_conv's ReLU on the branch the HIP forward took (gpu_util.inconv_relu_mask)
            im, _ = inconv_relu_mask(m, state, x, dates, cfg)
            vm = {**(vm or {}), **im}
        ot, lo, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx, relu_masks=vm)
        _, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx, relu_masks=vm)
        e_train = rel_err(out.detach().cpu().numpy(), ot.numpy())
        worst = (0.0, 0.0, "")
        viol = []
        for k, v in m.named_parameters():
            if v.grad is None or g64.get(k) is None or is_zero_grad(k, g64) or float(g64[k].abs().max()) < 1e-7:
                continue
            eh, ec = rel_err(v.grad.cpu().numpy(), g64[k].numpy()), rel_err(g32[k].numpy(), g64[k].numpy())
            if eh > worst[0]:
                worst = (eh, ec, k)
            if eh > max(1e-4, 3 * ec):
                viol.append((eh, ec, k))
        flag = "" if (e_eval < 1e-4 and e_train < 1e-4 and not viol and abs(l.item() - lo.item()) < 1e-4 * abs(lo.item())) else "  <<<<<<"
        bad += bool(flag)
        print(f"{tag}: eval {e_eval:.1e} train {e_train:.1e} worst grad {worst[0]:.1e} (cpu {worst[1]:.1e}) {worst[2]}{flag}", flush=True)
        for eh, ec, k in sorted(viol, reverse=True)[:10]:
            print(f"      {eh:.2e} cpu {ec:.2e} {k}")
    except NotImplementedError as exc:
        refused = True
        print(f"{tag}: refused -- {str(exc)[:110]}", flush=True)
    except Exception as exc:
        bad += 1
        print(f"{tag}: {type(exc).__name__}: {str(exc)[:200]}  <<<<<<", flush=True)
        traceback.print_exc(limit=3)
    return bool(bad), refused


if __name__ == "__main__":
    total = sum(run_case(c, sys.argv[3:])[0] for c in range(first, first + n_cases))
    print("cases outside the contract:", total)
