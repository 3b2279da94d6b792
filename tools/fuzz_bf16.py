"""bf16 activation storage on random supported configurations (GPU box): finite results, loss within 3 % of the fp32 HIP path, every
gradient's direction within cos > 0.98 of the fp32 path's.  A sanity sweep for crashes / NaNs / gross errors -- the contract itself is
tests/test_bf16.py.     python tools/fuzz_bf16.py [n] [first]"""
import os, random, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(first, first + n_cases):
    rnd = random.Random(case)
    kw = {}
    if rnd.random() < 0.4:
        kw["agg_mode"] = rnd.choice(["att_mean", "mean"])
    if rnd.random() < 0.3:
        kw["encoder_norm"] = rnd.choice(["batch", "instance"])
    if rnd.random() < 0.3:
        kw["decoder_norm"] = rnd.choice(["group", "instance"])
    if rnd.random() < 0.3:
        w = rnd.choice([64, 96, 256])
        kw["encoder_widths"], kw["decoder_widths"] = [w], [w] * rnd.choice([1, 2])
    elif rnd.random() < 0.4:
        kw["decoder_widths"] = [128] * rnd.choice([1, 2, 3])
    if rnd.random() < 0.25:
        kw["covmode"], kw["out_conv"] = "iso", [14]
    if rnd.random() < 0.2:
        kw["separate_out"] = True
    if rnd.random() < 0.15:
        kw["n_head"], kw["d_k"] = rnd.choice([(8, 4), (32, 4)])
    mono = rnd.random() < 0.1
    if mono:
        kw["is_mono"] = True
        kw.pop("agg_mode", None)
    B, T = rnd.choice([1, 2, 4]), (1 if mono else rnd.choice([1, 2, 3, 6]))
    H, W = rnd.choice([(64, 64), (96, 96), (128, 64), (64, 128), (32, 32), (256, 256), (128, 128)])
    tag = f"case {case}: {kw} B={B} T={T} {H}x{W}"
    try:
        x, y, dates = orc.synthetic_batch(B, T, H, W, seed=500 + case)
        if T > 1 and rnd.random() < 0.4:
            x[B - 1, T - 1] = 0.0
        mk = dict(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0)
        mk.update(kw)
        res = {}
        for mode in ("fp32", "bf16"):
            torch.manual_seed(case)
            m = U.UNCRTAINTS(**mk)
            if hasattr(m, "temporal_aggregator"):
                m.temporal_aggregator.attn_dropout.p = 0.0
            m = m.cuda().train()
            if mode == "bf16":
                m.set_act_dtype("bf16")
            out = m(x.cuda(), batch_positions=dates.cuda())
            cov = kw.get("covmode", "diag")
            l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode=cov)(out[:, :, :13], y.cuda(), out[:, :, 13:m.vars_idx])
            l.backward()
            res[mode] = (float(l), out.detach().float(), {k: v.grad.detach().float() for k, v in m.named_parameters() if v.grad is not None})
        l32, o32, g32 = res["fp32"]
        l16, o16, g16 = res["bf16"]
        probs = []
        if not (torch.isfinite(o16).all() and all(torch.isfinite(v).all() for v in g16.values())):
            probs.append("non-finite")
        if abs(l16 - l32) > 0.03 * abs(l32):
            probs.append(f"loss {l16:.5f} vs {l32:.5f}")
        eo = float((o16 - o32).abs().max() / o32.abs().max())
        if eo > 0.08:
            probs.append(f"output {eo:.2e}")
        worst = (1.0, "")
        for k, v in g32.items():
            if float(v.abs().max()) < 1e-7 * max(float(t.abs().max()) for t in g32.values()):
                continue
            # a bias in front of a norm that removes the mean per plane / per channel has a zero gradient: both paths hold rounding noise
            if k == "in_conv.conv.conv.0.bias" and kw.get("encoder_norm") in ("instance", "batch"):
                continue
            c = float(torch.nn.functional.cosine_similarity(v.flatten().double(), g16[k].flatten().double(), dim=0))
            if c < worst[0]:
                worst = (c, k)
        if worst[0] < 0.98:
            probs.append(f"gradient direction cos {worst[0]:.3f} {worst[1]}")
        print(f"{tag}: loss {l32:.5f} / {l16:.5f} out {eo:.1e} worst cos {worst[0]:.4f} {worst[1]}" + (f"  {probs}  <<<<<<" if probs else ""), flush=True)
        bad += bool(probs)
    except NotImplementedError as exc:
        print(f"{tag}: refused -- {str(exc)[:110]}", flush=True)
    except Exception as exc:
        bad += 1
        print(f"{tag}: {type(exc).__name__}: {str(exc)[:200]}  <<<<<<", flush=True)
        traceback.print_exc(limit=3)
print("cases outside the sanity bounds:", bad)
