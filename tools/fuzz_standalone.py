"""Random shapes / arguments for the stand-alone entry points against the oracle (GPU box): MGNLL (diag / iso, three reductions, any
H x W, values near the clamp), GaussianNLL, and LTAE2d on its own (channels, heads, key width, MLP depth, dates, padding).
    python tools/fuzz_standalone.py [n_cases] [first_seed]"""
import os, random, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import rel_err
from gpu_util import relu_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones.ltae import LTAE2d

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
dev = lambda t: t.cuda()
for case in range(first, first + n_cases):
    rnd = random.Random(case)
    g = torch.Generator().manual_seed(case)
    try:
        # ---- MGNLL
        B, H, W = rnd.choice([1, 2, 3, 5]), rnd.choice([1, 7, 32, 33, 64, 100]), rnd.choice([1, 5, 32, 47, 64, 130])
        mode, red = rnd.choice(["diag", "iso"]), rnd.choice(["mean", "sum", "none"])
        pred, tgt = torch.rand(B, 1, 13, H, W, generator=g), torch.rand(B, 1, 13, H, W, generator=g)
        var = torch.rand(B, 1, 13 if mode == "diag" else 1, H, W, generator=g) * rnd.choice([1.0, 1e-3, 1e-6]) + rnd.choice([0.0, 1e-9, 1e-7])
        if rnd.random() < 0.3:
            var.view(-1)[:: max(1, var.numel() // 17)] = 0.0           # below the clamp
        tag = f"case {case} mgnll {mode} {red} B={B} {H}x{W}"
        outs = {}
        for name, dt_ in (("hip", None), ("o32", torch.float32), ("o64", torch.float64)):
            if dt_ is None:
                p_, v_ = dev(pred).requires_grad_(True), dev(var).requires_grad_(True)
                l, vv = losses.multi_gaussian_nll_loss(p_, dev(tgt), v_, full=True, eps=1e-8, reduction=red, mode=mode)
            else:
                p_, v_ = pred.detach().clone().to(dt_).requires_grad_(True), var.detach().clone().to(dt_).requires_grad_(True)
                l, _ = orc.mgnll(p_, tgt.to(dt_), v_, mode=mode, eps=1e-8, reduction=red)
            gw = torch.ones_like(l) if l.dim() else None
            (l * (gw if gw is not None else 1.0)).sum().backward()
            outs[name] = (l.detach().double().cpu(), p_.grad.double().cpu(), v_.grad.double().cpu())
        errs = []
        for i, what in enumerate(("loss", "dpred", "dvar")):
            t = outs["o64"][i]
            if tuple(outs["hip"][i].shape) != tuple(t.shape):
                errs.append((what, "shape", tuple(outs["hip"][i].shape), tuple(t.shape)))
                continue
            sc = float(t.abs().max()) or 1.0
            eh, ec = float((outs["hip"][i] - t).abs().max()) / sc, float((outs["o32"][i] - t).abs().max()) / sc
            if eh > max(1e-5, 4 * ec):
                errs.append((what, f"{eh:.1e}", f"cpu {ec:.1e}"))
        print(tag + (":  ok" if not errs else f":  {errs}  <<<<<<"), flush=True)
        bad += bool(errs)
        # ---- LTAE2d on its own
        nh = rnd.choice([4, 8, 16])
        C = nh * rnd.choice([4, 8])
        dk = rnd.choice([4, 8])
        dm = rnd.choice([None, 128, 256]) if C in (128,) or True else None
        dm = rnd.choice([None, nh * 8, nh * 16])
        d_in = dm if dm is not None else C
        mlp = [d_in] + rnd.choice([[C], [64], [128, 64], [96, 64, 32]])
        if mlp[-1] % nh:
            mlp[-1] = nh * 4
        Bq, T = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 4, 7])
        tag = f"case {case} LTAE2d C={C} nh={nh} dk={dk} d_model={dm} mlp={mlp} B={Bq} T={T}"
        torch.manual_seed(case)
        m = LTAE2d(in_channels=C, n_head=nh, d_k=dk, mlp=mlp, dropout=0.0, d_model=dm, return_att=True, use_dropout=False,
                   positional_encoding=rnd.random() < 0.8)
        with torch.no_grad():
            for i in range(1, len(m.mlp), 3):
                m.mlp[i].bias.copy_(1.5 + 0.2 * torch.randn(m.mlp[i].bias.shape, generator=g))      # keep the GroupNorm groups alive
        state = {k: v.detach().clone() for k, v in m.state_dict().items()}
        x = torch.randn(Bq, T, C, 32, 32, generator=g)
        dates = torch.sort(torch.randint(1400, 1800, (Bq, T), generator=g), dim=1).values.float()
        pad = torch.zeros(Bq, T, dtype=torch.bool)
        if T > 1 and rnd.random() < 0.5:
            pad[Bq - 1, T - 1] = True
        gv, ga = torch.randn(Bq, mlp[-1], 32, 32, generator=g), torch.randn(nh, Bq, T, 32, 32, generator=g)
        training = rnd.random() < 0.7
        m = m.cuda().train(training)
        m.keep_relu_branch = True
        xh = dev(x).requires_grad_(True)
        o, a = m(xh, batch_positions=dev(dates), pad_mask=dev(pad))
        ((o * dev(gv)).sum() + (a * dev(ga)).sum()).backward()
        # the last ReLU sits in front of a GroupNorm over C/nh values per pixel: differentiate the oracle on the branch the HIP forward took
        # (gpu_util.value_relu_mask: one pre-activation within rounding of zero in a dead group moves every gradient by ~316 / (n*C))
        m1_, A_, B_ = m._last_relu
        b_, c_, s_ = m1_.shape
        rmask = relu_branch(m1_, A_.view(-1, c_, 1) if A_.numel() == b_ * c_ else A_.view(1, c_, 1),
                            B_.view(-1, c_, 1) if B_.numel() == b_ * c_ else B_.view(1, c_, 1)).permute(0, 2, 1).reshape(b_ * s_, c_).cpu()
        res = {}
        for dt_ in (torch.float32, torch.float64):
            p = {"temporal_encoder." + k: (v.clone().to(dt_) if v.dtype.is_floating_point else v.clone()) for k, v in state.items()}
            for k, v in p.items():
                if v.dtype.is_floating_point and "running" not in k:
                    v.requires_grad_(True)
            cfg = orc.OracleConfig(n_head=nh, d_k=dk, d_model=d_in, ltae_dropout=0.0, positional_encoding=m.positional_encoder is not None)
            xo = x.detach().clone().to(dt_).requires_grad_(True)
            vo, ao = orc.ltae2d_values_attention(xo, dates.to(dt_), pad, p, cfg, training, relu_mask=rmask)
            ((vo * gv.to(dt_)).sum() + (ao * ga.to(dt_)).sum()).backward()
            res[dt_] = (vo.detach().double(), ao.detach().double(), xo.grad.double(), {k[len("temporal_encoder."):]: v.grad.double() for k, v in p.items() if getattr(v, "grad", None) is not None})
        t = res[torch.float64]
        errs = []
        def chk(what, h, r32, r64, floor=1e-4):
            sc = float(r64.abs().max()) or 1.0
            eh, ec = float((h.double().cpu() - r64).abs().max()) / sc, float((r32 - r64).abs().max()) / sc
            if eh > max(floor, 4 * ec):
                errs.append((what, f"{eh:.1e}", f"cpu {ec:.1e}"))
        chk("values", o, res[torch.float32][0], t[0]); chk("attn", a, res[torch.float32][1], t[1]); chk("dx", xh.grad, res[torch.float32][2], t[2])
        for k, par in m.named_parameters():
            if k not in t[3] or par.grad is None:
                continue
            if k.endswith(".bias") and k.replace(".bias", ".weight") in t[3] and float(t[3][k].abs().max()) < 1e-3 * float(t[3][k.replace(".bias", ".weight")].abs().max()):
                continue
            chk("grad " + k, par.grad, res[torch.float32][3][k], t[3][k])
        print(tag + f" train={training}" + (":  ok" if not errs else f":  {errs[:4]}  <<<<<<"), flush=True)
        bad += bool(errs)
    except NotImplementedError as exc:
        print(f"{tag}: refused -- {str(exc)[:120]}", flush=True)
    except Exception as exc:
        bad += 1
        print(f"{tag}: {type(exc).__name__}: {str(exc)[:200]}  <<<<<<", flush=True)
        traceback.print_exc(limit=2)
print("cases outside the contract:", bad)
