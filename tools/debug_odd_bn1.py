"""Where does grad[out_block.4.conv.fn.1.weight] of the odd-size batch-norm variant pick up its error?  (GPU box)
Captures the first depthwise backward of the step (out_block.4) and compares the statistics it leaves with fp64 sums over its own du1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from conftest import rel_err
from gpu_util import dev, oracle_run, pool_branch
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd import hip_backend as hb
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U

kw = dict(encoder_norm="batch", encoder_widths=[128, 128])
B, T, H, W = 2, 2, 34, 70
cfg = orc.OracleConfig(attn_dropout=0.0, ltae_dropout=0.0, **kw)
x, y, dates = orc.synthetic_batch(B, T, H, W, seed=7)
x[B - 1, T - 1] = 0.0
torch.manual_seed(6)
m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag", scale_by=1.0, **kw)
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
m = m.to("cuda").train()
cap = {}
orig = hb.call


def spy(name, *a):
    r = orig(name, *a)
    if name == "uncr_dw_bwd_any" and "dw" not in cap:
        cap["dw"] = [t.clone() if torch.is_tensor(t) else t for t in a]
    if name == "uncr_norm_finalize_bwd" and "dw" in cap and "fin" not in cap:
        cap["fin"] = [t.clone() if torch.is_tensor(t) else t for t in a]
    return r


hb.call = spy
E.hb.call = spy
out = m(dev(x), batch_positions=dev(dates))
l, _ = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")(out[:, :, :13], dev(y), out[:, :, 13:26])
l.backward()
hb.call = orig
E.hb.call = orig
pidx, _ = pool_branch(m, state, x, dates, cfg)
taps = {}
_, _, _, g64, _ = oracle_run(state, x, y, dates, cfg, torch.float64, pool_idx=pidx)
_, _, _, g32, _ = oracle_run(state, x, y, dates, cfg, torch.float32, pool_idx=pidx)
k = "out_block.4.conv.fn.1.weight"
print(f"{k}: hip {rel_err(m.get_parameter(k).grad.cpu().numpy(), g64[k].numpy()):.2e} cpu {rel_err(g32[k].numpy(), g64[k].numpy()):.2e}")
du2, h2, h1, k1, k2, k3, kmu, A1, B1, w, du1, part, dwp, mean1, mg, N, C, Hh, Ww, Pc = cap["dw"][:20]
P = Hh * Ww
val = lambda t: t.reshape(N, C, -1)[:, :, :P].double()
s = part.double().sum(1)                      # [N*C, 2]
m1 = mean1.double().view(1, C, 1)
ref0, ref1 = val(du1).sum(-1).reshape(-1), (val(du1) * (val(h1) - m1)).sum(-1).reshape(-1)
print(f"statistics of the kernel's own du1: sum {float((s[:, 0] - ref0).abs().max() / ref0.abs().max()):.2e}  centred sum "
      f"{float((s[:, 1] - ref1).abs().max() / ref1.abs().max()):.2e}")
# gamma gradient from these sums: dgamma[c] = rstd[c] * sum_n s1[n, c]
fin = cap["fin"]
print("finalize args:", [tuple(t.shape) if torch.is_tensor(t) else t for t in fin])
# the mean the kernel centred on, against the fp64 mean of h1
mu64 = val(h1).mean(dim=(0, 2))
print(f"mean1 vs fp64 mean of h1: {float((mean1.double() - mu64).abs().max()):.3e} (|mean| max {float(mu64.abs().max()):.3e}, std of h1 "
      f"{float(val(h1).std()):.3e})")
var64 = val(h1).var(dim=(0, 2), unbiased=False)
print(f"channel |mean| / std of h1: max {float((mu64.abs() / var64.sqrt()).max()):.2f}")
