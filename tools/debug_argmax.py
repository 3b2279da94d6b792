import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
from conftest import load_golden
from oracle import uncrtaints_oracle as orc
from uncrtaints_amd import engine as E
from uncrtaints_amd.src.backbones import uncrtaints as U
g = load_golden("g1_iso_t6")
state = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
x, d = torch.from_numpy(g["x"]), torch.from_numpy(g["dates"])
cfg = orc.OracleConfig(covmode="iso", out_conv=[14], attn_dropout=0.0)
taps = {}
with torch.no_grad():
    orc.forward({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in state.items()}, x.double(), d.double(), cfg, training=True, taps=taps)
e64 = taps["e"]
_, idx64 = F.adaptive_max_pool2d(e64, (32, 32), return_indices=True)
taps32 = {}
with torch.no_grad():
    orc.forward({k: v.clone() for k, v in state.items()}, x, d, cfg, training=True, taps=taps32)
_, idx32 = F.adaptive_max_pool2d(taps32["e"], (32, 32), return_indices=True)
m = U.UNCRTAINTS(input_dim=15, out_conv=[14], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="iso")
m.load_state_dict(state); m = m.cuda().train()
with torch.no_grad():
    a0 = m.in_conv.smart_forward(x.cuda())
    b, t, c, h, w = a0.shape
    e = m.in_block[0](a0.view(b * t, c, h, w))
    _, idx = E.maxpool_forward(e, 32, 32)
idx = idx.cpu().long()
print("argmax mismatches vs fp64: HIP", int((idx != idx64).sum()), " CPU-fp32", int((idx32 != idx64).sum()), "of", idx64.numel())
mm = (idx != idx64).nonzero()
print("frames of HIP mismatches:", sorted(set(mm[:, 0].tolist())))
for r in mm[:5]:
    n, ch, oy, ox = r.tolist()
    win = e64[n, ch, oy*2:(oy+1)*2, ox*2:(ox+1)*2].reshape(-1)
    top = torch.topk(win, 2).values
    print("  frame", n, "ch", ch, "cell", (oy, ox), "top-2 gap (fp64):", float(top[0]-top[1]), " |e|max", float(e64.abs().max()))
