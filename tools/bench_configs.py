"""Side benchmarks for BASELINE configs 4 (T=6 training) and 5 (5-member iso ensemble inference).
Not the driver's bench (that is bench.py, config 2/3); prints one JSON line per config."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_model, synthetic
from uncrtaints_amd import engine as E
from uncrtaints_amd.src import losses
from uncrtaints_amd.src.backbones import uncrtaints as U
from uncrtaints_amd.src.learning.weight_init import weight_init

dev = torch.device("cuda", 0)

def timed(fn, warm, steps):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

# config 4: --input_t 6, B=2/GPU, train step
model = build_model(dev, 1); model.temporal_aggregator.set_seed(1)
crit = losses.MultiGaussianNLLLoss(reduction="mean", full=True, mode="diag")
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
x, y, dates = synthetic(2, 6, 256, 256, 1, dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(x, batch_positions=dates)
    l, _ = crit(out[:, :, :13], y, out[:, :, 13:26]); l.backward(); opt.step()
dt = timed(step, 3, 10)
print(json.dumps({"config": "4: --input_t 6, B=2, 256x256, fwd+MGNLL+bwd+Adam, fp32", "samples_per_s": round(2 / dt, 2), "ms_per_step": round(dt * 1e3, 2)}))

# config 5: iso, 5 members, inference only
members = []
for s in range(5):
    torch.manual_seed(s)
    m = U.UNCRTAINTS(input_dim=15, out_conv=[14], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="iso")
    m.apply(weight_init); members.append(m.to(dev).eval())
x, y, dates = synthetic(4, 3, 256, 256, 2, dev)
def infer():
    with torch.no_grad():
        outs = [m(x, batch_positions=dates) for m in members]
        mu = torch.stack([o[:, 0, :13].contiguous() for o in outs])
        var = torch.stack([o[:, 0, 13:14].expand(-1, 13, -1, -1).contiguous() for o in outs])
        return E.ensemble_combine(mu, var, "both")
dt = timed(infer, 2, 5)
print(json.dumps({"config": "5: covmode iso, 5-member ensemble, inference, B=4, T=3, 256x256, fp32", "samples_per_s": round(4 / dt, 2), "ms_per_batch": round(dt * 1e3, 2)}))
