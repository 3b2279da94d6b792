import sys, time, torch
sys.path.insert(0, '/root/repo')
from oracle import uncrtaints_oracle as orc
cfg = orc.OracleConfig(); p = orc.init_params(cfg, seed=1)
x, y, dates = orc.synthetic_batch(1, 3, 256, 256, seed=1)
for thr in (8, 16, 32, 64):
    torch.set_num_threads(thr)
    pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
    ts=[]
    for i in range(2):
        t0=time.perf_counter(); out = orc.forward(pt, x, dates, cfg, training=True); orc.loss_from_output(out, y, cfg).backward(); ts.append(time.perf_counter()-t0)
    print(thr, [round(t,2) for t in ts], flush=True)
