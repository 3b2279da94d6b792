"""BaseModel.optimize_parameters at the bench shape (B=4, T=3, 256x256): eager launches vs config.hip_graph (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from uncrtaints_amd.src.backbones.base_model import BaseModel
dev = "cuda"
x, y, dates = bench.synthetic(4, 3, 256, 256, seed=1, device=torch.device(dev))
for hip_graph in (False, True):
    cfg = SimpleNamespace(model="uncrtaints", use_sar=True, encoder_widths=[128], decoder_widths=[128] * 5, out_conv=[26],
                          mean_nonLinearity=True, var_nonLinearity="softplus", agg_mode="att_group", encoder_norm="group",
                          decoder_norm="batch", n_head=16, d_model=256, d_k=4, pad_value=0, padding_mode="reflect",
                          positional_encoding=True, covmode="diag", scale_by=1.0, separate_out=False, use_v=False,
                          block_type="mbconv", pretrain=False, loss="MGNLL", lr=1e-3, gamma=1.0, device=dev, chunk_size=None,
                          hip_graph=hip_graph)
    torch.manual_seed(1)
    m = BaseModel(cfg).to(dev).train()
    batch = {"A": x, "B": y, "dates": dates, "masks": None}
    for _ in range(5):
        m.set_input(batch); m.optimize_parameters()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        m.set_input(batch); m.optimize_parameters()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"BaseModel.optimize_parameters hip_graph={hip_graph}: {dt*1e3:.2f} ms/step = {4/dt:.1f} samples/s, loss {m.loss_G.item():.4f}")
