"""The reference's validation loop (train_reconstruct.py:302-309: eval mode, no_grad, set_input / forward / get_loss_G / rescale)
through BaseModel at B = 1 and B = 4 (T=3, 256x256): eager launches vs config.hip_graph (run on the GPU box)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import bench
from uncrtaints_amd.src.backbones.base_model import BaseModel
dev = "cuda"
out = {}
for B in (1, 4):
    x, y, dates = bench.synthetic(B, 3, 256, 256, seed=1, device=torch.device(dev))
    for hip_graph in (False, True):
        cfg = SimpleNamespace(model="uncrtaints", use_sar=True, encoder_widths=[128], decoder_widths=[128] * 5, out_conv=[26],
                              mean_nonLinearity=True, var_nonLinearity="softplus", agg_mode="att_group", encoder_norm="group",
                              decoder_norm="batch", n_head=16, d_model=256, d_k=4, pad_value=0, padding_mode="reflect",
                              positional_encoding=True, covmode="diag", scale_by=1.0, separate_out=False, use_v=False,
                              block_type="mbconv", pretrain=False, loss="MGNLL", lr=1e-3, gamma=1.0, device=dev, chunk_size=None,
                              hip_graph=hip_graph)
        torch.manual_seed(1)
        m = BaseModel(cfg).to(dev).eval()
        batch = {"A": x, "B": y, "dates": dates, "masks": None}

        def it():
            with torch.no_grad():
                m.set_input(batch); m.forward(); m.get_loss_G(); m.rescale()
        for _ in range(5):
            it()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            it()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        out[f"B{B}_{'graph' if hip_graph else 'eager'}_ms"] = round(dt * 1e3, 3)
        print(f"validation iteration B={B} hip_graph={hip_graph}: {dt*1e3:.2f} ms = {B/dt:.1f} samples/s, loss {m.loss_G.item():.4f}")
print(json.dumps(out))
