"""Numerics of the 3-way bf16 split on the bf16 MFMA vs fp64 truth and vs a plain fp32 dot (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import hip_backend as hb
torch.manual_seed(0)
for K in (128, 256, 4096):
    for scale in (1.0, 1e-6):
        A = (torch.randn(32, K, device="cuda") * scale).contiguous(); B = torch.randn(K, 32, device="cuda").contiguous()
        truth = A.double() @ B.double()
        f32 = (A @ B)
        den = truth.abs().max()
        line = f"K={K} scale={scale:g}: torch fp32 {((f32.double()-truth).abs().max()/den).item():.2e}"
        for terms in (1, 3, 6, 9):
            out = torch.empty(32, 32, device="cuda")
            hb.call("uncr_debug_bf16split_probe", A, B, out, K, terms, torch.cuda.current_stream().cuda_stream)
            line += f" | x{terms} {((out.double()-truth).abs().max()/den).item():.2e}"
        print(line)
