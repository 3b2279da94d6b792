"""SE-pool (uncr_ew SE_POOL) micro-benchmark at the bench shape (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import engine as E
N, C, P = 4, 256, 65536
h2 = torch.randn(N, C, P, device="cuda"); A = torch.rand(N * C, device="cuda"); B = torch.randn(N * C, device="cuda")
def run():
    E.ew(E.EW_SE_POOL, h2, k=(A, B, None, None), want_part=True, planes=N * C, P=P)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("SE pool: %.1f us  %.0f GB/s" % (ms * 1e3, 4.0 * N * C * P / ms / 1e6))
