"""SE-pool micro-benchmark (uncr_ew ops 7 / 17) at the step's two shapes, one chunk per block against four (run on the GPU box).
"cold": 512 MB are pushed through the Infinity Cache before every call (the operand comes from HBM); "warm": a copy kernel
rewrites the operand right before the call, as the depthwise forward does in the step (at N = 4 most of it then sits in the cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import engine as E
C, P = 256, 65536
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for act in ("fp32", "bf16"):
    for N in (4, 12):
        src = torch.randn(N, C, P, device="cuda")
        if act == "bf16":
            src = E.cast(src, E.BF16)
        h2 = torch.empty_like(src)
        A = torch.rand(N * C, device="cuda"); B = torch.randn(N * C, device="cuda")
        for mode in ("cold", "warm"):
            for four in (False, True, False, True):
                if True:
                    op = E.EW_SE_POOL4 if four else E.EW_SE_POOL
                    ts = []
                    for it in range(25):
                        h2.copy_(src)
                        if mode == "cold":
                            flush.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); E.ew(op, h2, k=(A, B, None, None), want_part=True, planes=N * C, P=P); e1.record(); torch.cuda.synchronize()
                        if it >= 5:
                            ts.append(e0.elapsed_time(e1))
                    ts.sort(); ms = ts[len(ts) // 2]
                    print("%s N=%2d %s chunks/block=%d: %.1f us  %.0f GB/s" % (act, N, mode, 4 if four else 1, ms * 1e3, h2.element_size() * N * C * P / ms / 1e6))
