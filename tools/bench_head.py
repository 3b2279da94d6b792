"""uncr_head_fwd (out_conv 128 -> 26 + sigmoid / softplus) at the step's shape, event-timed; cold / producer-warmed input (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import engine as E
N, C, H, W, Co = 4, 128, 256, 256, 26
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
w = torch.randn(Co, C, 1, 1, device="cuda") * 0.05; b = torch.randn(Co, device="cuda") * 0.1
for act in ("fp32", "bf16"):
    src = torch.randn(N, C, H, W, device="cuda")
    if act == "bf16":
        src = E.cast(src, E.BF16)
    y = torch.empty_like(src)
    for mode in ("cold", "warm"):
        ts = []
        for it in range(25):
            y.copy_(src)
            if mode == "cold":
                flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out, sv = E.head_forward(y, w, b, 13, True, 1.0, 1e-3); e1.record(); torch.cuda.synchronize()
            if it >= 5:
                ts.append(e0.elapsed_time(e1))
        ts.sort(); ms = ts[len(ts) // 2]
        nb = y.element_size() * N * C * H * W + 2 * 4 * N * Co * H * W
        print("head_fwd %s %s: %.1f us  %.0f GB/s" % (act, mode, ms * 1e3, nb / ms / 1e6), flush=True)
