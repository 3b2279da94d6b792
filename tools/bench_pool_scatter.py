"""uncr_pool_scatter_stats at the step's shape (12 frames x 128 channels, 256x256 -> 32x32 windows), event-timed per launch; "cold":
512 MB pushed through the Infinity Cache before every call, "warm": de rewritten by a copy kernel right before (as the aggregation
backward leaves it in the step).  UNCR_HIP_LIB selects a library variant (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncrtaints_amd import engine as E, hip_backend as hb
planes, H, W = 12 * 128, 256, 256
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
tag = os.path.basename(os.environ.get("UNCR_HIP_LIB", "base"))
for act in ("fp32", "bf16"):
    dt = torch.float32 if act == "fp32" else torch.bfloat16
    e = torch.randn(planes, H, W, device="cuda").to(dt)
    down, idx = E.maxpool_forward(e, 32, 32)
    src = torch.randn(planes, H, W, device="cuda").to(dt)
    de = torch.empty_like(src)
    h3 = torch.randn(planes, H, W, device="cuda").to(dt)
    dd = torch.randn(planes, 32, 32, device="cuda")
    slots = hb.query("uncr_ew_slots", H * W)
    part = torch.empty(planes, slots, 2, device="cuda")
    amax = torch.empty(planes, slots, device="cuda") if act == "fp32" else None
    for mode in ("cold", "warm"):
        ts = []
        for it in range(25):
            de.copy_(src)
            if mode == "cold":
                flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hb.call("uncr_pool_scatter_stats", dd, idx, de, h3, part, planes, H, W, 32, 32, 0 if act == "fp32" else 1, amax, E._stream())
            e1.record(); torch.cuda.synchronize()
            if it >= 5:
                ts.append(e0.elapsed_time(e1))
        ts.sort(); ms = ts[len(ts) // 2]
        print("%s %s %s: %.1f us  %.0f GB/s" % (tag, act, mode, ms * 1e3, 2 * de.element_size() * planes * H * W / ms / 1e6), flush=True)
