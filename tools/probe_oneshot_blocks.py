"""The stream probe of libuncr_dev with one-shot blocks: read-only and 2r:1w streams over 1 GB with 2048 ... 262144 blocks, i.e. 128 ... 1
float4 per lane and block lifetime (run on the GPU box; profiles/r04g_probe_oneshot_blocks.log)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from uncrtaints_amd import hip_backend as hb
dev = hb.dev_lib()
n = 1 << 28
bufs = [torch.empty(n, device="cuda", dtype=torch.float32).normal_() for _ in range(5)]
s = torch.cuda.current_stream().cuda_stream
for mode, name in ((0, "read_only"), (3, "2r_1w")):
    for nt in (0, 1):
        for blocks in (2048, 8192, 16384, 65536, 262144):
            def run():
                rc = dev.fn["uncr_debug_stream_probe"](*[b.data_ptr() for b in bufs], n, mode, nt, blocks, s); assert rc == 0
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            nb = (1 if mode == 0 else 3) * n * 4
            print(f"{name} nt={nt} blocks={blocks}: {ms*1e3:.0f} us  {nb/ms/1e9:.2f} TB/s  (float4 per thread: {n/4/blocks/256:.0f})", flush=True)
