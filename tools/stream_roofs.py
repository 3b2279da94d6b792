"""HBM stream roofs of this MI355X for the read : write mixes the step's kernels have (run on the GPU box):
    python tools/stream_roofs.py gpurun_out/r04_stream_roofs.json
Each case moves `n_floats` fp32 per stream through libuncr_dev's stream probe (float4 lanes, a contiguous slab per block),
timed with HIP events over `reps` back-to-back launches after a warm-up; TB/s = all bytes read + written / time."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from uncrtaints_amd import hip_backend as hb  # noqa: E402

MODES = {0: ("read_only", 1, 0), 1: ("write_only", 0, 1), 2: ("copy_1r_1w", 1, 1), 3: ("2r_1w", 2, 1), 4: ("1r_2w", 1, 2),
         5: ("3r_1w", 3, 1),
         # the same bytes in the plane-tiled GEMMs' access pattern: 512-byte pieces of 256-row groups, 256 KB apart, persistent blocks
         6: ("tiled512B_2r_1w", 2, 1), 7: ("tiled512B_2r_0w", 2, 0), 8: ("tiled512B_16r_1w", 2, 0.125),
         # the weight-gradient kernels' pattern: one 512-thread block per CU walks a pixel range of a 512-row frame, reading 128 / 256 /
         # 512-byte pieces of every row per visit
         9: ("rows128B_1r_0w", 1, 0), 10: ("rows256B_1r_0w", 1, 0), 11: ("rows512B_1r_0w", 1, 0)}


def main(out_path):
    dev = hb.dev_lib()
    n = 1 << 28                        # 268 M floats = 1.07 GB per stream (beyond the 256 MB Infinity Cache)
    bufs = [torch.empty(n, device="cuda", dtype=torch.float32).normal_() for _ in range(5)]
    s = torch.cuda.current_stream().cuda_stream
    res = {"n_bytes_per_stream": n * 4, "method": __doc__.strip().splitlines()[-2].strip(), "cases": []}
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None       # optional: comma-separated mode names
    for blocks in (256, 512, 2048, 8192):
        for nt in (0, 1):
            for mode, (name, nr, nw) in MODES.items():
                # the tiled patterns run on the GEMMs' persistent grid (two blocks per CU), the rows patterns on the weight gradient's (one)
                if (blocks == 256) != (mode >= 9) or (mode in (6, 7, 8)) != (blocks == 512) or (only and name not in only):
                    continue
                def run():
                    rc = dev.fn["uncr_debug_stream_probe"](*[b.data_ptr() for b in bufs], n, mode, nt, blocks, s)
                    assert rc == 0, rc
                for _ in range(3):
                    run()
                reps = 10
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                tot = (nr + nw) * n * 4
                c = {"mode": name, "nt": bool(nt), "blocks": blocks, "ms": round(ms, 4), "TBps_total": round(tot / ms / 1e9, 3),
                     "TBps_read": round(nr * n * 4 / ms / 1e9, 3), "TBps_written": round(nw * n * 4 / ms / 1e9, 3)}
                res["cases"].append(c)
                print(c, flush=True)
    best = {}
    for c in res["cases"]:
        b = best.get(c["mode"])
        if b is None or c["TBps_total"] > b["TBps_total"]:
            best[c["mode"]] = c
    res["best_per_mode"] = best
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stream_roofs.json")
