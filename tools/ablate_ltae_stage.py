"""The L-TAE stage's cost INSIDE the captured training step, by ablation (VERDICT r3 item 4's second figure):
step time of the full captured step minus step time of a captured step whose two stage calls launch nothing and hand back the
results an earlier real step left behind (engine.dev_options(ltae_replay=...)): same tensors, same values downstream, no stage
kernels -- compose, fused forward, aggregation forward | aggregation backward (+ fold reduce), fused backward, compose backward AND
the pooled-gradient scatter + statistics pass.  Interleaved chunks of replays of the two graphs inside one process.

    python tools/ablate_ltae_stage.py [out.json] [--act-dtype bf16]
"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch

import bench
from uncrtaints_amd import engine as E
from uncrtaints_amd.optim import FusedAdam
from uncrtaints_amd.src import losses


def main():
    out_path = next((a for a in sys.argv[1:] if a.endswith(".json")), None)
    act = "bf16" if "bf16" in sys.argv else "fp32"
    dev = torch.device("cuda", 0)
    B, T, H = 4, 3, 256
    model = bench.build_model(dev, seed=1, act_dtype=act)
    model.temporal_aggregator.set_seed(1)
    crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
    opt = FusedAdam(model.parameters(), lr=1e-3)
    x, y, dates = bench.synthetic(B, T, H, H, seed=1, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    model.temporal_aggregator.step_counter = counter

    def step():
        counter.add_(1)
        opt.zero_grad(set_to_none=True)
        out = model(x, batch_positions=dates)
        mean, var = losses.split_prediction(out, 13, 26)
        loss, _ = crit(mean, y, var)
        loss.backward()
        opt.step()
        return loss

    def capture():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        return g

    full = capture()
    with E.dev_options(ltae_replay="record"):
        step()
    torch.cuda.synchronize()
    with E.dev_options(ltae_replay="replay"):
        ablated = capture()

    def timed(g, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    n = 200
    timed(full, 50), timed(ablated, 50)
    rows = [(timed(full, n), timed(ablated, n)) for _ in range(5)]
    f = sorted(r[0] for r in rows)[2]
    a = sorted(r[1] for r in rows)[2]
    deltas = sorted(r[0] - r[1] for r in rows)
    delta = deltas[2]
    a_bytes = (3 * T + 2) * 128 * H * H * (2 if act == "bf16" else 4) * B
    res = {"act_dtype": act, "step_ms_full": round(f, 4), "step_ms_without_stage_kernels": round(a, 4),
           "stage_ms_by_ablation_incl_scatter_stats": round(delta, 4), "pairs_ms": [[round(u, 4), round(v, 4)] for u, v in rows],
           "algorithmic_bytes": a_bytes,
           "roofline_frac_incl_scatter_stats": round(a_bytes / (delta * 1e-3) / 8e12, 4),
           "method": "median of 5 interleaved chunks of 200 graph replays each: the captured step vs the same step whose two L-TAE stage "
                     "calls launch nothing (engine.dev_options(ltae_replay='replay')); the difference contains the scatter + statistics "
                     "pass that bench.py's sum-of-kernels figure attributes to the encoder (its roofline_frac_with_scatter_stats is the "
                     "like-for-like number)"}
    print(json.dumps(res))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
