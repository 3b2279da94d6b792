"""The L-TAE stage's cost INSIDE the captured training step, by ablation (VERDICT r3 item 4's second figure):
step time of the full captured step minus step time of a captured step whose two stage calls launch nothing and hand back the
results an earlier real step left behind (engine.dev_options(ltae_replay=...)): same tensors, same values downstream, no stage
kernels -- compose, fused forward, aggregation forward | aggregation backward (+ fold reduce), fused backward, compose backward AND
the pooled-gradient scatter + statistics pass.  One process per variant (median of 5 x 200 graph replays); tools/ablate_ltae_stage.sh interleaves three pairs inside one GPU session
and writes the JSON.

    python tools/ablate_ltae_stage.py [--ablated] [bf16]
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

    def capture(record=False):
        # every eager step before the capture runs on a side stream: an AccumulateGrad node bound to the default stream would pull
        # that stream into the capture (and hipStreamEndCapture crashes on the unjoined fork)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if record:
                with E.dev_options(ltae_replay="record"):
                    step()
                E._LTAE_REPLAY = "replay"
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

    ablate = "--ablated" in sys.argv
    graph = capture(record=ablate)

    def timed(g, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    timed(graph, 50)
    ms = sorted(timed(graph, 200) for _ in range(5))[2]
    a_bytes = (3 * T + 2) * 128 * H * H * (2 if act == "bf16" else 4) * B
    print(json.dumps({"act_dtype": act, "ablated": ablate, "step_ms": round(ms, 4), "algorithmic_bytes_of_the_stage": a_bytes}))


if __name__ == "__main__":
    main()
