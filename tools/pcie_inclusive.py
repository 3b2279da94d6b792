"""Throughput with the batch starting in HOST memory, as the reference's training loop hands it over (a DataLoader batch, then `.to(device)`,
train_reconstruct.py) -- the PCIe-inclusive rate DESIGN.md quotes next to bench.py's `value` (which has the inputs resident in HBM).
    python tools/pcie_inclusive.py [steps]          (GPU box)
Three loops over the same captured step (B = 4, T = 3, 256 x 256, fwd + MGNLL + bwd + Adam):
  resident   : graph replays only (bench.py's definition)
  serial     : every step waits for its own pinned-host -> device copy on the compute stream
  prefetched : the next batch is copied on a second stream into a staging buffer while the current step runs; the step starts with a
               device-to-device copy into the graph's static inputs"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from uncrtaints_amd.optim import FusedAdam
from uncrtaints_amd.src import losses

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
B, T, H = 4, 3, 256
model = bench.build_model(dev, seed=1, act_dtype="fp32")
model.temporal_aggregator.set_seed(1)
crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
opt = FusedAdam(model.parameters(), lr=1e-3)
x, y, dates = bench.synthetic(B, T, H, H, seed=1, device=dev)
counter = torch.zeros(1, dtype=torch.int64, device=dev)
model.temporal_aggregator.step_counter = counter


def eager_step():
    counter.add_(1)
    opt.zero_grad(set_to_none=True)
    out = model(x, batch_positions=dates)
    mean, var = losses.split_prediction(out, 13, 26)
    loss, _ = crit(mean, y, var)
    loss.backward()
    opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager_step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    eager_step()
torch.cuda.synchronize()

host = [tuple(t.cpu().pin_memory() for t in (x, y, dates)) for _ in range(2)]       # two host batches, pinned
nbytes = sum(t.numel() * t.element_size() for t in host[0])


def timed(fn, n):
    for _ in range(10):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def resident(i):
    graph.replay()


def serial(i):
    for d, s in zip((x, y, dates), host[i & 1]):
        d.copy_(s, non_blocking=True)
    graph.replay()


copy_stream = torch.cuda.Stream()
stage = [tuple(torch.empty_like(t) for t in (x, y, dates)) for _ in range(2)]
ready = [torch.cuda.Event(), torch.cuda.Event()]
freed = [torch.cuda.Event(), torch.cuda.Event()]


def issue_copy(i):
    k = i & 1
    with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(freed[k])          # the step that last read this staging buffer has taken its copy
        for d, s in zip(stage[k], host[k]):
            d.copy_(s, non_blocking=True)
        ready[k].record(copy_stream)


for k in range(2):
    freed[k].record()
issue_copy(0)


def prefetched(i):
    k = i & 1
    issue_copy(i + 1)                              # batch i + 1 travels while step i computes
    cur = torch.cuda.current_stream()
    cur.wait_event(ready[k])
    for d, s in zip((x, y, dates), stage[k]):
        d.copy_(s, non_blocking=True)
    freed[k].record(cur)
    graph.replay()


# the copy alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    for d, s in zip((x, y, dates), host[i & 1]):
        d.copy_(s, non_blocking=True)
torch.cuda.synchronize()
copy_ms = (time.perf_counter() - t0) / 50 * 1e3
res = {"batch_bytes": nbytes, "pinned_host_to_device_ms": round(copy_ms, 3), "pinned_host_to_device_gbs": round(nbytes / copy_ms / 1e6, 1),
       "steps": steps}
for name, fn in (("resident", resident), ("serial", serial), ("prefetched", prefetched)):
    ms = timed(fn, steps)
    res[name] = {"ms_per_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1)}
print(json.dumps(res))
