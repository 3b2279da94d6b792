"""CPU, world_size 2, gloo: the bucketed all-reduce gives every rank the rank-averaged gradients, equal to a
single-process run on the concatenated batch for a per-sample-mean loss (the DDP contract of SURVEY 8(e))."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class Toy(torch.nn.Module):
    """Parameter names mimic the three UNCRTAINTS bucket groups."""

    def __init__(self):
        super().__init__()
        self.in_conv = torch.nn.Linear(6, 8)
        self.temporal_encoder = torch.nn.Linear(8, 8)
        self.out_block = torch.nn.Linear(8, 4)
        self.out_conv = torch.nn.Linear(4, 2)

    def forward(self, x):
        return self.out_conv(torch.relu(self.out_block(torch.relu(self.temporal_encoder(torch.relu(self.in_conv(x)))))))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uncrtaints_amd.parallel import BucketedDataParallel, default_buckets
    torch.manual_seed(100 + rank)          # deliberately different initial weights: broadcast must fix them
    m = Toy()
    dp = BucketedDataParallel(m)
    assert len(dp.buckets) == 3
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    for _ in range(2):                     # two steps: zero_grad/finish bookkeeping must reset
        dp.zero_grad()
        loss = ((dp(xs) - ys) ** 2).mean()
        loss.backward()
        dp.finish()
    # numpy, not tensors: torch tensors travel through mp queues as shared-memory handles that die with the sender
    grads = {n: p.grad.numpy().copy() for n, p in m.named_parameters()}
    weights = {n: p.detach().numpy().copy() for n, p in m.named_parameters()}
    # optimizer.zero_grad() (torch's default set_to_none=True) is as good as dp.zero_grad(): the wrapper packs whatever gradient
    # tensors autograd produced into its buckets (one multi-tensor copy per bucket) and points .grad at the bucket views
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad()
    dp.zero_grad()
    ((dp(xs) - ys) ** 2).mean().backward()
    dp.finish()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, torch.from_numpy(grads[n]), rtol=0, atol=0), n
        assert any(b["flat"].data_ptr() <= p.grad.data_ptr() < b["flat"].data_ptr() + 4 * b["flat"].numel() for b in dp.buckets)
    # ... and so is zeroing in place: autograd then accumulates straight into the bucket views
    for p in m.parameters():
        p.grad.zero_()
    for b in dp.buckets:
        b["ready"], b["handle"], b["packed"] = 0, None, False
    ((dp(xs) - ys) ** 2).mean().backward()
    dp.finish()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, torch.from_numpy(grads[n]), rtol=1e-6, atol=1e-7), n
    # bookkeeping of the bench's N > 1 line: bytes per all-reduce, the wait-time record (empty without a GPU), and the segmented
    # interface: reduce_bucket drives the collectives by hand, so it refuses a wrapper whose hooks launch them too
    assert dp.bucket_bytes() == [4 * b["flat"].numel() for b in dp.buckets] and sum(dp.bucket_bytes()) == 4 * sum(p.numel() for p in m.parameters())
    assert dp.wait_ms() == []
    try:
        dp.reduce_bucket(0)
        refused = False
    except RuntimeError as exc:
        refused = "overlap=False" in str(exc)
    assert refused
    m2 = Toy()
    dp2 = BucketedDataParallel(m2, overlap=False)
    dp2.zero_grad()
    ((dp2(xs) - ys) ** 2).mean().backward()
    for bi in range(len(dp2.buckets)):
        dp2.reduce_bucket(bi)
    try:
        dp2.reduce_bucket(0)
        twice = False
    except RuntimeError:
        twice = True
    assert twice
    dp2.finish()
    q.put((rank, grads, weights))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, w0), (_, g1, w1) = [(r, {k: torch.from_numpy(v) for k, v in g.items()},
                                 {k: torch.from_numpy(v) for k, v in w.items()}) for r, g, w in res]
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=1e-7), n        # every rank holds the same averaged gradient
        assert torch.equal(w0[n], w1[n]), n                      # broadcast made the replicas identical
    # single-process reference on the concatenated batch with rank-0 weights
    m = Toy()
    m.load_state_dict(w0)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
    ((m(X) - Y) ** 2).mean().backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, g0[n], atol=1e-6), n


def test_default_buckets_cover_uncrtaints_parameters():
    from uncrtaints_amd.parallel import default_buckets
    from uncrtaints_amd.src.backbones import uncrtaints
    m = uncrtaints.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus")
    b = default_buckets(list(m.named_parameters()))
    assert len(b) == 3
    assert sorted(n for x in b for n in x) == sorted(n for n, _ in m.named_parameters())
    assert b[0][0].startswith("out_block") or b[0][0].startswith("out_conv")


class ToyUnused(Toy):
    """One parameter of the middle bucket never receives a gradient (an unused / frozen-in-effect branch): that bucket's hook count
    stays short of its size, no hook launches its all-reduce, and finish() has to pack (zeros for the missing gradient) and reduce."""

    def __init__(self):
        super().__init__()
        self.temporal_encoder_unused = torch.nn.Parameter(torch.ones(5))


def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uncrtaints_amd.parallel import BucketedDataParallel
    torch.manual_seed(200 + rank)
    m = ToyUnused()
    dp = BucketedDataParallel(m)
    sizes = [b["n"] for b in dp.buckets]
    g = torch.Generator().manual_seed(11)
    X, Y = torch.randn(4 * world + 3, 6, generator=g), torch.randn(4 * world + 3, 2, generator=g)
    # uneven shards: the last rank takes the remainder (a per-rank mean then differs from the global mean: the test compares with
    # the mean of the per-shard gradients, which is what an averaging all-reduce defines)
    lo = rank * 4
    hi = lo + 4 if rank < world - 1 else X.shape[0]
    launched_by_hooks = None
    for step in range(2):
        dp.zero_grad()
        ((dp(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
        if step == 0:
            launched_by_hooks = [b["handle"] is not None for b in dp.buckets]
        dp.finish()
    grads = {n: (p.grad.numpy().copy() if p.grad is not None else None) for n, p in m.named_parameters()}
    q.put((rank, grads, {n: p.detach().numpy().copy() for n, p in m.named_parameters()}, launched_by_hooks, sizes, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_uneven_bucket_and_shards():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the bucket that holds the gradient-less parameter was NOT launched from a hook (finish() reduced it); the others were
    for _, _, _, launched, sizes, _ in res:
        assert launched == [True, False, True], launched
        assert sizes[1] == 3
    g0 = res[0][1]
    for r in range(1, world):
        for n in g0:
            assert torch.allclose(torch.from_numpy(res[r][1][n]), torch.from_numpy(g0[n]), atol=1e-7), (r, n)
            assert (res[r][2][n] == res[0][2][n]).all(), n
    assert (g0["temporal_encoder_unused"] == 0).all()           # zeros went through the all-reduce (torch-DDP semantics)
    # reference: mean over ranks of the per-shard gradients, with rank 0's (broadcast) weights
    g = torch.Generator().manual_seed(11)
    X, Y = torch.randn(4 * world + 3, 6, generator=g), torch.randn(4 * world + 3, 2, generator=g)
    acc = None
    for r in range(world):
        lo, hi = res[r][5]
        m = ToyUnused()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in res[0][2].items()})
        ((m(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
        gs = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
        acc = gs if acc is None else {n: acc[n] + gs[n] for n in gs}
    for n in acc:
        assert torch.allclose(acc[n] / world, torch.from_numpy(g0[n]), atol=1e-6), n


# ---- bench.py's rank bookkeeping on a world of 8 (the driver's `--gpus 8` launch, here eight gloo CPU processes) ----
def _bench_mod():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_w8", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def _bench_worker(rank, world, port, q):
    # what torch.distributed.run sets for `--nproc-per-node 8`
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    b = _bench_mod()
    w, r, lr = b.rank_env(world)
    assert (w, r, lr) == (world, rank, rank)
    info = b.join_ranks("gloo", r, w, None, f"node0:gpu-uuid-{lr}:{lr}", "cpu-stand-in")
    assert info["devices"] == [f"node0:gpu-uuid-{i}:{i}" for i in range(world)]
    b.check_roster(info["devices"], world, "nccl")          # eight distinct devices: what the RCCL launch requires
    from uncrtaints_amd.parallel import BucketedDataParallel
    torch.manual_seed(rank)
    m = Toy()
    dp = BucketedDataParallel(m)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(4 * world, 6, generator=g), torch.randn(4 * world, 2, generator=g)
    steps, B = 3, 4
    for _ in range(steps):
        dp.zero_grad()
        ((dp(X[rank * B:(rank + 1) * B]) - Y[rank * B:(rank + 1) * B]) ** 2).mean().backward()
        dp.finish()
    # the job's time is the slowest rank's; the value is the whole job's samples per second (weak scaling)
    dt = b.max_over_ranks(0.01 * (rank + 1), torch.device("cpu"))
    assert abs(dt - 0.01 * world) < 1e-12
    value = b.job_value(w, B, steps, dt)
    assert abs(value - world * B * steps / (0.01 * world)) < 1e-9
    coll = b.collective_object(info, dp, [0.0] * steps)
    if rank == 0:
        q.put({"collective": coll, "value": value, "grad0": m.in_conv.weight.grad.numpy().copy()})
    dist.barrier()
    dist.destroy_process_group()


def test_bench_rank_bookkeeping_world_8():
    """bench.py's N > 1 plumbing on eight gloo ranks: environment parsing, the device roster (eight distinct devices), the
    max-over-ranks clock, the weak-scaling arithmetic of `value`, the complete `collective` object, and bucketed all-reduces that
    give every rank the gradient of the concatenated batch."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    c = res["collective"]
    assert len(c["devices"]) == 8 and len(set(c["devices"])) == 8
    for k in ("devices", "rccl_version", "device_name", "bucket_bytes", "all_reduces_per_step", "wait_ms_per_step", "wait_ms_max"):
        assert k in c, k
    assert c["all_reduces_per_step"] == 3 and len(c["bucket_bytes"]) == 3 and all(v > 0 for v in c["bucket_bytes"])
    assert abs(res["value"] - 8 * 4 * 3 / 0.08) < 1e-6
    # single-process reference on the concatenated batch (per-sample-mean loss: the rank average IS the global gradient)
    torch.manual_seed(0)
    m = Toy()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(32, 6, generator=g), torch.randn(32, 2, generator=g)
    ((m(X) - Y) ** 2).mean().backward()
    assert torch.allclose(torch.from_numpy(res["grad0"]), m.in_conv.weight.grad, atol=1e-6)


def test_bench_roster_rejects_shared_devices_and_bad_env(monkeypatch):
    import pytest
    b = _bench_mod()
    with pytest.raises(SystemExit):
        b.check_roster(["n:a:0", "n:a:0"], 2, "nccl")       # two RCCL ranks on one GPU
    b.check_roster(["n:a:0", "n:a:0"], 2, "gloo")           # the single-GPU development mode may share it
    with pytest.raises(SystemExit):
        b.check_roster(["n:a:0", None], 2, "nccl")
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(SystemExit):
        b.rank_env(8)                                       # --gpus 8 under a 4-rank launcher
    assert b.rank_env(4)[:2] == (4, 1)
