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
