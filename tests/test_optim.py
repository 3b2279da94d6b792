"""FusedAdam (uncrtaints_amd/optim.py): the reference's optimizer (base_model.py:48, torch.optim.Adam defaults) as one HIP launch.
Compared step by step with torch.optim.Adam itself; checkpoints travel between the two classes."""
import pytest
import torch

SHAPES = [(128, 15, 1, 1), (128,), (256, 128, 1, 1), (256, 1, 3, 3), (32, 256), (16, 4), (1,), (2049,), (4099, 3), (26, 128, 1, 1)]


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 10.0 ** float(torch.randint(-3, 2, (1,), generator=g))).to(dev)) for s in SHAPES]


def _grads(ps, step, scale=1.0):
    g = torch.Generator().manual_seed(100 + step)
    for p in ps:
        p.grad = (torch.randn(p.shape, generator=g) * scale * 10.0 ** float(torch.randint(-4, 3, (1,), generator=g))).to(p.device)


def test_cpu_parameters_take_torchs_own_step():
    from uncrtaints_amd.optim import FusedAdam
    a, b = _params("cpu"), _params("cpu")
    oa, ob = FusedAdam(a, lr=3e-3), torch.optim.Adam(b, lr=3e-3)
    for s in range(3):
        _grads(a, s); _grads(b, s)
        oa.step(); ob.step()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert set(oa.state_dict()["state"][0]) == set(ob.state_dict()["state"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adam_follows_torch_adam(wd):
    """Ten steps with gradients of very different magnitudes and a learning-rate schedule: parameters and both moments stay within a few
    fp32 roundings of torch.optim.Adam's (whose own foreach and fused implementations differ from each other by as much)."""
    from uncrtaints_amd.optim import FusedAdam
    a, b = _params("cuda"), _params("cuda")
    oa, ob = FusedAdam([{"params": a}], lr=2e-3, weight_decay=wd), torch.optim.Adam([{"params": b}], lr=2e-3, weight_decay=wd)
    sa, sb = (torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.8) for o in (oa, ob))
    for s in range(10):
        _grads(a, s); _grads(b, s)
        oa.step(); ob.step()
        if s % 3 == 2:
            sa.step(); sb.step()
    worst = {"update": 0.0, "exp_avg": 0.0, "exp_avg_sq": 0.0}
    for x, y, y0 in zip(a, b, _params("cuda")):
        # the parameters are compared through what the optimizer did to them -- the distance travelled in ten steps -- plus the
        # rounding of the parameter itself (a few ulp of its magnitude: the two implementations round different partial results)
        worst["update"] = max(worst["update"], float((x.detach() - y.detach()).abs().max() / ((y.detach() - y0.detach()).abs().max() + 0.25 * y.detach().abs().max())))
        for k in ("exp_avg", "exp_avg_sq"):
            u, v = oa.state[x][k], ob.state[y][k]
            worst[k] = max(worst[k], float((u - v).abs().max() / v.abs().max().clamp_min(1e-30)))
        assert float(oa.state[x]["step"]) == float(ob.state[y]["step"]) == 10.0
    print(f"[parity] FusedAdam vs torch.optim.Adam after 10 steps (weight_decay {wd}): worst rel err {worst}")
    assert max(worst.values()) < 2e-6


@pytest.mark.gpu
def test_checkpoints_travel_between_fused_and_torch_adam():
    from uncrtaints_amd.optim import FusedAdam
    a, b = _params("cuda"), _params("cuda")
    oa, ob = FusedAdam(a, lr=1e-3), torch.optim.Adam(b, lr=1e-3)
    for s in range(3):
        _grads(a, s); _grads(b, s)
        oa.step(); ob.step()
    # torch -> fused and fused -> torch, then three more steps on both sides
    c, d = _params("cuda"), _params("cuda")
    for src, dst in ((b, c), (a, d)):
        for x, y in zip(src, dst):
            y.data.copy_(x.data)
    oc, od = FusedAdam(c, lr=1e-3), torch.optim.Adam(d, lr=1e-3)
    import copy       # (a checkpoint on disk is a copy; load_state_dict itself shares the tensors it is given)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    od.load_state_dict(copy.deepcopy(oa.state_dict()))
    for s in range(3, 6):
        for ps in (a, b, c, d):
            _grads(ps, s)
        for o in (oa, ob, oc, od):
            o.step()
    for x, y, z, w in zip(a, b, c, d):
        for t in (y, z, w):
            assert float((x - t).abs().max() / (6e-3 + 0.25 * x.abs().max())) < 5e-6        # 6e-3: the distance six steps of 1e-3 cover
    assert float(oc.state[c[0]]["step"]) == float(od.state[d[0]]["step"]) == 6.0


@pytest.mark.gpu
def test_fused_adam_in_a_captured_graph_with_a_device_learning_rate():
    """Captured once, replayed: the step counter advances on the device, fresh gradients are read from the same addresses, and a
    learning rate changed on the device reaches the replays."""
    from uncrtaints_amd.optim import FusedAdam
    a, b = _params("cuda"), _params("cuda")
    lr = torch.tensor(1e-3, device="cuda")
    oa, ob = FusedAdam(a, lr=lr), torch.optim.Adam(b, lr=1e-3)
    gbuf = [torch.zeros_like(p) for p in a]
    for p, g in zip(a, gbuf):
        p.grad = g

    def feed(step):
        _grads(b, step)
        for g, q in zip(gbuf, b):
            g.copy_(q.grad)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        feed(0); oa.step()
    torch.cuda.current_stream().wait_stream(side)
    ob.step()
    graph = torch.cuda.CUDAGraph()
    feed(1)
    with torch.cuda.graph(graph):
        oa.step()
    graph.replay()          # (a capture records, it does not execute)
    ob.step()
    for s in range(2, 6):
        if s == 4:
            lr.fill_(5e-4)
            ob.param_groups[0]["lr"] = 5e-4
        feed(s)
        graph.replay()
        ob.step()
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert float((x - y).abs().max() / (5e-3 + 0.25 * y.abs().max())) < 5e-6            # 5e-3: the distance the six steps cover
    assert float(oa.state[a[0]]["step"]) == 6.0


@pytest.mark.gpu
def test_parameters_that_join_later_keep_their_own_step_count():
    """torch.optim.Adam counts steps PER PARAMETER: layers unfrozen after `unfreeze_after` epochs (train_reconstruct.py:657-660), or
    any parameter without a gradient in some steps, start their bias corrections at 1 while the others are far ahead.  Half of the
    group receives gradients from step 0, the other half only from step 4 on -- the first tensor of the group among the late ones."""
    from uncrtaints_amd.optim import FusedAdam
    a, b = _params("cuda"), _params("cuda")
    oa, ob = FusedAdam(a, lr=2e-3), torch.optim.Adam(b, lr=2e-3)
    late = {0, 2, 5, 7}
    for s in range(9):
        _grads(a, s); _grads(b, s)
        if s < 4:
            for i in late:
                a[i].grad = None; b[i].grad = None
        oa.step(); ob.step()
    for i, (x, y, y0) in enumerate(zip(a, b, _params("cuda"))):
        assert float(oa.state[x]["step"]) == float(ob.state[y]["step"]) == (5.0 if i in late else 9.0)
        e = float((x.detach() - y.detach()).abs().max() / ((y.detach() - y0.detach()).abs().max() + 0.25 * y.detach().abs().max()))
        assert e < 2e-6, (i, e)
        for k in ("exp_avg", "exp_avg_sq"):
            u, v = oa.state[x][k], ob.state[y][k]
            assert float((u - v).abs().max() / v.abs().max().clamp_min(1e-30)) < 2e-6, (i, k)


@pytest.mark.gpu
def test_load_state_dict_keeps_a_device_learning_rate_and_signals_new_state():
    """A checkpoint of the reference (or of an eager run) holds `lr` as a float; torch's load_state_dict would replace the device scalar
    a captured step reads.  FusedAdam keeps the tensor, fills it with the loaded value, drops its address tables and bumps state_epoch."""
    import copy
    from uncrtaints_amd.optim import FusedAdam
    a, b = _params("cuda"), _params("cuda")
    ob = torch.optim.Adam(b, lr=7e-4)
    for s in range(2):
        _grads(b, s); ob.step()
    lr = torch.tensor(1e-3, device="cuda")
    oa = FusedAdam(a, lr=lr)
    sched = torch.optim.lr_scheduler.ExponentialLR(oa, gamma=0.5)
    _grads(a, 0); oa.step()
    e0 = oa.state_epoch
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))
    assert oa.param_groups[0]["lr"] is lr and abs(float(lr) - 7e-4) < 1e-10
    assert oa.state_epoch == e0 + 1 and not oa._uncr_tables
    for x, y in zip(a, b):
        x.data.copy_(y.data)
    sched.step(); ob.param_groups[0]["lr"] = float(oa.param_groups[0]["lr"])
    assert oa.param_groups[0]["lr"] is lr          # the scheduler updates the SAME device scalar
    _grads(a, 5); _grads(b, 5)
    oa.step(); ob.step()
    for x, y in zip(a, b):
        assert float((x - y).abs().max() / (1e-3 + 0.25 * y.abs().max())) < 5e-6
    # the raw-pointer update is visible to version-checked caches
    v0 = a[1]._version
    _grads(a, 6); oa.step()
    assert a[1]._version > v0
