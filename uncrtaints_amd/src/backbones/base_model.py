"""Training wrapper with the reference's surface (model/src/backbones/base_model.py:10-131): holds netG,
the criterion, Adam and ExponentialLR; `optimize_parameters` runs forward -> zero_grad -> loss/backward
-> step -> rescale in the reference's order.  Device->host copies of the predictions (base_model.py:130-131)
are opt-in (`config.export_to_host`), they are a per-step sync used only for logging."""
import torch
import torch.nn as nn

from .. import losses, model_utils

S2_BANDS = 13


class BaseModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.frozen = False
        self.len_epoch = 0
        self.scale_by = config.scale_by
        self.netG = model_utils.get_generator(self.config)
        self.criterion = losses.get_loss(self.config)
        self.log_vars = None
        # config.hip_graph (opt-in, not a reference flag): optimize_parameters replays the whole step -- forward, loss, backward,
        # Adam, ~300 kernel launches -- from ONE captured HIP graph (see _graph_step).  Adam then keeps its step count and its
        # learning rate on the device (capturable), so that ExponentialLR's per-epoch update reaches the captured kernels.
        self._use_graph = bool(getattr(config, "hip_graph", False))
        self._graphs = {}            # launch-list key -> captured step (a few shapes, e.g. the smaller last batch of an epoch); LRU
        self._graph_cap = 6          # e.g. train / eval shapes plus the smaller last batch of each loader
        # base_model.py:48: torch.optim.Adam(params, lr).  FusedAdam IS that class (same state, checkpoints exchange with it) with the
        # update as one HIP launch over all parameters; on CPU parameters it runs torch's own step.
        from ...optim import FusedAdam
        if self._use_graph:
            self.optimizer_G = FusedAdam([{'params': self.netG.parameters()}], lr=torch.tensor(float(config.lr), device=config.device))
        else:
            self.optimizer_G = FusedAdam([{'params': self.netG.parameters()}], lr=config.lr)
        self.scheduler_G = torch.optim.lr_scheduler.ExponentialLR(self.optimizer_G, gamma=self.config.gamma)
        self.real_A = self.fake_B = self.real_B = self.dates = self.masks = None
        self.netG.variance = None
        self.data_parallel = None      # set to a uncrtaints_amd.parallel.BucketedDataParallel(self.netG) for multi-GPU training

    def forward(self):
        # config.hip_graph also covers the reference's validation / test loops (train_reconstruct.py:302-309 under the model.eval() of :692 / :734):
        # `model.eval(); with torch.no_grad(): model.set_input(...); model.forward()` replays the eval-mode forward from a captured
        # HIP graph (see _graph_forward); every other call -- the training forward, a forward with autograd on -- launches eagerly.
        if (self._use_graph and not torch.is_grad_enabled() and not self.netG.training and self.real_A is not None
                and self.real_A.is_cuda and not torch.cuda.is_current_stream_capturing()):
            return self._graph_forward()
        self.fake_B = self.netG(self.real_A, batch_positions=self.dates)
        self.netG.variance = None

    def backward_G(self):
        self.get_loss_G()
        self.loss_G.backward()

    def get_loss_G(self):
        mi, vi = self.netG.mean_idx, self.netG.vars_idx
        if vi > mi and self.fake_B.shape[2] == vi:
            # the slices of base_model.py:72-85 as one autograd node: the loss kernels read them in place and their backward hands
            # ONE gradient buffer back to the head (losses.split_prediction)
            mean, var = losses.split_prediction(self.fake_B, mi, vi)
        else:
            mean, var = self.fake_B[:, :, :mi, ...], self.fake_B[:, :, mi:vi, ...]
        self.loss_G, self.netG.variance = losses.calc_loss(self.criterion, self.config, mean, self.real_B, var=var)

    def set_input(self, input):
        dev = self.config.device
        self.real_A = self.scale_by * input['A'].to(dev)
        self.real_B = self.scale_by * input['B'].to(dev)
        self.dates = None if input['dates'] is None else input['dates'].to(dev)
        self.masks = input['masks'].to(dev) if input.get('masks') is not None else None

    def reset_input(self):
        self.real_A = self.real_B = self.dates = self.masks = None

    def rescale(self):
        if getattr(self, 'real_A', None) is not None:
            self.real_A = 1 / self.scale_by * self.real_A
        self.real_B = 1 / self.scale_by * self.real_B
        self.fake_B = 1 / self.scale_by * self.fake_B[:, :, :S2_BANDS, ...]
        if getattr(self.netG, 'variance', None) is not None:
            self.netG.variance = 1 / self.scale_by ** 2 * self.netG.variance

    # ---- the same step from a captured HIP graph (config.hip_graph) ----------------------------------------------------------
    # An eager step enqueues ~300 launches from Python: on a free host that keeps up with the GPU (12.3 ms at the bench shape,
    # tools/bench_basemodel.py), on a busy one it does not (18-20 ms measured in round 1); replayed from a graph the step takes the
    # GPU time whatever the host does.  The first two steps of a given input shape run eagerly (they are real training
    # steps and size every buffer); the third is captured and replayed, and so is every later one: the batch is copied into static
    # input tensors, the graph is launched, `fake_B`, `loss_G` (and the covariance export, if requested) are its static outputs.
    # The dropout stream of the aggregator draws its per-step seed from a device counter the graph increments.  Anything that changes
    # the launch list (another input shape, train/eval switch, freezing layers, a data-parallel wrapper) falls back to eager steps
    # and a fresh capture.
    def _graph_key(self):
        # state_epoch: bumped by FusedAdam.load_state_dict -- a captured step holds raw pointers to the moments it was captured with
        return (tuple(self.real_A.shape), tuple(self.real_B.shape), None if self.dates is None else tuple(self.dates.shape),
                self.netG.training, tuple(p.requires_grad for p in self.netG.parameters()),
                getattr(self.optimizer_G, "state_epoch", 0))

    def _graph_capable(self):
        """BatchNorm2d(momentum=None) derives its update factor from num_batches_tracked on the HOST (a sync, illegal in a capture and
        a constant in a replay): such models take eager steps."""
        ok = getattr(self, "_graph_ok", None)
        if ok is None:
            ok = self._graph_ok = not any(isinstance(m, nn.BatchNorm2d) and m.momentum is None for m in self.netG.modules())
        return ok

    def _device_lr(self):
        """The captured Adam kernel reads the learning rate from a device scalar (ExponentialLR then reaches the replays).  A
        checkpoint written by the reference or by an eager run restores a Python float (torch's load_state_dict replaces the
        param_group entries): FusedAdam.load_state_dict puts the value back into the device scalar; anything else that left a float
        there (a user assignment) is moved onto the device here, BEFORE a capture could bake it in by value."""
        for grp in self.optimizer_G.param_groups:
            lr = grp["lr"]
            if not (isinstance(lr, torch.Tensor) and lr.is_cuda and lr.dtype == torch.float32):
                grp["lr"] = torch.tensor(float(lr), dtype=torch.float32, device=self.real_A.device)
                self._graphs.clear()

    def _graph_step(self):
        self._device_lr()
        key = self._graph_key()
        if getattr(self.optimizer_G, "state_epoch", 0) != getattr(self, "_seen_state_epoch", 0):
            self._seen_state_epoch = getattr(self.optimizer_G, "state_epoch", 0)
            self._graphs.clear()           # every captured step points at the old exp_avg / exp_avg_sq buffers
        g = self._graphs.pop(key, None)
        if g is None:
            g = dict(key=key, eager_left=2, graph=None)
            while len(self._graphs) >= self._graph_cap:       # least recently used first (dicts keep insertion order)
                self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g              # (re-)inserted last = most recently used
        agg = getattr(self.netG, "temporal_aggregator", None)
        if agg is not None and agg.step_counter is None:
            agg.step_counter = torch.zeros(1, dtype=torch.int64, device=self.real_A.device)
        cur = torch.cuda.current_stream()
        if getattr(self, "_gstream", None) is None:
            # warm-up steps and capture share one side stream (autograd's AccumulateGrad nodes remember the stream they were
            # created on: a capture on another stream than the warm-up's would have to synchronise the two inside the capture)
            self._gstream = torch.cuda.Stream(device=self.real_A.device)
        if g["eager_left"] > 0:
            g["eager_left"] -= 1
            self._gstream.wait_stream(cur)
            with torch.cuda.stream(self._gstream):
                if agg is not None:
                    agg.step_counter.add_(1)
                self._eager_step()
            cur.wait_stream(self._gstream)
            return
        if g["graph"] is None:
            self._gstream.wait_stream(cur)
            with torch.cuda.stream(self._gstream):
                g["A"], g["B"] = self.real_A.clone(), self.real_B.clone()
                g["dates"] = None if self.dates is None else self.dates.clone()
                self.optimizer_G.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=self._gstream):
                    if agg is not None:
                        agg.step_counter.add_(1)
                    self.real_A, self.real_B, self.dates = g["A"], g["B"], g["dates"]
                    self.forward()
                    self.optimizer_G.zero_grad(set_to_none=True)
                    self.backward_G()
                    self.optimizer_G.step()
                    g["fake_B"], g["loss_G"], g["variance"] = self.fake_B, self.loss_G, self.netG.variance
            cur.wait_stream(self._gstream)
            g["graph"] = graph
        else:
            g["A"].copy_(self.real_A)
            g["B"].copy_(self.real_B)
            if g["dates"] is not None:
                g["dates"].copy_(self.dates)
        g["graph"].replay()
        self.real_A = None
        self.real_B = g["B"]
        self.fake_B, self.loss_G, self.netG.variance = g["fake_B"], g["loss_G"], g["variance"]
        self.rescale()          # new tensors: the static outputs stay untouched
        self.reset_input()
        self._export()

    # ---- eval-mode forward from a captured HIP graph (config.hip_graph) ---------------------------------------------------------
    # An eager eval forward is ~70 launches enqueued from Python; at B = 1 the GPU needs a fraction of the time the host takes to
    # enqueue them.  The first forward of a given input shape runs eagerly, the second is captured, later ones copy the batch into
    # the static input and replay.  What the replay reads through raw pointers -- parameters, BatchNorm running statistics -- is
    # updated IN PLACE by training steps and by load_state_dict, so validation after further training replays the same graph on the
    # new weights: the weight packing (engine.prepack / pack_wt) is part of the captured launch list (the version-keyed pack cache is
    # emptied before the capture, so nothing packed earlier is baked in).  The key holds the parameters' addresses: a model moved or
    # re-created is captured afresh.  `fake_B` is a copy of the static output (the caller may keep it across iterations).
    def _graph_forward(self):
        # everything that shapes the launch list: input / dates shapes and dtypes, the activation-storage mode, every sub-module's
        # train / eval flag (a block left in train() mode takes batch statistics), the engine's development switches, and the
        # parameters' addresses
        from ... import engine
        key = ("eval_forward", tuple(self.real_A.shape), self.real_A.dtype,
               None if self.dates is None else (tuple(self.dates.shape), self.dates.dtype),
               str(getattr(self.netG, "act_dtype", None)), tuple(m.training for m in self.netG.modules()),
               tuple(getattr(engine, v) for v in engine._DEV_OPTIONS.values()),
               tuple(p.data_ptr() for p in self.netG.parameters()))
        g = self._graphs.pop(key, None)
        if g is None:
            g = dict(key=key, eager_left=1, graph=None)
            while len(self._graphs) >= self._graph_cap:
                self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g
        if g["eager_left"] > 0 or g["graph"] is False:
            g["eager_left"] = max(g["eager_left"] - 1, 0)
            self.fake_B = self.netG(self.real_A, batch_positions=self.dates)
            self.netG.variance = None
            return
        cur = torch.cuda.current_stream()
        if g["graph"] is None:
            if getattr(self, "_gstream", None) is None:
                self._gstream = torch.cuda.Stream(device=self.real_A.device)
            self._gstream.wait_stream(cur)
            with torch.cuda.stream(self._gstream):
                g["A"] = self.real_A.clone()
                g["dates"] = None if self.dates is None else self.dates.clone()
                engine._PACK_CACHE.clear()
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                try:
                    with torch.cuda.graph(graph, stream=self._gstream):
                        g["out"] = self.netG(g["A"], batch_positions=g["dates"])
                except Exception as exc:      # noqa: BLE001 -- a model variant whose forward cannot be captured (a host sync in it)
                    import warnings
                    warnings.warn(f"config.hip_graph: the eval forward of this model could not be captured ({type(exc).__name__}: {exc}); "
                                  "validation forwards of this shape launch eagerly")
                    graph = False
            cur.wait_stream(self._gstream)
            g["graph"] = graph
            if graph is False:
                torch.cuda.synchronize()
                g.pop("out", None)
                self.fake_B = self.netG(self.real_A, batch_positions=self.dates)
                self.netG.variance = None
                return
        else:
            g["A"].copy_(self.real_A)
            if g["dates"] is not None:
                g["dates"].copy_(self.dates)
        g["graph"].replay()
        self.fake_B = g["out"].clone()
        self.netG.variance = None

    def _export(self):
        if self.netG.training and getattr(self.config, "export_to_host", False):
            self.fake_B = self.fake_B.cpu()
            if self.netG.variance is not None:
                self.netG.variance = self.netG.variance.cpu()

    def optimize_parameters(self):
        if self._use_graph and self.data_parallel is None and self.real_A is not None and self.real_A.is_cuda \
                and self._graph_capable():
            return self._graph_step()
        return self._eager_step()

    def _eager_step(self):
        self.forward()
        self.real_A = None
        # The reference's plain zero_grad() (base_model.py:117, set_to_none=True): a parameter without a gradient in a step is
        # skipped by Adam, and autograd hands every parameter its fresh gradient tensor instead of accumulating into an old one.
        # Under BucketedDataParallel (`self.data_parallel`, optional) the wrapper resets its bucket bookkeeping as well, packs and
        # all-reduces the buckets from its hooks, and is waited on before the optimizer step.
        dp = getattr(self, "data_parallel", None)
        if dp is not None:
            dp.zero_grad()
        else:
            self.optimizer_G.zero_grad()
        self.backward_G()
        if dp is not None:
            dp.finish()
        self.optimizer_G.step()
        self.rescale()
        self.reset_input()
        self._export()
