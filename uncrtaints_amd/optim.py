"""The optimizer of the reference's train step (model/src/backbones/base_model.py:48: `torch.optim.Adam(params, lr=config.lr)`, stepped
in `optimize_parameters`, :118-131) with its update as ONE HIP launch over the whole parameter set (csrc/optim.hip).

`FusedAdam` IS a `torch.optim.Adam`: same constructor, same `param_groups`, same per-parameter `state` entries (`step`, `exp_avg`,
`exp_avg_sq`), so `state_dict()` / `load_state_dict()` exchange checkpoints with the stock class in both directions
(model_utils.py:117-196 saves and restores `optimizer_G`), and learning-rate schedulers work on it unchanged.  What differs is `step()`:
torch's multi-tensor path takes three launches of ~26 us plus two for the step counters for this model's 91 tensors; here the
addresses of (param, grad, exp_avg, exp_avg_sq) go into a small device table and one kernel updates everything.  Options the kernel
does not implement (amsgrad, maximize, differentiable, sparse or non-fp32 parameters, CPU parameters) fall back to torch's own step."""
from typing import List

import torch

from . import hip_backend as hb


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("fused", None)
        kw.pop("foreach", None)
        kw.pop("capturable", None)
        params = list(params)
        if params and isinstance(params[0], dict):      # (a group's "params" may be a generator: read it once)
            params = [dict(g, params=[g["params"]] if isinstance(g["params"], torch.Tensor) else list(g["params"])) for g in params]
            flat = [p for g in params for p in g["params"]]
        else:
            flat = params
        # capturable: the step counters live on the device, as the kernel (and a captured graph) needs them (GPU parameters only)
        on_gpu = bool(flat) and all(p.is_cuda for p in flat)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, capturable=on_gpu, **kw)
        self._uncr_tables = {}       # group index -> (key, desc, chunks, n_chunks, keep-alive)
        self._uncr_pinned: List[torch.Tensor] = []
        self.state_epoch = 0         # bumped whenever the state tensors are replaced (load_state_dict): captured steps hold raw pointers

    def load_state_dict(self, state_dict):
        """torch's load_state_dict replaces every param_group entry with the checkpoint's -- a checkpoint of the reference (or of an
        eager run) carries `lr` as a Python float.  A learning rate that lived on the device (graph mode: the captured kernel reads it,
        ExponentialLR updates it in place) stays THAT tensor and receives the loaded value; the moments are new tensors, so the address
        tables are dropped and `state_epoch` tells holders of captured steps (BaseModel._graph_step) to re-capture."""
        dev_lrs = [g["lr"] if isinstance(g["lr"], torch.Tensor) and g["lr"].is_cuda else None for g in self.param_groups]
        super().load_state_dict(state_dict)
        for g, t in zip(self.param_groups, dev_lrs):
            if t is not None:
                t.fill_(float(g["lr"]))
                g["lr"] = t
                if "initial_lr" in g and not isinstance(g["initial_lr"], torch.Tensor):
                    g["initial_lr"] = float(g["initial_lr"])
        self._uncr_tables.clear()
        self.state_epoch += 1

    # ------------------------------------------------------------------------------------------------------------------
    def _native_ok(self, group) -> bool:
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable"):
            return False
        for p in group["params"]:
            if p.grad is None:
                continue
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse or p.grad.dtype != torch.float32 \
                    or not p.is_contiguous() or not p.grad.is_contiguous():
                return False
        return True

    def _init_state(self, group, params):
        """State entries as torch.optim.Adam keeps them -- per parameter `step` (a device scalar of its own: torch's foreach update
        adds 1 to every entry of the list, shared tensors would be counted many times after a checkpoint exchange), `exp_avg`,
        `exp_avg_sq` -- with the moments of a group as views of two flat buffers (two fills instead of two per parameter)."""
        fresh = [p for p in params if len(self.state[p]) == 0]
        if fresh:
            dev = fresh[0].device
            total = sum(p.numel() for p in fresh)
            m, v = torch.zeros(total, device=dev), torch.zeros(total, device=dev)
            steps = torch.zeros(len(fresh), dtype=torch.float32, device=dev)
            off = 0
            for i, p in enumerate(fresh):
                n = p.numel()
                self.state[p]["step"] = steps[i]
                self.state[p]["exp_avg"] = m[off:off + n].view_as(p)
                self.state[p]["exp_avg_sq"] = v[off:off + n].view_as(p)
                off += n
        for p in params:        # after load_state_dict: whatever device / type the checkpoint's entries had
            st = self.state[p]
            if not (st["step"].is_cuda and st["step"].dtype == torch.float32):
                st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
            for k in ("exp_avg", "exp_avg_sq"):
                if not st[k].is_contiguous() or st[k].dtype != torch.float32 or st[k].device != p.device:
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
        return [self.state[p]["step"] for p in params]

    def _tables(self, gi, params):
        """Device tables of a batch of at most uncr_adam_max_tensors() parameters: addresses of (param, exp_avg, exp_avg_sq), the element
        count, the address of the parameter's OWN step counter (torch.optim.Adam counts steps per parameter: a layer unfrozen at epoch k
        starts its bias corrections at 1 while the others are at k * steps) and the chunk -> tensor map.  They change only when the state is re-created (first step, load_state_dict): built in an eager step."""
        key = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel(),
                     self.state[p]["step"].data_ptr()) for p in params)
        cached = self._uncr_tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2], cached[3]
        chunk = hb.query("uncr_adam_chunk")
        desc = [x for row in key for x in row]
        chunks = []
        for t, row in enumerate(key):
            for s in range(0, row[3], chunk):
                chunks += [t, s]
        dev = params[0].device
        # pinned host tables, copied without blocking; should this happen inside a graph capture (no eager step before it) the copies
        # become nodes of the graph, so the host buffers are kept alive, unchanged, for the life of the optimizer
        hd = torch.tensor(desc, dtype=torch.int64).pin_memory()
        hc = torch.tensor(chunks, dtype=torch.int32).pin_memory()
        d_desc = torch.empty(len(desc), dtype=torch.int64, device=dev)
        d_chunks = torch.empty(len(chunks), dtype=torch.int32, device=dev)
        d_desc.copy_(hd, non_blocking=True)
        d_chunks.copy_(hc, non_blocking=True)
        if torch.cuda.is_current_stream_capturing():
            self._uncr_pinned += [hd, hc]
        self._uncr_tables[gi] = (key, d_desc, d_chunks, len(chunks) // 2, (hd, hc))
        return d_desc, d_chunks, len(chunks) // 2

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._native_ok(g) for g in self.param_groups):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            steps = self._init_state(group, params)
            torch._foreach_add_(steps, 1.0)      # one launch; the kernel reads every tensor's own counter
            lr = group["lr"]
            lr_dev = lr if isinstance(lr, torch.Tensor) and lr.is_cuda else None
            if lr_dev is not None and lr_dev.dtype != torch.float32:
                raise TypeError("FusedAdam: a tensor learning rate must be a float32 CUDA scalar")
            b1, b2 = group["betas"]
            maxt = hb.query("uncr_adam_max_tensors")
            for bi in range(0, len(params), maxt):
                batch = params[bi:bi + maxt]
                desc, chunks, n_chunks = self._tables((gi, bi), batch)
                grads = torch.tensor([p.grad.data_ptr() for p in batch], dtype=torch.int64)      # host: read by the launcher
                hb.call("uncr_adam_step", desc, grads.data_ptr(), len(batch), chunks, n_chunks, 0.0 if lr_dev is not None else float(lr), lr_dev,
                        float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                        torch.cuda.current_stream().cuda_stream)
            # the kernel wrote the parameters through raw pointers: tell autograd and the version-checked caches (engine._PACK_CACHE)
            torch.autograd.graph.increment_version(params)
        return loss
