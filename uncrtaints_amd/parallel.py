"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; xGMI underneath), gradients all-reduced in a few buckets that are launched from autograd hooks
as soon as their last gradient is produced -- i.e. overlapped with the rest of the backward pass.

The reference has no distributed code at all; semantics follow torch-DDP defaults:
  * gradients are averaged over ranks;
  * BatchNorm statistics and the batch-summed MGNLL log-det are per replica -- `sync_bn=True` all-reduces the
    BatchNorm sums (forward and backward) instead, so N ranks compute what one process would on the concatenated batch;
  * BatchNorm buffers are broadcast from rank 0 at construction (and on demand via `sync_buffers`);
  * the aggregator's dropout stream is decorrelated per rank (seed + rank).

The gradient message is 570 010 fp32 = 2.28 MB: latency-bound on xGMI (a ring over 7 x ~153 GB/s links moves it
in tens of microseconds), so three buckets in reverse-graph order are plenty: the decoder's gradients are ready
first and go out while the encoder backward (the expensive half at T frames) is still running.

Gradients live as views into one flat buffer per bucket, so no packing copies are needed."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def default_buckets(named_params) -> List[List[str]]:
    """Reverse-graph order for UNCRTAINTS: decoder+head first, then L-TAE, then encoder."""
    names = [n for n, _ in named_params]
    b0 = [n for n in names if n.startswith("out_conv") or n.startswith("out_block")]
    b1 = [n for n in names if n.startswith("temporal_encoder")]
    b2 = [n for n in names if n not in set(b0) | set(b1)]
    return [b for b in (b0, b1, b2) if b]


class BucketedDataParallel:
    def __init__(self, module: torch.nn.Module, buckets: Optional[Sequence[Sequence[str]]] = None,
                 process_group=None, seed: int = 0, overlap: bool = True, sync_bn: bool = False):
        """overlap=True: all-reduces are launched from autograd hooks during backward.  overlap=False: nothing is
        launched from hooks and `finish()` reduces all buckets -- for a forward/backward that is replayed from a
        captured HIP graph (hooks do not run on replay, and collectives are kept out of the capture)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        from . import engine
        engine.set_sync_bn((process_group or dist.group.WORLD) if sync_bn else None)
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        byname = dict(named)
        if buckets is None:
            buckets = default_buckets(named)
        covered = [n for b in buckets for n in b]
        assert sorted(covered) == sorted(byname), "buckets must cover every trainable parameter exactly once"
        self.buckets = []
        self._pending = []
        self._use_avg = dist.get_backend(process_group) == "nccl"
        self.overlap = overlap
        for bi, names in enumerate(buckets):
            params = [byname[n] for n in names]
            total = sum(p.numel() for p in params)
            flat = torch.zeros(total, device=params[0].device, dtype=params[0].dtype)
            off = 0
            for p in params:
                p.grad = flat[off:off + p.numel()].view_as(p)     # gradients accumulate straight into the bucket
                off += p.numel()
            self.buckets.append(dict(flat=flat, n=len(params), ready=0, handle=None, params=params))
            for p in params:
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
        # same weights / buffers everywhere
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=0, group=process_group)
        agg = getattr(module, "temporal_aggregator", None)
        if agg is not None and hasattr(agg, "set_seed"):
            agg.set_seed(seed + self.rank)

    def _make_hook(self, bi: int) -> Callable:
        def hook(param):
            b = self.buckets[bi]
            b["ready"] += 1
            if self.overlap and b["ready"] == b["n"]:
                op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
                b["handle"] = dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=True)
        return hook

    # ---- segmented backward: forward+backward replayed from HIP graphs, one graph per bucket --------------------------------
    def reduce_bucket(self, bi: int) -> None:
        """Launch the all-reduce of bucket `bi` now (asynchronous; RCCL enqueues it behind the work already on the current
        stream).  For training loops that drive the backward pass in bucket-sized segments themselves -- e.g. one captured
        HIP graph per segment, where autograd hooks do not run on replay: segment k's gradients travel over xGMI while segment
        k+1 computes.  `finish()` then only waits."""
        if self.overlap:
            # with overlap=True the post-accumulate hooks launch the same all-reduce on every eager step (warm-up, capture): a
            # second one here would sum the bucket twice (SUM + divide) and drop the first handle without a wait
            raise RuntimeError("reduce_bucket() drives the all-reduces by hand: construct BucketedDataParallel(overlap=False)")
        b = self.buckets[bi]
        if b["handle"] is not None:
            raise RuntimeError(f"bucket {bi} already has an all-reduce in flight (reduce_bucket called twice before finish())")
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        b["handle"] = dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=True)
        b["ready"] = b["n"]

    def bucket_params(self, bi: int):
        return list(self.buckets[bi]["params"])

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()
            b["ready"] = 0
            b["handle"] = None

    def _check_views(self):
        """Every gradient must still be a view into its bucket: `optimizer.zero_grad()` (set_to_none=True, the torch
        default) or `p.grad = None` detaches them, after which the buckets would be reduced as stale zeros and every rank
        would step on its local gradients -- silently diverging replicas.  Raise instead."""
        for b in self.buckets:
            lo = b["flat"].data_ptr()
            hi = lo + b["flat"].numel() * b["flat"].element_size()
            for p in b["params"]:
                if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError("a parameter's .grad no longer points into its all-reduce bucket: use "
                                       "BucketedDataParallel.zero_grad() (or optimizer.zero_grad(set_to_none=False)), "
                                       "never set_to_none=True")

    def bucket_bytes(self):
        """bytes each all-reduce moves per rank (one flat fp32 bucket each)"""
        return [int(b["flat"].numel() * b["flat"].element_size()) for b in self.buckets]

    def wait_ms(self):
        """With `time_waits = True`: milliseconds the compute stream stood still in finish() waiting for the collectives, per
        recorded step (a HIP event pair around the waits) -- how much of the all-reduce was NOT hidden behind the backward."""
        ev = getattr(self, "_wait_events", [])
        if ev:
            torch.cuda.synchronize()
        out = [e0.elapsed_time(e1) for e0, e1 in ev]
        self._wait_events = []
        return out

    def finish(self):
        """Wait for the in-flight all-reduces (call after backward, before the optimizer step)."""
        self._check_views()
        timed = getattr(self, "time_waits", False) and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if not self.overlap:
            op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
            for b in self.buckets:
                if b["handle"] is None:          # not already launched through reduce_bucket()
                    b["handle"] = dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=True)
        for b in self.buckets:
            if b["handle"] is None:
                raise RuntimeError("a gradient bucket never became ready (parameter unused in this step?)")
            b["handle"].wait()
            if not self._use_avg:
                b["flat"].div_(self.world)
            b["ready"] = 0
            b["handle"] = None
        if timed:
            e1.record()
            if not hasattr(self, "_wait_events"):
                self._wait_events = []
            self._wait_events.append((e0, e1))

    def sync_buffers(self):
        for t in self.module.buffers():
            dist.broadcast(t.data, src=0, group=self.pg)

    def __call__(self, *a, **kw):
        return self.module(*a, **kw)
