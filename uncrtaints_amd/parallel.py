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

Every bucket is one flat buffer.  Autograd is left to hand each parameter its freshly computed gradient tensor (`.grad` is None
when the backward starts, so AccumulateGrad keeps the incoming tensor instead of launching one `grad += new` kernel per parameter:
91 launches, 0.3-0.4 ms per step, measured with tools/probe_grad_accumulate.py); when a bucket's last gradient has landed, ONE
multi-tensor copy packs the bucket (2.3 MB in total), the all-reduce goes out, and `.grad` of its parameters is pointed at the
bucket's views, which is what the optimizer then reads."""
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
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                p.grad = None
            self.buckets.append(dict(flat=flat, views=views, n=len(params), ready=0, handle=None, params=params, packed=False))
            for p in params:
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
        # same weights / buffers everywhere
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=0, group=process_group)
        agg = getattr(module, "temporal_aggregator", None)
        if agg is not None and hasattr(agg, "set_seed"):
            agg.set_seed(seed + self.rank)

    def _op(self):
        return dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM

    def _make_hook(self, bi: int) -> Callable:
        def hook(param):
            b = self.buckets[bi]
            b["ready"] += 1
            if self.overlap and b["ready"] == b["n"]:
                self.pack_bucket(bi)
                b["handle"] = dist.all_reduce(b["flat"], op=self._op(), group=self.pg, async_op=True)
        return hook

    def pack_bucket(self, bi: int) -> None:
        """Copy the gradients of bucket `bi` into its flat buffer (one multi-tensor launch) and point the parameters' `.grad` at
        the buffer's views.  A parameter without a gradient in this step contributes zeros (torch-DDP semantics); a gradient that
        already lives in the buffer (a caller that zeroed in place and let autograd accumulate) is left where it is.  Inside a
        captured segment (one HIP graph per bucket) the copy is part of the graph: call it at the end of the segment."""
        b = self.buckets[bi]
        src, dst, zero = [], [], []
        for p, v in zip(b["params"], b["views"]):
            g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr():
                if g.shape != v.shape or g.dtype != v.dtype:
                    raise RuntimeError("gradient shape / dtype does not match its parameter")
                src.append(g)
                dst.append(v)
        if zero:
            torch._foreach_zero_(zero)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(b["params"], b["views"]):
            p.grad = v
        b["packed"] = True

    # ---- segmented backward: forward+backward replayed from HIP graphs, one graph per bucket --------------------------------
    def reduce_bucket(self, bi: int, pack: bool = True) -> None:
        """Launch the all-reduce of bucket `bi` now (asynchronous; RCCL enqueues it behind the work already on the current
        stream).  For training loops that drive the backward pass in bucket-sized segments themselves -- e.g. one captured
        HIP graph per segment, where autograd hooks do not run on replay: segment k's gradients travel over xGMI while segment
        k+1 computes.  `finish()` then only waits.  pack=False: the segment packed the bucket itself (`pack_bucket` inside the
        captured graph -- on a replay the Python-side `.grad` attributes say nothing about what the graph just wrote)."""
        if self.overlap:
            # with overlap=True the post-accumulate hooks launch the same all-reduce on every eager step (warm-up, capture): a
            # second one here would sum the bucket twice (SUM + divide) and drop the first handle without a wait
            raise RuntimeError("reduce_bucket() drives the all-reduces by hand: construct BucketedDataParallel(overlap=False)")
        b = self.buckets[bi]
        if b["handle"] is not None:
            raise RuntimeError(f"bucket {bi} already has an all-reduce in flight (reduce_bucket called twice before finish())")
        if pack and not b["packed"]:
            self.pack_bucket(bi)
        b["handle"] = dist.all_reduce(b["flat"], op=self._op(), group=self.pg, async_op=True)
        b["ready"] = b["n"]

    def bucket_params(self, bi: int):
        return list(self.buckets[bi]["params"])

    def zero_grad(self):
        """Start of a step: every `.grad` back to None (no kernel: the backward hands over fresh tensors, see the module
        docstring).  `optimizer.zero_grad()` -- set_to_none or in place -- is equally fine."""
        for b in self.buckets:
            for p in b["params"]:
                p.grad = None
            b["ready"] = 0
            b["handle"] = None
            b["packed"] = False

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

    def finish(self, packed_by_graph: bool = False):
        """Wait for the in-flight all-reduces (call after backward, before the optimizer step); buckets nobody reduced yet are
        packed and reduced here.  packed_by_graph: see reduce_bucket(pack=False)."""
        timed = getattr(self, "time_waits", False) and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for bi, b in enumerate(self.buckets):
            if b["handle"] is None:          # not launched from a hook or through reduce_bucket()
                if not b["packed"] and not packed_by_graph:
                    self.pack_bucket(bi)
                b["handle"] = dist.all_reduce(b["flat"], op=self._op(), group=self.pg, async_op=True)
        for b in self.buckets:
            b["handle"].wait()
            if not self._use_avg:
                b["flat"].div_(self.world)
            b["ready"] = 0
            b["handle"] = None
            b["packed"] = False
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
