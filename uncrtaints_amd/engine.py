"""Kernel orchestration for the UnCRtainTS hot path on MI355X.

Every function here only allocates device buffers through torch and enqueues hand-written HIP kernels
(libuncr_hip.so, include/uncr_hip.h) on torch's current stream.  No torch compute ops are used on the
path; there is no CPU fallback.  Stages mirror SURVEY.md section 8(a):

    in_conv (a4) -> MBConv encoder (a5) -> max-pool (a6) + L-TAE attention (a7-a9) + temporal
    aggregation (a10) -> 5 x MBConv decoder (a5) -> out_conv + head (a2) ; MGNLL (a12)

Normalisation layers never run on their own: producers emit partial statistics, `norm_fwd` turns them
into per-(frame,channel) coefficients and the consuming kernel applies them in its prologue.
"""
from __future__ import annotations

import threading
import weakref

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import hip_backend as hb

PRO_NONE, PRO_AFFINE, PRO_AFFINE_GELU, PRO_NORMBWD, PRO_AFFINE_RELU = 0, 1, 2, 3, 4
NORM_GROUP, NORM_BATCH_TRAIN, NORM_BATCH_EVAL = 0, 1, 2
EW_STATS_SQ, EW_STATS_AUX, EW_AFFINE_RELU, EW_RESIDUAL, EW_PASSB, EW_PASSE, EW_RELU_BWD, EW_SE_POOL, \
    EW_HEAD_FWD, EW_HEAD_BWD = range(10)
EW_AFFINE, EW_NORMBWD, EW_SE_POOL4 = 15, 16, 17

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32(shape, dev) -> Tensor:
    return torch.empty(shape, device=dev, dtype=torch.float32)


# Activation storage (include/uncr_hip.h: UNCR_F32 / UNCR_BF16).  The mode is carried by the tensors themselves: every stage
# allocates its activation outputs in the storage type of its activation input, so casting the model input to bf16
# (UNCRTAINTS.act_dtype) switches the whole path -- "bf16 activations, fp32 accumulate", BASELINE config 3.  Statistics,
# coefficients, weights, weight gradients, the 32x32 L-TAE branch, the model outputs and the loss stay fp32.
F32, BF16 = 0, 1


def _dt(t: Tensor) -> int:
    return BF16 if t.dtype == torch.bfloat16 else F32


def _act(shape, dev, dt: int) -> Tensor:
    return torch.empty(shape, device=dev, dtype=torch.bfloat16 if dt == BF16 else torch.float32)


def cast(x: Tensor, dt: int) -> Tensor:
    """x converted to the storage type `dt` by a HIP kernel (no-op when it already has it)."""
    if _dt(x) == dt:
        return x
    x = x.contiguous()
    out = _act(x.shape, x.device, dt)
    hb.call("uncr_cast", x, out, x.numel(), _dt(x), dt, _stream())
    return out


def _check4(x: Tensor):
    if x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda:
        raise RuntimeError("expected a 4-D fp32 / bf16 CUDA tensor [frames, C, H, W]")
    N, C, H, W = x.shape
    if (H * W) % 1024 or W % 4:
        raise RuntimeError(f"unsupported spatial size {H}x{W}: H*W must be a multiple of 1024 and W of 4")
    return N, C, H, W


@dataclass
class Part:
    """Per-block partial statistics float2[N*C][slots] emitted by a producer kernel."""
    buf: Tensor
    slots: int
    masked: bool = False     # backward partials of a gradient that already carries the producer's ReLU mask (in_conv)
    amax: Optional[Tensor] = None   # [N][n] upper bounds on |value| per frame (per-block maxima of the producer), or None
    owner: tuple = ()        # (data_ptr, _version) of the tensor these partials describe (hand-offs between autograd nodes)
    centered: bool = False   # backward partials whose second component is sum du*(h - mean) (the producer was given the norm's mean)


def tag_part(t: Tensor, part: Optional["Part"], attr: str = "_uncr_bpart") -> None:
    """Attach partial statistics to the tensor they were computed from, for the next autograd node.  The tag records the
    tensor's address and version counter: `claim_part` hands the partials out only while both still match, so a gradient
    that autograd accumulated into (a second consumer, a tensor hook) is never paired with stale statistics."""
    if t is not None and part is not None:
        part.owner = (t.data_ptr(), t._version)
        setattr(t, attr, part)


def claim_part(t: Tensor, attr: str = "_uncr_bpart") -> Optional["Part"]:
    part = getattr(t, attr, None)
    if part is None or part.owner != (t.data_ptr(), t._version):
        return None
    return part


# ------------------------------------------------------------------------------------------------
# Any H x W (uncrtaints.py:391-447 takes any spatial size; csrc/anysize.hip).  For a size outside the tuned tilings the model keeps every
# full-resolution tensor as [frames, C, 1, Pc]: dense planes of H*W pixels + a ZERO tail up to Pc = uncr_any_plane_stride(H, W).  The flat
# kernels see planes of Pc pixels and run unchanged; while a `geom_scope` is active
#   * every finalisation takes the TRUE pixel count (`_pcount`),
#   * `fix_tail` behind a point-wise producer takes the tail's share out of its statistics and re-zeroes the tail,
#   * the 2-D kernels (depthwise, max-pool, aggregation) run their any-size variants on (H, W, Pc).
# The scope is entered by UNCRTAINTS.forward and re-entered by every backward from the geometry its forward saved.
# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Geom:
    H: int
    W: int
    Pc: int

    @property
    def P(self) -> int:
        return self.H * self.W

    @property
    def ntail(self) -> int:
        return self.Pc - self.H * self.W


# The active geometry lives in thread-local storage: autograd runs a device's backward on its own thread, and two models of
# different sizes (train 256 x 256, validate 250 x 250; two streams) may be in flight in one process.  Every backward re-enters the
# geometry its forward saved; nothing else about the scope is global (the kernels' variants follow from the arguments of each call).
_GEOM_TLS = threading.local()


def _geom() -> Optional[Geom]:
    return getattr(_GEOM_TLS, "geom", None)


def plan_geom(H: int, W: int) -> Optional[Geom]:
    """None when the tuned tilings take H x W as it is, else the padded-plane geometry."""
    pc = hb.query("uncr_any_plane_stride", H, W)
    if pc < 0:
        raise NotImplementedError(f"unsupported spatial size {H}x{W}")
    if pc == 0:
        return None
    return Geom(H, W, pc)


class geom_scope:
    """`with geom_scope(geom):` -- the calls inside (on this thread) treat planes of geom.Pc pixels as padded planes of an H x W image."""

    def __init__(self, geom: Optional[Geom]):
        self.geom = geom

    def __enter__(self):
        self.old = _geom()
        _GEOM_TLS.geom = self.geom
        return self.geom

    def __exit__(self, *exc):
        _GEOM_TLS.geom = self.old
        return False


def current_geom() -> Optional[Geom]:
    return _geom()


def _geom_for(P: int) -> Optional[Geom]:
    """the active geometry if planes of P pixels are its padded planes, else None"""
    g = _geom()
    return g if (g is not None and P == g.Pc) else None


def _dw_any_slots(geom, bwd: bool) -> int:
    slots = hb.query("uncr_dw_any_slots", geom.H, geom.W, 1 if bwd else 0)
    if slots <= 0:
        raise NotImplementedError(f"depthwise 3x3 on any-size planes: width {geom.W} is beyond the row-band kernels' 3998 (forward) / 2281 (backward)")
    return slots


def _pcount(P: int) -> int:
    """pixels of a plane that carry data: the image's H*W for a padded plane of the active geometry, else P itself"""
    g = _geom()
    return g.P if (g is not None and P == g.Pc) else P


def fix_tail(t: Tensor, part: Optional["Part"], mode: int, planes: int, unit: int = 1, aux: Optional[Tensor] = None,
             pivot: Optional[Tensor] = None) -> None:
    """Behind a pointwise GEMM on padded planes (geometry active, t's planes have the padded stride): add the valid pixels of the
    boundary `unit`-pixel tile -- which the GEMM's statistics epilogue left out, like every tile that reaches into the tail -- to
    `part` (mode 0: (sum, sum^2); 1: (sum, sum*aux); 2: none) and zero the tail."""
    g = _geom()
    if g is None or t.numel() != planes * g.Pc:
        return
    if t.dtype != torch.float32:
        raise NotImplementedError("any-size planes are built for fp32 storage")
    hb.call("uncr_fix_tail", t, aux if (part is not None and mode == 1) else None, part.buf if part is not None else None,
            part.slots if part is not None else 0, planes, g.P, g.Pc, mode if part is not None else 2, unit,
            pivot if (part is not None and mode == 1) else None, _stream())


def embed_tail(x: Tensor, geom: Geom) -> Tensor:
    """[planes..., H, W] dense fp32 -> [planes..., 1, Pc] with a zero tail"""
    lead = tuple(x.shape[:-2])
    planes = x.numel() // geom.P
    out = _f32(lead + (1, geom.Pc), x.device)
    hb.call("uncr_embed_tail", x.contiguous().float(), out, planes, geom.P, geom.Pc, _stream())
    return out


def extract_tail(x: Tensor, geom: Geom) -> Tensor:
    """[planes..., 1, Pc] -> dense [planes..., H, W]"""
    lead = tuple(x.shape[:-2])
    planes = x.numel() // geom.Pc
    out = _f32(lead + (geom.H, geom.W), x.device)
    hb.call("uncr_extract_tail", x.contiguous(), out, planes, geom.P, geom.Pc, _stream())
    return out


@dataclass
class NormSpec:
    kind: str        # 'group' | 'batch' | 'instance'
    groups: int = 4

    def code(self, training: bool) -> int:
        if self.kind in ("group", "instance"):      # nn.InstanceNorm2d (no affine, no running statistics) = one group per channel
            return NORM_GROUP
        if self.kind == "batch":
            return NORM_BATCH_TRAIN if training else NORM_BATCH_EVAL
        raise NotImplementedError(f"norm '{self.kind}' is not built (group | batch | instance)")

    def needs_stats(self, training: bool) -> bool:
        return self.code(training) != NORM_BATCH_EVAL


# ------------------------------------------------------------------------------------------------
# side chains: latency-bound launches that only feed parameter gradients run on a second HIP stream next to the bandwidth-bound
# kernels of the critical path (inside a captured step they become a parallel branch of the graph)
# ------------------------------------------------------------------------------------------------
# Opt-in (dev_options(side_stream=True)): measured neutral on MI355X inside the captured step (12.69 ms with, 12.70 ms without: the ~90 us
# of parameter-gradient launches do not overlap usefully with the bandwidth-bound kernels next to them), so the default keeps one stream.
_USE_SIDE = False
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}
_SIDE_TLS = threading.local()


class side_chain:
    """`with side_chain(t1, t2, ...):` -- the launches inside go to the device's side stream, ordered behind everything already
    enqueued on the current stream.  Rules that keep the caching allocator honest without record_stream(): every tensor the chain
    WRITES for later use is allocated before entering (on the current stream); the tensors it reads are passed as arguments and
    kept alive until `join_side()`, which makes the current stream wait for the chain and must be called before the caller
    returns (temporaries allocated inside belong to the side stream's pool)."""

    def __init__(self, *keep):
        self.keep = keep
        self.side = None

    def __enter__(self):
        if not _USE_SIDE:
            return self
        cur = torch.cuda.current_stream()
        dev = cur.device.index
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=cur.device)
        side.wait_stream(cur)
        self._ctx = torch.cuda.stream(side)
        self._ctx.__enter__()
        self.side = side
        return self

    def __exit__(self, *exc):
        if self.side is None:
            return False
        self._ctx.__exit__(*exc)
        pend = getattr(_SIDE_TLS, "pending", None)
        if pend is None:
            pend = _SIDE_TLS.pending = []
        pend.append((self.side, self.keep))
        return False


def join_side() -> None:
    """The current stream waits for every side chain this thread started since the last join."""
    pend = getattr(_SIDE_TLS, "pending", None)
    if not pend:
        return
    cur = torch.cuda.current_stream()
    seen = set()
    for side, _ in pend:
        if id(side) not in seen:
            cur.wait_stream(side)
            seen.add(id(side))
    pend.clear()


# ------------------------------------------------------------------------------------------------
# thin wrappers over the C ABI
# ------------------------------------------------------------------------------------------------

def ew(op: int, a: Tensor, *, b=None, c=None, aux=None, out: Optional[Tensor] = None, k=(None, None, None, None),
       want_part: bool = False, planes: int, P: int, C: int = 1, n_mean: int = 0, scale: float = 1.0,
       eps: float = 0.0) -> Tuple[Optional[Tensor], Optional[Part]]:
    part = None
    if want_part:
        slots = hb.query("uncr_ew_slots", P)
        part = Part(_f32((planes, slots, 2), a.device), slots)
    hb.call("uncr_ew", op, a, b, c, aux, out, k[0], k[1], k[2], k[3], part.buf if part else None, planes, P, C,
            n_mean, float(scale), float(eps), _dt(out if out is not None else a), _pcount(P), _stream())
    return out, part


def se_pool(h2: Tensor, A: Tensor, B: Tensor, planes: int, P: int) -> Part:
    """Partials of sum_p gelu(A*h2 + B) per plane (SE squeeze, uncrtaints.py:82-97): P / 1024 slots per plane either way."""
    op = EW_SE_POOL4 if (_SE_POOL4 and P % 4096 == 0 and _dt(h2) == BF16) else EW_SE_POOL
    return ew(op, h2, k=(A, B, None, None), want_part=True, planes=planes, P=P)[1]


def stats_sq(x: Tensor, planes: int, P: int) -> Part:
    return ew(EW_STATS_SQ, x, want_part=True, planes=planes, P=P)[1]


def stats_aux(a: Tensor, b: Tensor, planes: int, P: int) -> Part:
    return ew(EW_STATS_AUX, a, b=b, want_part=True, planes=planes, P=P)[1]


@dataclass
class NormFwd:
    A: Tensor
    B: Tensor
    mean: Tensor
    rstd: Tensor
    kind: int
    groups: int
    sync_count: float = 0.0     # > 0: BatchNorm statistics were all-reduced over the data-parallel group (global count)
    ub: Optional[Tensor] = None  # [N*C] upper bounds on |A*h + B| per plane (range bookkeeping of the fp16 two-part GEMMs), or None
    hb: Optional[Tensor] = None  # [N*C] upper bounds on |h| itself (for the gradient GEMMs that read h through a norm backward)
    # consumer-side finalisation (norm_fwd(defer=True)): nothing has been launched yet; the kernel that applies this norm fills A, B,
    # mean, rstd (ub, hb) itself from (part.buf, part.slots, gamma, beta, running_mean, running_var, momentum, eps)
    fin: Optional[tuple] = None


_SYNC_BN = None      # process group for synchronised BatchNorm statistics (None: per-replica statistics, torch-DDP default)


def set_sync_bn(group) -> None:
    """group: a torch.distributed process group (e.g. dist.group.WORLD) or None.  With a group, every train-mode BatchNorm
    on the path all-reduces its per-channel (sum, sum^2) and, in backward, (sum du, sum du*h): N ranks then compute what one
    process would on the concatenated batch.  Equal local batch sizes are assumed."""
    global _SYNC_BN
    _SYNC_BN = group


def _all_reduce_sums(sums: Tensor) -> float:
    import torch.distributed as dist
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_SYNC_BN)
    return float(dist.get_world_size(_SYNC_BN))


def norm_fwd(part: Optional[Part], N: int, C: int, P: int, spec: NormSpec, training: bool, gamma: Tensor,
             beta: Tensor, running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None,
             momentum: float = 0.1, eps: float = 1e-5, bound_part: Optional[Part] = None, want_hb: bool = False,
             defer: bool = False, src: Optional[Tensor] = None) -> NormFwd:
    """bound_part: (sum h, sum h^2) partials of the tensor this norm is applied to (in train mode `part` itself; in BatchNorm eval
    mode they serve only this): the finalize kernel then also emits per-plane upper bounds on |A*h + B| (NormFwd.ub), with which
    the consuming wide GEMM multiplies in two range-safe fp16 parts instead of the exact bf16 split (pw_gemm).
    defer: the caller's next kernel can finalise a train-mode BatchNorm itself (csrc/bn_inline.h): where that applies nothing is
    launched here and NormFwd.fin carries what that kernel needs (the caller MUST then run such a kernel); otherwise as usual.
    src: the tensor the statistics are of: train-mode statistics sets 8 sigma or more from zero are re-read once and take their second
    moment about the mean (uncr_norm_finalize_fwd's src; the consumer-side finalisation does the same with its own input from 32 sigma).
    InstanceNorm2d PreNorm in addition: planes far from zero in units of their own spread (|mean| > 8 sigma:
    an un-normalised tensor, e.g. the decoder's first PreNorm behind an eval-mode BatchNorm encoder) get their statistics recomputed
    from it about the mean (uncr_instance_repair); every other plane costs its block two loads."""
    kind = spec.code(training)
    groups = C if spec.kind == "instance" else spec.groups
    if gamma is None:                      # norm without affine parameters (InstanceNorm2d): gamma = 1, beta = 0
        gamma, beta = _const_planes((part.buf if part is not None else running_mean).device, C)
    dev = gamma.device
    A, B = _f32((N * C,), dev), _f32((N * C,), dev)
    nstat = N * groups if kind == NORM_GROUP else C
    mean, rstd = _f32((nstat,), dev), _f32((nstat,), dev)
    if kind == NORM_BATCH_TRAIN and _SYNC_BN is not None:
        sums = torch.empty((C, 2), device=dev, dtype=torch.float64)
        hb.call("uncr_bn_channel_sums", part.buf, part.slots, N, C, sums, _stream())
        count = _all_reduce_sums(sums) * N * _pcount(P)
        ub = _f32((N * C,), dev) if (bound_part is part and _H2_FWD) else None
        hbt = _f32((N * C,), dev) if (ub is not None and want_hb) else None
        csums = None
        if src is not None and _STATS_REPAIR:
            # channels the GLOBAL raw sums put 8 sigma or more from zero: local sums about the global mean, a second (small) all-reduce
            csums = torch.empty((C, 2), device=dev, dtype=torch.float64)
            hb.call("uncr_bn_centred_sums", sums, count, src, N, C, _pcount(P), P, _dt(src), csums, _stream())
            _all_reduce_sums(csums)
        hb.call("uncr_bn_finalize_fwd_sums", sums, count, N, C, gamma, beta, running_mean, running_var, float(momentum),
                float(eps), A, B, mean, rstd, part.buf if ub is not None else None, part.slots if ub is not None else 0, ub, hbt,
                csums, _stream())
        return NormFwd(A, B, mean, rstd, kind, groups, sync_count=count, ub=ub, hb=hbt)
    ub = None
    if bound_part is not None and _H2_FWD and (kind != NORM_GROUP or C // groups <= 256):
        if part is None:
            part = bound_part
        if part is bound_part:
            ub = _f32((N * C,), dev)
    hbt = _f32((N * C,), dev) if (ub is not None and want_hb) else None
    if defer and _BN_CONSUMER and kind == NORM_BATCH_TRAIN and part is not None and (ub is None or part is bound_part):
        return NormFwd(A, B, mean, rstd, kind, groups, ub=ub, hb=hbt,
                       fin=(part.buf, part.slots, gamma, beta, running_mean, running_var, float(momentum), float(eps)))
    hb.call("uncr_norm_finalize_fwd", part.buf if part else None, part.slots if part else 0, N, C, groups, _pcount(P),
            kind, gamma, beta, running_mean, running_var, float(momentum), float(eps), A, B, mean, rstd, ub, hbt,
            src if (_STATS_REPAIR and part is not None) else None, P, _dt(src) if src is not None else F32, _stream())
    if src is not None and spec.kind == "instance" and _INSTANCE_REPAIR:
        hb.call("uncr_instance_repair", src, N, C, _pcount(P), P, gamma, beta, float(eps), A, B, mean, rstd, ub, hbt, _dt(src),
                _stream())
    return NormFwd(A, B, mean, rstd, kind, groups, ub=ub, hb=hbt)


# Development switches.  The product path never reads the process environment: these are module constants with the shipped values,
# and tools / tests flip them for an A/B run through the explicit `dev_options(...)` context below.
_CENTRED_NORMBWD = True       # False: raw norm backward dh = c1*du + c2*h + c3 instead of the centred form
# fp16 two-part split in the dz GEMM of an MBConv backward, scaled per frame from the producers' magnitude bookkeeping
# (False: exact bf16 split there, no bookkeeping)
_H2_BWD = True
_H2_WGRAD = True              # the dW2 weight-gradient products alone
_H2_DX = True                 # the dx GEMM (+ max |du1| in the depthwise backward) alone
# the same split in the forward GEMMs behind a norm (pw1 / pw2 of an MBConv), scaled per frame from the bounds the statistics
# finalisation emits (False: exact bf16 split)
_H2_FWD = True
_PREPACK = True               # False: every pack_wt call packs on its own (bisecting)
_DW_VARIANT = 0      # uncr_dw_fwd / uncr_dw_bwd `variant`: 0 = automatic, 1 = LDS-tiled kernels for every width (tests)
# train-mode BatchNorm finalised by the kernel that applies it, where that kernel works on whole planes (csrc/bn_inline.h);
# False: one uncr_norm_finalize_fwd launch per norm (A/B runs, bisecting)
_BN_CONSUMER = True
# in_conv (Conv2d k=1 + GroupNorm + ReLU on <= 15 input channels) without its pre-norm tensor: statistics and parameter gradients
# from the frames' second-moment matrices (csrc/inconv.hip); False: GEMM -> finalize -> element-wise pass, c0 kept for the backward
_INCONV_MOMENTS = True
# SE pooling pass of the bf16-storage mode with four chunks of a plane per block (four 8-byte loads in flight per lane; bit-identical
# partials): -16 ... -20 % on the kernel behind a producer (tools/bench_sepool.py), -0.03 ms on the bf16 step (3 of 3 interleaved
# pairs).  fp32 storage keeps one chunk per block (the four-chunk kernel is 6-7 % SLOWER there).  False: one chunk per block everywhere
_SE_POOL4 = True
# inference: an eval-mode MBConv's closing BatchNorm + skip in the epilogue of its pw2 GEMM (uncr_pw_gemm epi 10); False: GEMM, then the
# element-wise residual pass (tests: the two are bit-identical in fp32 storage)
_EVAL_TAIL = True
# InstanceNorm PreNorm: pw1's weight-gradient products on x - mean (False: raw x, bisecting)
_CENTRED_PW1 = True
# in_conv's backward statistics centred on the norm's mean (False: raw sum du0*c0, bisecting)
_CENTRED_INCONV = True
# InstanceNorm PreNorm: statistics of planes with |mean| > 8 sigma recomputed about the mean (False: raw moments only, bisecting)
_INSTANCE_REPAIR = True
# statistics sets with |mean| >= 8 sigma re-read by the finalisation kernels (False: raw moments only; the consumer-side finalisation
# of uncr_dw_fwd_bn keeps its own re-read: switch bn_consumer off as well to bisect)
_STATS_REPAIR = True

# development (tools/ablate_ltae_stage.py): "record" keeps the L-TAE stage's results of the next forward / backward, "replay" hands
# them back without launching anything -- the stage's cost inside the captured step = step time with it minus step time without it
_LTAE_REPLAY = None
_LTAE_STORE: Dict[str, tuple] = {}

_DEV_OPTIONS = {"ltae_replay": "_LTAE_REPLAY", "side_stream": "_USE_SIDE", "centred_normbwd": "_CENTRED_NORMBWD", "h2_bwd": "_H2_BWD", "h2_wgrad": "_H2_WGRAD",
                "h2_dx": "_H2_DX", "h2_fwd": "_H2_FWD", "prepack": "_PREPACK", "fused_dx": "_FUSED_DX", "fused_ltae": "_FUSED_LTAE",
                "dw_variant": "_DW_VARIANT", "bn_consumer": "_BN_CONSUMER", "inconv_moments": "_INCONV_MOMENTS", "se_pool4": "_SE_POOL4", "eval_tail": "_EVAL_TAIL", "agg_two_pass": "_AGG_TWO_PASS",
                "centred_pw1": "_CENTRED_PW1", "centred_inconv": "_CENTRED_INCONV",
                "instance_repair": "_INSTANCE_REPAIR", "stats_repair": "_STATS_REPAIR"}


class dev_options:
    """`with engine.dev_options(h2_fwd=False, fused_dx=False): ...` -- flip development switches of the Python layer for the duration
    of an A/B run or a test (tools/, tests/) and restore them afterwards.  Nothing here is read from the environment, and the
    library below (include/uncr_hip.h) keeps no switches at all: a kernel variant follows from the arguments of each call."""

    def __init__(self, **opts):
        unknown = set(opts) - set(_DEV_OPTIONS)
        if unknown:
            raise KeyError(f"unknown development option(s) {sorted(unknown)}; known: {sorted(_DEV_OPTIONS)}")
        self.opts, self.old = opts, {}

    def __enter__(self):
        g = globals()
        for k, v in self.opts.items():
            self.old[k] = g[_DEV_OPTIONS[k]]
            g[_DEV_OPTIONS[k]] = v
        return self

    def __exit__(self, *exc):
        g = globals()
        for k, v in self.old.items():
            g[_DEV_OPTIONS[k]] = v
        return False


@dataclass
class NormBwd:
    """dh = c1*du + c2*(h - mu) + c3 per plane (centred form: the constant carries no rounding offset of size |c2*mean|)."""
    c1: Tensor
    c2: Tensor
    c3: Tensor
    dgamma: Tensor
    dbeta: Tensor
    mu: Optional[Tensor] = None

    @property
    def k(self):
        return (self.c1, self.c2, self.c3, self.mu)


def plane_means(nf: NormFwd, N: int, C: int, scale: float = 1.0) -> Tensor:
    """[N*C]: scale * the mean of the statistics set each plane belongs to (pivots of centred statistics / products)"""
    out = _f32((N * C,), nf.mean.device)
    hb.call("uncr_plane_means", nf.mean, N, C, nf.groups if nf.kind == NORM_GROUP else 0, float(scale), out, _stream())
    return out


def norm_bwd(part: Part, N: int, C: int, P: int, nf: NormFwd, gamma: Tensor, centered: bool = False) -> NormBwd:
    """centered: the partials' second component is sum du*(h - mean) (producer was given nf.mean)."""
    if gamma is None:
        gamma = _const_planes(part.buf.device, C)[0]
    dev = gamma.device
    c1, c2, c3 = _f32((N * C,), dev), _f32((N * C,), dev), _f32((N * C,), dev)
    mu = _f32((N * C,), dev) if _CENTRED_NORMBWD else None      # None: raw form dh = c1*du + c2*h + c3 (A/B switch)
    dg, db = _f32((C,), dev), _f32((C,), dev)
    if nf.kind == NORM_BATCH_TRAIN and nf.sync_count > 0:
        loc = torch.empty((C, 2), device=dev, dtype=torch.float64)
        hb.call("uncr_bn_channel_sums", part.buf, part.slots, N, C, loc, _stream())
        glob = loc.clone()
        _all_reduce_sums(glob)
        hb.call("uncr_bn_finalize_bwd_sums", loc, glob, nf.sync_count, N, C, gamma, nf.mean, nf.rstd, c1, c2, c3, mu, dg, db,
                1 if centered else 0, _stream())
        return NormBwd(c1, c2, c3, dg, db, mu)
    scratch = _f32((2 * N * C,), dev) if nf.kind == NORM_GROUP else None
    hb.call("uncr_norm_finalize_bwd", part.buf, part.slots, N, C, nf.groups, _pcount(P), nf.kind, gamma, nf.mean, nf.rstd,
            c1, c2, c3, mu, dg, db, scratch, 1 if centered else 0, _stream())
    return NormBwd(c1, c2, c3, dg, db, mu)


# ---- stand-alone norm layer and squeeze-excite (PreNorm uncrtaints.py:72-79 / SE uncrtaints.py:82-97 called on their own;
#      inside MBConv both are prologues of other kernels) ----
def norm_apply_forward(x: Tensor, spec: NormSpec, training: bool, gamma: Optional[Tensor], beta: Optional[Tensor],
                       running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None, momentum: float = 0.1,
                       eps: float = 1e-5):
    """GroupNorm / BatchNorm2d / InstanceNorm2d of x [N,C,H,W]: one statistics pass, the finalize kernel, one affine pass."""
    N, C, H, W = _check4(x)
    P = H * W
    x = x.contiguous()
    part = stats_sq(x, N * C, P) if spec.needs_stats(training) else None
    nf = norm_fwd(part, N, C, P, spec, training, gamma, beta, running_mean, running_var, momentum, eps, src=x)
    out = _act((N, C, H, W), x.device, _dt(x))
    ew(EW_AFFINE, x, out=out, k=(nf.A, nf.B, None, None), planes=N * C, P=P)
    return out, dict(x=x, nf=nf, dims=(N, C, H, W))


def norm_apply_backward(dy: Tensor, sv: dict, gamma: Optional[Tensor], need_dx: bool = True):
    N, C, H, W = sv["dims"]
    P = H * W
    dy = dy.contiguous()
    x, nf = sv["x"], sv["nf"]
    nb = norm_bwd(stats_aux(dy, x, N * C, P), N, C, P, nf, gamma)
    dx = None
    if need_dx:
        dx = _act((N, C, H, W), dy.device, _dt(dy))
        ew(EW_NORMBWD, dy, b=x, out=dx, k=nb.k, planes=N * C, P=P)
    return dx, nb.dgamma, nb.dbeta


def se_forward(x: Tensor, W1: Tensor, W2: Tensor):
    """x * sigmoid(W2 gelu(W1 mean_p x)) for x [N,C,H,W] (uncrtaints.py:92-97).  C <= 256, hidden <= 64."""
    N, C, H, W = _check4(x)
    P, R = H * W, W1.shape[0]
    x = x.contiguous()
    pp = stats_sq(x, N * C, P)                          # (sum x, sum x^2) partials; the MLP kernel takes the first component
    pooled, hid_pre, s = _f32((N, C), x.device), _f32((N, R), x.device), _f32((N * C,), x.device)
    hb.call("uncr_se_mlp_fwd", pp.buf, pp.slots, N, C, R, P, W1.contiguous(), W2.contiguous(), pooled, hid_pre, s, _stream())
    out = _act((N, C, H, W), x.device, _dt(x))
    zero = torch.zeros(N * C, device=x.device, dtype=torch.float32)
    ew(EW_AFFINE, x, out=out, k=(s, zero, None, None), planes=N * C, P=P)
    return out, dict(x=x, s=s, pooled=pooled, hid_pre=hid_pre, dims=(N, C, H, W, R))


def se_backward(dy: Tensor, sv: dict, W1: Tensor, W2: Tensor, need_dx: bool = True):
    """ds[n,c] = sum_p dy*x runs through the SE-MLP backward kernel as a one-row product (G [N,1,C], unit weights);
    dx = s*dy + dpool/P."""
    N, C, H, W, R = sv["dims"]
    P = H * W
    dev = dy.device
    dy = dy.contiguous()
    pp = stats_aux(dy, sv["x"], N * C, P)
    G = _f32((N, 1, C), dev)
    hb.call("uncr_part_sums", pp.buf, pp.slots, N * C, None, G, _stream())
    ones = torch.ones(1, C, device=dev, dtype=torch.float32)
    ds_pre, dhid_pre, dpool = _f32((N, C), dev), _f32((N, R), dev), _f32((N * C,), dev)
    dWpw, dW1, dW2 = _f32((1, C), dev), _f32((R, C), dev), _f32((C, R), dev)
    hb.call("uncr_se_mlp_bwd", G, ones, N, 1, C, R, P, W1.contiguous(), W2.contiguous(), sv["s"], sv["pooled"], sv["hid_pre"],
            ds_pre, dhid_pre, dpool, dWpw, dW1, dW2, _stream())
    dx = None
    if need_dx:
        dx = _act((N, C, H, W), dev, _dt(dy))
        ew(EW_AFFINE, dy, out=dx, k=(sv["s"], dpool, None, None), planes=N * C, P=P)
    return dx, dW1, dW2


# ---- batched weight packing: every 1x1-conv weight of a model in ONE launch at the start of a forward ----
# (data_ptr, transpose, R, Cc) -> (packed view, weight version, the weight view itself).  The entry OWNS a reference to the
# weight: while it exists the allocator cannot hand the same address to another tensor, so a (pointer, version) match can
# only be the tensor that was packed (a freed model's address re-used by a new model's weights would otherwise hit).
_PACK_CACHE: Dict[tuple, tuple] = {}
_PACK_PLANS: Dict[tuple, tuple] = {}     # plans of callers without an owner: key -> (descriptor, flat output, views)
# plans of a module (prepack(owner=...)): held here, keyed weakly by the module, NOT in the module's __dict__ -- copy.deepcopy(model)
# and torch.save(model) would otherwise carry the device buffers and their raw weight pointers along
_OWNER_PLANS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def prepack(weights, owner=None) -> None:
    """weights: iterable of (W2d [R][Ccols] contiguous view of a parameter, transpose).  Packs all of them with one
    kernel launch into a per-plan static buffer and remembers the results for `pack_wt` until the next `prepack`
    (entries are ignored when the parameter's version counter has moved, i.e. after an in-place update).
    owner: the module the weights belong to.  Its plans (train / eval pack lists) live in `_OWNER_PLANS[owner]` for as
    long as the module does and are never evicted: a captured HIP graph holds raw pointers into a plan's buffers, so a
    plan must not be freed while its model can still be replayed."""
    weights = [(w, bool(tr)) for w, tr in weights if w.is_contiguous()]
    if not weights or not _PREPACK:
        return
    key = tuple((w.data_ptr(), tr, w.shape[0], w.shape[1]) for w, tr in weights)
    plans = _PACK_PLANS
    if owner is not None:
        plans = _OWNER_PLANS.setdefault(owner, {})
    plan = plans.get(key)
    if plan is None:
        dev = weights[0][0].device
        sizes, dims = [], []
        for w, tr in weights:
            R, Cc = w.shape
            rows_k, cols_co = (Cc, R) if tr else (R, Cc)
            nf = hb.query("uncr_pw_wt_floats", rows_k, cols_co)
            if nf <= 0:
                raise NotImplementedError(f"1x1 convolution {rows_k} -> {cols_co}: the GEMM kernels are built for at most 256 "
                                          "input and 256 output channels")
            sizes.append((nf + 3) // 4 * 4)     # keep 16-B alignment
            dims.append((rows_k, cols_co, Cc))
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        views, rows, off = [], [], 0
        for (w, tr), n, (rows_k, cols_co, ld) in zip(weights, sizes, dims):
            v = flat[off:off + n]
            views.append(v)
            rows.append([w.data_ptr(), v.data_ptr(), rows_k, cols_co, ld, 1 if tr else 0, 0, 0])
            off += n
        desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        plan = (desc, flat, views)
        plans[key] = plan
    desc, _, views = plan
    hb.call("uncr_pack_wt_batch", desc, len(weights), _stream())
    _PACK_CACHE.clear()
    for (w, tr), v, k in zip(weights, views, key):
        _PACK_CACHE[k] = (v, w._version, w)
    while len(_PACK_PLANS) > 8 and not torch.cuda.is_current_stream_capturing():   # owner-less plans only
        _PACK_PLANS.pop(next(iter(_PACK_PLANS)))


def pack_wt(W2d: Tensor, transpose: bool) -> Tensor:
    """W2d [R][Ccols] -> the packed, zero-padded weight operand of pw_gemm for Wt[k][co] = W[co][k] (transpose) or
    W[k][co] (opaque layout: fp32 [Kpad][COUTP] for Cout <= 64, pre-split bf16 MFMA fragments otherwise)."""
    W2d = W2d.contiguous()
    R, Cc = W2d.shape
    hit = _PACK_CACHE.get((W2d.data_ptr(), bool(transpose), R, Cc))
    if hit is not None and hit[1] == W2d._version:
        return hit[0]
    rows_k, cols_co = (Cc, R) if transpose else (R, Cc)
    nf = hb.query("uncr_pw_wt_floats", rows_k, cols_co)
    if nf <= 0:
        raise NotImplementedError(f"1x1 convolution {rows_k} -> {cols_co}: the GEMM kernels are built for at most 256 input and "
                                  "256 output channels")
    out = _f32((nf,), W2d.device)
    hb.call("uncr_pack_wt", W2d, rows_k, cols_co, Cc, 1 if transpose else 0, out, _stream())
    return out


def pw_gemm(x: Tensor, Wt: Tensor, N: int, Cin: int, Cout: int, P: int, *, pro: int = PRO_NONE, k=(None, None, None),
            x2: Optional[Tensor] = None, bias: Optional[Tensor] = None, bias_per_frame: bool = False, epi: int = 0,
            aux: Optional[Tensor] = None, out: Optional[Tensor] = None,
            ek=(None, None, None, None), out_dt: Optional[int] = None, want_amax: bool = False,
            in_amax: Optional[Tensor] = None, in2_amax: Optional[Tensor] = None, dense: bool = False) -> Tuple[Tensor, Optional[Part]]:
    """dense: the planes are NOT padded planes of the active any-size geometry whatever their pixel count (the dense 3x3 convolution's
    padded grids, which may happen to have the geometry's stride).
    out_dt: storage of the output (default: that of the input; the narrow Cout <= 64 kernels write fp32 only).
    want_amax (epi 1 / 2, 64 < Cout <= 128, fp32): the returned Part carries per-block maxima of |out| ([N][slots]).
    in_amax / in2_amax ([N][n] each): magnitude bounds of the two NORMBWD operands; with both, the epi-3 GEMM of an MBConv backward
    multiplies in two scaled fp16 parts (three products) instead of the exact bf16 split (six).
    in_amax alone with pro AFFINE / AFFINE_GELU (epi 0 / 1, Cout > 64, fp32): NormFwd.ub of the norm in the prologue ([N*Cin] bounds
    on |A*h + B|): the forward GEMM takes the same two-part fp16 route, scaled per frame from the bound."""
    if out is None:
        out = _act((N, Cout, P), x.device, _dt(x) if out_dt is None else out_dt)
    part = None
    if epi and epi != 4:          # epi 4 = accumulate into `out`, no statistics
        slots = hb.query("uncr_pw_stat_slots", N, Cout, P)
        if slots <= 0:
            raise RuntimeError(f"pw_gemm: P={P} is not a multiple of the {hb.query('uncr_pw_tile_px', Cout)}-pixel tile")
        part = Part(_f32((N * Cout, slots, 2), x.device), slots)
    amax = None
    if want_amax and part is not None and epi in (1, 2) and 64 < Cout <= 128 and _dt(x) == F32 and _dt(out) == F32 and _H2_BWD:
        amax = _f32((N, part.slots), x.device)
        part.amax = amax
    fp32_wide = Cout > 64 and _dt(x) == F32 and _dt(out) == F32
    if pro == PRO_NORMBWD:
        use_in = in_amax is not None and in2_amax is not None and _H2_BWD and epi == 3 and fp32_wide
        n1, n2 = (in_amax.shape[1], in2_amax.shape[1]) if use_in else (0, 0)
    else:
        use_in = in_amax is not None and _H2_FWD and pro in (PRO_AFFINE, PRO_AFFINE_GELU) and epi in (0, 1, 10) and fp32_wide
        if use_in and in_amax.numel() != N * Cin:
            raise RuntimeError("pw_gemm: in_amax of an affine prologue must hold one bound per (frame, input channel)")
        n1, n2, in2_amax = (Cin if use_in else 0), 0, None
    hb.call("uncr_pw_gemm", x, x2, Wt, out, k[0], k[1], k[2], k[3] if len(k) > 3 else None, bias,
            Cout if bias_per_frame else 0, aux,
            ek[0], ek[1], ek[2], ek[3], part.buf if part else None, N, Cin, Cout, P, pro, epi, _dt(x), _dt(out),
            amax, in_amax if use_in else None, n1, in2_amax if use_in else None, n2, P if dense else _pcount(P), _stream())
    if not dense and _geom_for(P) is not None and epi != 4 and not (epi == 0 and pro == PRO_NONE and bias is None):
        # padded planes of an any-size image: the statistics left out the tiles that reach into the tail, which holds f(0) now
        # (a plain product of zero columns without a bias leaves zeros: nothing to do then)
        fix_tail(out, part, 1 if epi in (2, 3) else (0 if part is not None else 2), N * Cout, hb.query("uncr_pw_tile_px", Cout), aux)
    return out, part


def pw_wgrad(d: Tensor, x: Tensor, N: int, Cd: int, Cx: int, P: int, *, pro_d: int = PRO_NONE, dk=(None, None, None),
             d2: Optional[Tensor] = None, pro_x: int = PRO_NONE, xk=(None, None, None), x2: Optional[Tensor] = None,
             per_frame: bool = False, rowsum: bool = False, partials: bool = False, d_amax: Optional[Tensor] = None,
             d2_amax: Optional[Tensor] = None, x_ub: Optional[Tensor] = None, dense: bool = False):
    """dW[co,ci] = sum_{n,p} fD(d)[n,co,p] * fX(x)[n,ci,p]  (-> [Cd,Cx], or [N,Cd,Cx] if per_frame);
    dense: as in pw_gemm.
    optionally also rowsum[co] = sum_{n,p} fD(d).  partials: the per-block partials themselves, (part [N*nbx, cop, cip], nbx, cop,
    cip), for a consumer that reduces them on the way (uncr_prenorm_bwd_finish)."""
    import ctypes
    cop, cip = ctypes.c_int(), ctypes.c_int()
    if hb.lib().cdll.uncr_wgrad_shape(Cd, Cx, ctypes.byref(cop), ctypes.byref(cip)) < 0:
        if Cd <= 256 and 128 < Cx <= 256 and not partials:
            # no kernel holds a [256][256] product: two calls over halves of the x channels (contiguous copies; a slow path for the
            # configurations wider than BASELINE's, e.g. the L-TAE input convolution behind a 256-wide encoder)
            half, outs, rs = (Cx + 1) // 2, [], None
            cutk = lambda t, o, n: None if t is None else t.view(N, Cx)[:, o:o + n].contiguous().view(-1)
            for o, n in ((0, half), (half, Cx - half)):
                xs = x.reshape(N, Cx, P)[:, o:o + n].contiguous()
                x2s = None if x2 is None else x2.reshape(N, Cx, P)[:, o:o + n].contiguous()
                dWp, r = pw_wgrad(d, xs, N, Cd, n, P, pro_d=pro_d, dk=dk, d2=d2, pro_x=pro_x, xk=tuple(cutk(t, o, n) for t in xk),
                                  x2=x2s, per_frame=per_frame, rowsum=rowsum and rs is None, dense=dense)
                outs.append(dWp)
                rs = r if rs is None else rs
            return torch.cat(outs, dim=-1).contiguous(), rs
        raise RuntimeError(f"weight-gradient shape ({Cd},{Cx}) not built")
    cop, cip = cop.value, cip.value
    Pv = P if dense else _pcount(P)
    if d.dtype != x.dtype:
        raise RuntimeError("pw_wgrad: both operands must have the same storage type")
    act = _dt(d)
    nbx = hb.query("uncr_wgrad_nbx", N, Cd, Cx, P, pro_d, pro_x, 1 if rowsum else 0, act)
    if nbx <= 0:
        raise RuntimeError(f"weight-gradient problem (N={N}, P={P}) not supported")
    dev = d.device
    part = _f32((N * nbx, cop, cip), dev)
    rs_part = _f32((N * nbx, cop), dev) if rowsum else None
    if not (_H2_BWD and _H2_WGRAD and d_amax is not None and d2_amax is not None and x_ub is not None and d_amax.shape[0] == N
            and d2_amax.shape[0] == N and x_ub.numel() == N * Cx):
        d_amax = d2_amax = x_ub = None          # the row-scaled fp16 split needs every bound; the exact split otherwise
    hb.call("uncr_pw_wgrad", d, d2, x, x2, dk[0], dk[1], dk[2], dk[3] if len(dk) > 3 else None, xk[0], xk[1], xk[2], part,
            rs_part, N, Cd, Cx, P, nbx, pro_d, pro_x, act, d_amax, d_amax.numel() // N if d_amax is not None else 0, d2_amax,
            d2_amax.numel() // N if d2_amax is not None else 0, x_ub, Pv, _stream())
    if Pv != P:      # padded planes: the kernel summed the whole 32-pixel chunks below the image's pixel count; the rest here
        hb.call("uncr_wgrad_boundary", d, d2, x, dk[0], dk[1], dk[2], dk[3] if len(dk) > 3 else None, xk[0], xk[1], xk[2], part, rs_part,
                N, Cd, Cx, P, Pv, nbx, pro_d, pro_x, _stream())
    if partials:
        return part, nbx, cop, cip
    n_out = N if per_frame else 1
    dW = _f32((n_out, Cd, Cx), dev)
    hb.call("uncr_wgrad_reduce", part, n_out, (N * nbx) // n_out, cop, cip, Cd, Cx, dW, _stream())
    rs = None
    if rowsum:
        rs = _f32((Cd,), dev)
        hb.call("uncr_wgrad_reduce", rs_part, 1, N * nbx, cop, 1, Cd, 1, rs, _stream())
    return (dW if per_frame else dW[0]), rs


# ------------------------------------------------------------------------------------------------
# MBConv (uncrtaints.py:100-146)
# ------------------------------------------------------------------------------------------------
MB_KEYS = ("n0w", "n0b", "w1", "n1w", "n1b", "wdw", "n2w", "n2b", "se1", "se2", "w2", "n3w", "n3b")


def mbconv_forward(x: Tensor, p: Dict[str, Tensor], spec: NormSpec, training: bool,
                   x_part: Optional[Part] = None, buffers: Optional[Dict[str, Tensor]] = None,
                   want_out_stats: bool = True, x_h3: Optional[Tensor] = None, pool: Optional[int] = None,
                   inference: bool = False):
    """x [N,C,H,W] -> y, saved-for-backward dict, partial stats of y (for the next PreNorm).
    inference: no gradient will be asked of this call (the saved dict may then be empty).
    `buffers` holds BatchNorm running_mean/var tensors keyed n{0..3}rm / n{0..3}rv (updated in place).
    pool: also return AdaptiveMaxPool2d((pool, pool))(y) and its argmax as saved["ypool"] (last encoder block: the
    L-TAE stage's pooling, uncrtaints.py:403-404, taken inside the residual kernel where the shape allows)."""
    N, C, H, W = _check4(x)
    P = H * W
    Ch = p["w1"].shape[0]
    R = p["se1"].shape[0]
    if Ch > 256:        # hidden width beyond the GEMM kernels' 256 channels: the hidden axis in groups (slow path, any expansion)
        if _geom_for(P) is not None:
            raise NotImplementedError("any-size planes with a hidden width beyond 256 channels are not built")
        return _mbconv_forward_wide(x, p, spec, training, x_part, buffers or {}, want_out_stats, x_h3, pool)
    need = spec.needs_stats(training)
    buffers = buffers or {}
    dt = _dt(x)

    def rm(i):
        return buffers.get(f"n{i}rm"), buffers.get(f"n{i}rv")

    if need and x_part is None:
        x_part = stats_sq(x, N * C, P)
    # range bookkeeping of the two wide forward GEMMs (fp32 storage): the (sum, sum^2) partials of their inputs bound the
    # inputs' magnitude; where a producer left none (eval-mode BatchNorm behind a foreign tensor) the GEMM takes the exact split
    h2ok = dt == F32 and _H2_FWD and Ch > 64 and C > 64
    n0 = norm_fwd(x_part if need else None, N, C, P, spec, training, p["n0w"], p["n0b"], *rm(0),
                  bound_part=x_part if h2ok else None, src=x)
    W1t = pack_wt(p["w1"].reshape(Ch, C), transpose=True)
    geom = _geom_for(P)      # padded planes of an any-size image (csrc/anysize.hip)
    if geom is not None and dt != F32:
        raise NotImplementedError("any-size planes are built for fp32 storage")
    h1, part1 = pw_gemm(x, W1t, N, C, Ch, P, pro=PRO_AFFINE, k=(n0.A, n0.B, None), epi=1 if need else 0, in_amax=n0.ub)
    # (hb: the bound on |h1| itself, for the backward's dx GEMM, which reads h1 through the norm-1 backward)
    n1 = norm_fwd(part1, N, Ch, P, spec, training, p["n1w"], p["n1b"], *rm(1),
                  bound_part=part1 if (h2ok and _H2_BWD and part1 is not None) else None, want_hb=True,
                  defer=_DW_VARIANT == 0 and hb.query("uncr_dw_fwd_bn_supported", H, W) == 1, src=h1)

    h2 = _act((N, Ch, H, W), x.device, dt)
    slots = hb.query("uncr_dw_slots_fwd", H) if geom is None else _dw_any_slots(geom, False)
    part2 = Part(_f32((N * Ch, slots, 2), x.device), slots) if (need or h2ok) else None
    if geom is not None:
        hb.call("uncr_dw_fwd_any", h1, n1.A, n1.B, p["wdw"].reshape(Ch, 9).contiguous(), h2, part2.buf if part2 is not None else None,
                N, Ch, geom.H, geom.W, geom.Pc, _stream())          # (writes the zero tail too)
    elif n1.fin is not None:      # train-mode BatchNorm 1 finalised by the depthwise kernel's waves themselves
        hb.call("uncr_dw_fwd_bn", h1, *n1.fin, n1.A, n1.B, n1.mean, n1.rstd, n1.ub, n1.hb, p["wdw"].reshape(Ch, 9).contiguous(),
                h2, part2.buf if part2 is not None else None, N, Ch, H, W, dt, _stream())
    else:
        hb.call("uncr_dw_fwd", h1, n1.A, n1.B, p["wdw"].reshape(Ch, 9).contiguous(), h2,
                part2.buf if part2 is not None else None, N, Ch, H, W, dt, _DW_VARIANT, _stream())
    n2 = norm_fwd(part2 if need else None, N, Ch, P, spec, training, p["n2w"], p["n2b"], *rm(2),
                  bound_part=part2 if h2ok else None, src=h2)

    ppool = se_pool(h2, n2.A, n2.B, N * Ch, P)
    pooled, hid_pre, s = _f32((N, Ch), x.device), _f32((N, R), x.device), _f32((N * Ch,), x.device)
    hb.call("uncr_se_mlp_fwd", ppool.buf, ppool.slots, N, Ch, R, _pcount(P), p["se1"].contiguous(), p["se2"].contiguous(),
            pooled, hid_pre, s, _stream())

    W2t = pack_wt(p["w2"].reshape(C, Ch), transpose=True)
    if (_EVAL_TAIL and inference and not need and pool is None and 64 < C <= 128
            and spec.code(training) == NORM_BATCH_EVAL):
        # inference (eval-mode BatchNorm, no autograd): norm 3 is a fixed affine map, so it and the skip ride on pw2's epilogue
        # (uncr_pw_gemm epi 10: the element-wise residual kernel's arithmetic and statistics on an h3 that is never stored)
        n3 = norm_fwd(None, N, C, P, spec, training, p["n3w"], p["n3b"], *rm(3))
        y, party = pw_gemm(h2, W2t, N, Ch, C, P, pro=PRO_AFFINE_GELU, k=(n2.A, n2.B, s), epi=10, aux=x,
                           ek=(n3.A, n3.B, None, None), in_amax=n2.ub)
        return y.view(N, C, H, W), dict(ypool=None, h3=None, dims=(N, C, Ch, R, H, W)), (party if want_out_stats else None)
    h3, part3 = pw_gemm(h2, W2t, N, Ch, C, P, pro=PRO_AFFINE_GELU, k=(n2.A, n2.B, s), epi=1 if need else 0, want_amax=True,
                        in_amax=n2.ub)
    n3 = norm_fwd(part3, N, C, P, spec, training, p["n3w"], p["n3b"], *rm(3), src=h3)

    y = _act((N, C, H, W), x.device, dt)
    ypool = None
    if pool is not None and hb.query("uncr_residual_pool_supported", H, W, pool, pool) == 1:
        down = _f32((N, C, pool, pool), x.device)
        idx = torch.empty((N, C, pool, pool), device=x.device, dtype=torch.int32)
        party = None
        if want_out_stats:
            slots = hb.query("uncr_residual_pool_slots", H)
            party = Part(_f32((N * C, slots, 2), x.device), slots)
        hb.call("uncr_residual_pool", x, h3, n3.A, n3.B, y, party.buf if party else None, down, idx, N * C, H, W, pool,
                pool, dt, _stream())
        ypool = (down, idx)
    else:
        _, party = ew(EW_RESIDUAL, x, b=h3, out=y, k=(n3.A, n3.B, None, None), want_part=want_out_stats, planes=N * C,
                      P=P)
        if pool is not None:
            ypool = maxpool_forward(y, pool, pool)
    saved = dict(ypool=ypool, x=x, h1=h1, h2=h2, h3=h3, n0=n0, n1=n1, n2=n2, n3=n3, pooled=pooled, hid_pre=hid_pre, s=s,
                 dims=(N, C, Ch, R, H, W), x_h3=x_h3, part1f=part1, h3_amax=part3.amax if part3 is not None else None, geom=geom)
    return y, saved, party


_FUSED_DX = True     # dev_options(fused_dx=False): pw1 backward as GEMM + element-wise pass (tests, A/B runs)

# ------------------------------------------------------------------------------------------------
# MBConv with a hidden width beyond 256 channels (block width > 128 at the reference's expansion 2, uncrtaints.py:100-105): the GEMM,
# weight-gradient and SE kernels are built for at most 256 channels per operand, so the HIDDEN axis is cut into groups of <= 128
# channels that live in separate tensors.  Everything between the two pointwise convolutions is per channel (norm 1, GELU, depthwise
# 3x3, norm 2, GELU, SE scale) and runs per group on the kernels of the fast path; pw1 is one GEMM per group, pw2 and the pw1 data
# gradient are sums over the groups (first group stores, the others accumulate: epilogue 4) followed by a statistics pass, the SE MLP
# sees the concatenated pooled vector.  A functional path for the wider configurations, not a tuned one: no fp16 operand split, no
# fused dx epilogue, one extra pass over h3 / da for their statistics.
# ------------------------------------------------------------------------------------------------
_WIDE_GROUP = 128


def _wide_groups(Ch: int, spec: NormSpec):
    """[(offset, channels)] of the hidden axis.  GroupNorm groups must not straddle a cut: the step is the largest multiple of the norm
    group's channel count that fits into _WIDE_GROUP."""
    step = _WIDE_GROUP
    if spec.kind == "group":
        per = Ch // spec.groups
        if Ch % spec.groups or per > _WIDE_GROUP:
            raise NotImplementedError(f"MBConv hidden width {Ch} in {spec.groups} norm groups: a norm group of {Ch / spec.groups:g} "
                                      f"channels does not fit the {_WIDE_GROUP}-channel groups of the wide path")
        step = (_WIDE_GROUP // per) * per
    return [(o, min(step, Ch - o)) for o in range(0, Ch, step)]


def _wide_spec(spec: NormSpec, Ch: int, Cg: int) -> NormSpec:
    """The hidden norms of one group of Cg channels."""
    return NormSpec("group", Cg // (Ch // spec.groups)) if spec.kind == "group" else spec


def _cut(t: Optional[Tensor], o: int, n: int) -> Optional[Tensor]:
    return None if t is None else t[o:o + n]


def _mbconv_forward_wide(x: Tensor, p: Dict[str, Tensor], spec: NormSpec, training: bool, x_part: Optional[Part],
                         buffers: Dict[str, Tensor], want_out_stats: bool, x_h3: Optional[Tensor], pool: Optional[int]):
    N, C, H, W = _check4(x)
    P = H * W
    Ch, R = p["w1"].shape[0], p["se1"].shape[0]
    dt, dev = _dt(x), x.device
    if dt != F32:
        raise NotImplementedError("bf16 activations: MBConv blocks with more than 256 hidden channels are not built")
    if C > 256:
        raise NotImplementedError(f"MBConv width {C}: the GEMM kernels take at most 256 channels per operand")
    need = spec.needs_stats(training)
    groups = _wide_groups(Ch, spec)

    def rm(i, o=None, n=None):
        a, b = buffers.get(f"n{i}rm"), buffers.get(f"n{i}rv")
        return (a, b) if o is None else (_cut(a, o, n), _cut(b, o, n))

    if need and x_part is None:
        x_part = stats_sq(x, N * C, P)
    n0 = norm_fwd(x_part if need else None, N, C, P, spec, training, p["n0w"], p["n0b"], *rm(0), src=x)
    w1, wdw, w2 = p["w1"].reshape(Ch, C), p["wdw"].reshape(Ch, 9), p["w2"].reshape(C, Ch)
    slots = hb.query("uncr_dw_slots_fwd", H)
    h1s, h2s, n1s, n2s, pools = [], [], [], [], []
    for o, n in groups:
        gs = _wide_spec(spec, Ch, n)
        h1, part1 = pw_gemm(x, pack_wt(w1[o:o + n], transpose=True), N, C, n, P, pro=PRO_AFFINE, k=(n0.A, n0.B, None),
                            epi=1 if need else 0)
        n1 = norm_fwd(part1, N, n, P, gs, training, _cut(p["n1w"], o, n), _cut(p["n1b"], o, n), *rm(1, o, n))
        h2 = _act((N, n, H, W), dev, dt)
        part2 = Part(_f32((N * n, slots, 2), dev), slots) if need else None
        hb.call("uncr_dw_fwd", h1, n1.A, n1.B, wdw[o:o + n].contiguous(), h2, part2.buf if part2 is not None else None, N, n, H, W,
                dt, _DW_VARIANT, _stream())
        n2 = norm_fwd(part2, N, n, P, gs, training, _cut(p["n2w"], o, n), _cut(p["n2b"], o, n), *rm(2, o, n))
        pp = se_pool(h2, n2.A, n2.B, N * n, P)
        h1s.append(h1); h2s.append(h2); n1s.append(n1); n2s.append(n2); pools.append(pp)
    ps = pools[0].slots
    ppool = torch.cat([q.buf.view(N, n, ps, 2) for q, (_, n) in zip(pools, groups)], dim=1).contiguous()     # [N][Ch][slots]
    pooled, hid_pre, s = _f32((N, Ch), dev), _f32((N, R), dev), _f32((N * Ch,), dev)
    hb.call("uncr_se_mlp_fwd", ppool, ps, N, Ch, R, P, p["se1"].contiguous(), p["se2"].contiguous(), pooled, hid_pre, s, _stream())
    sg = [s.view(N, Ch)[:, o:o + n].contiguous().view(-1) for o, n in groups]

    h3 = _act((N, C, H, W), dev, dt)
    for i, (o, n) in enumerate(groups):      # h3 = sum over the groups of W2[:, group] gelu(norm2(h2 group)) * s
        pw_gemm(h2s[i], pack_wt(w2[:, o:o + n].contiguous(), transpose=True), N, n, C, P, pro=PRO_AFFINE_GELU,
                k=(n2s[i].A, n2s[i].B, sg[i]), epi=0 if i == 0 else 4, out=h3.view(N, C, P))
    part3 = stats_sq(h3, N * C, P) if need else None
    n3 = norm_fwd(part3, N, C, P, spec, training, p["n3w"], p["n3b"], *rm(3), src=h3)

    y = _act((N, C, H, W), dev, dt)
    _, party = ew(EW_RESIDUAL, x, b=h3, out=y, k=(n3.A, n3.B, None, None), want_part=want_out_stats, planes=N * C, P=P)
    ypool = maxpool_forward(y, pool, pool) if pool is not None else None
    saved = dict(wide=True, ypool=ypool, x=x, h1=h1s, h2=h2s, h3=h3, n0=n0, n1=n1s, n2=n2s, n3=n3, pooled=pooled, hid_pre=hid_pre,
                 s=s, sg=sg, groups=groups, dims=(N, C, Ch, R, H, W), x_h3=x_h3)
    return y, saved, party


def _mbconv_backward_wide(dy: Tensor, sv: dict, p: Dict[str, Tensor], need_dx: bool, dy_part: Optional[Part]):
    N, C, Ch, R, H, W = sv["dims"]
    P = H * W
    dev = dy.device
    x, h3, n0, n3 = sv["x"], sv["h3"], sv["n0"], sv["n3"]
    dt = _dt(x)
    dy = cast(dy.contiguous(), dt)
    groups = sv["groups"]
    g: Dict[str, Tensor] = {}
    part3 = dy_part if dy_part is not None else stats_aux(dy, h3, N * C, P)
    b3 = norm_bwd(part3, N, C, P, n3, p["n3w"])
    g["n3w"], g["n3b"] = b3.dgamma, b3.dbeta
    k3 = b3.k
    w1, wdw, w2 = p["w1"].reshape(Ch, C), p["wdw"].reshape(Ch, 9), p["w2"].reshape(C, Ch).contiguous()

    # pw2: per-frame products dh3 (x) g2 per group, concatenated along the hidden axis for the SE backward
    Gs = [pw_wgrad(dy, sv["h2"][i], N, C, n, P, pro_d=PRO_NORMBWD, dk=k3, d2=h3, pro_x=PRO_AFFINE_GELU,
                   xk=(sv["n2"][i].A, sv["n2"][i].B, None), per_frame=True)[0] for i, (o, n) in enumerate(groups)]
    G = torch.cat(Gs, dim=2).contiguous()            # [N][C][Ch]
    ds_pre, dhid_pre, dpool = _f32((N, Ch), dev), _f32((N, R), dev), _f32((N * Ch,), dev)
    dW2, dse1, dse2 = _f32((C, Ch), dev), _f32((R, Ch), dev), _f32((Ch, R), dev)
    hb.call("uncr_se_mlp_bwd", G, w2, N, C, Ch, R, P, p["se1"].contiguous(), p["se2"].contiguous(), sv["s"], sv["pooled"],
            sv["hid_pre"], ds_pre, dhid_pre, dpool, dW2, dse1, dse2, _stream())
    g["w2"], g["se1"], g["se2"] = dW2.view_as(p["w2"]), dse1, dse2

    slots = hb.query("uncr_dw_slots_bwd", H)
    dW1 = _f32((Ch, C), dev)
    dwdw = _f32((Ch, 9), dev)
    n1w, n1b, n2w, n2b = (_f32((Ch,), dev) for _ in range(4))
    da = _act((N, C, H, W), dev, dt)
    for i, (o, n) in enumerate(groups):
        h1, h2, n1, n2 = sv["h1"][i], sv["h2"][i], sv["n1"][i], sv["n2"][i]
        dpg = dpool.view(N, Ch)[:, o:o + n].contiguous().view(-1)
        # dz = W2[:, group]^T dh3 with the SE / GELU backward and its statistics in the epilogue
        du2, part2 = pw_gemm(dy, pack_wt(w2[:, o:o + n].contiguous(), transpose=False), N, C, n, P, pro=PRO_NORMBWD, k=k3, x2=h3,
                             epi=3, aux=h2, ek=(n2.A, n2.B, sv["sg"][i], dpg))
        b2 = norm_bwd(part2, N, n, P, n2, _cut(p["n2w"], o, n))
        n2w[o:o + n], n2b[o:o + n] = b2.dgamma, b2.dbeta
        du1 = _act((N, n, H, W), dev, dt)
        part1 = Part(_f32((N * n, slots, 2), dev), slots)
        dw_part = _f32((N * n, slots, 9), dev)
        hb.call("uncr_dw_bwd", du2, h2, h1, b2.c1, b2.c2, b2.c3, b2.mu, n1.A, n1.B, wdw[o:o + n].contiguous(), du1, part1.buf, dw_part,
                n1.mean, n1.groups if n1.kind == NORM_GROUP else 0, N, n, H, W, dt, _DW_VARIANT, None, _stream())
        dwg = _f32((n, 9), dev)
        hb.call("uncr_dw_wgrad_reduce", dw_part, N, n, slots, dwg, _stream())
        dwdw[o:o + n] = dwg
        b1 = norm_bwd(part1, N, n, P, n1, _cut(p["n1w"], o, n), centered=True)
        n1w[o:o + n], n1b[o:o + n] = b1.dgamma, b1.dbeta
        dW1g, _ = pw_wgrad(du1, x, N, n, C, P, pro_d=PRO_NORMBWD, dk=b1.k, d2=h1, pro_x=PRO_AFFINE, xk=(n0.A, n0.B, None))
        dW1[o:o + n] = dW1g
        # da = sum over the groups of W1[group]^T dh1
        pw_gemm(du1, pack_wt(w1[o:o + n], transpose=False), N, n, C, P, pro=PRO_NORMBWD, k=b1.k, x2=h1, epi=0 if i == 0 else 4,
                out=da.view(N, C, P))
    g["w1"], g["wdw"] = dW1.view_as(p["w1"]), dwdw.view_as(p["wdw"])
    for key, val in (("n1w", n1w), ("n1b", n1b), ("n2w", n2w), ("n2b", n2b)):
        g[key] = val if p[key] is not None else None
    part0 = stats_aux(da, x, N * C, P)
    b0 = norm_bwd(part0, N, C, P, n0, p["n0w"])
    g["n0w"], g["n0b"] = b0.dgamma, b0.dbeta
    dx, dx_part = None, None
    if need_dx:
        dx = _act((N, C, H, W), dev, dt)
        x_h3 = sv.get("x_h3")
        _, dx_part = ew(EW_PASSE, dy, b=da, c=x, aux=x_h3, out=dx, k=b0.k, want_part=x_h3 is not None, planes=N * C, P=P)
    return dx, g, dx_part


_CONST_PLANES: Dict[tuple, Tuple[Tensor, Tensor]] = {}


def _const_planes(dev, n: int) -> Tuple[Tensor, Tensor]:
    key = (str(dev), n)
    if key in _CONST_PLANES:
        return _CONST_PLANES[key]
    pair = (torch.ones(n, device=dev, dtype=torch.float32), torch.zeros(n, device=dev, dtype=torch.float32))
    if not torch.cuda.is_current_stream_capturing():     # memory from a graph's private pool must not outlive the capture
        _CONST_PLANES[key] = pair
    return pair


def mbconv_backward(dy: Tensor, sv: dict, p: Dict[str, Tensor], need_dx: bool = True,
                    dy_part: Optional[Part] = None):
    """-> dx, {param key: grad}, partials (sum dx, sum dx*h3_prev) for the producing block (or None).
    `dy_part` = (sum dy, sum dy*h3) partials if the kernel that produced dy already emitted them."""
    if sv.get("wide"):
        return _mbconv_backward_wide(dy, sv, p, need_dx, dy_part)
    N, C, Ch, R, H, W = sv["dims"]
    P = H * W
    geom = sv.get("geom")
    if geom is not None and _geom() != geom:        # (a backward entered outside the model's scope: re-enter the forward's geometry)
        with geom_scope(geom):
            return mbconv_backward(dy, sv, p, need_dx, dy_part)
    dev = dy.device
    x, h1, h2, h3 = sv["x"], sv["h1"], sv["h2"], sv["h3"]
    n0, n1, n2, n3 = sv["n0"], sv["n1"], sv["n2"], sv["n3"]
    dt = _dt(x)
    dy = cast(dy.contiguous(), dt)
    g: Dict[str, Tensor] = {}

    # norm 3 backward coefficients: needs (sum dy, sum dy*h3)
    part3 = dy_part if dy_part is not None else stats_aux(dy, h3, N * C, P)
    dy_amax = dy_part.amax if dy_part is not None else None       # per-frame bounds on |dy| left by dy's producer (or None)
    if dy_amax is not None and dy_amax.shape[0] != N:
        dy_amax = None
    b3 = norm_bwd(part3, N, C, P, n3, p["n3w"])
    g["n3w"], g["n3b"] = b3.dgamma, b3.dbeta
    k3 = b3.k

    # pw2: per-frame products G[n] = dh3 (x) g2  -> dW2 and the SE gradient
    # (fp32 storage, magnitude bookkeeping at hand: two row-scaled fp16 parts, like the dz GEMM below)
    G, _ = pw_wgrad(dy, h2, N, C, Ch, P, pro_d=PRO_NORMBWD, dk=k3, d2=h3, pro_x=PRO_AFFINE_GELU,
                    xk=(n2.A, n2.B, None), per_frame=True, d_amax=dy_amax, d2_amax=sv.get("h3_amax"), x_ub=n2.ub)
    ds_pre, dhid_pre, dpool = _f32((N, Ch), dev), _f32((N, R), dev), _f32((N * Ch,), dev)
    dW2, dse1, dse2 = _f32((C, Ch), dev), _f32((R, Ch), dev), _f32((Ch, R), dev)
    w2 = p["w2"].reshape(C, Ch).contiguous()
    # (per-frame phase only: dpool is what the dz GEMM waits for; dW2 and the SE weight gradients are reduced further down, in one
    # launch with the depthwise weight gradient -- uncr_mbconv_param_grads)
    hb.call("uncr_se_mlp_bwd", G, w2, N, C, Ch, R, _pcount(P), p["se1"].contiguous(), p["se2"].contiguous(), sv["s"],
            sv["pooled"], sv["hid_pre"], ds_pre, dhid_pre, dpool, None, None, None, _stream())
    g["w2"], g["se1"], g["se2"] = dW2.view_as(p["w2"]), dse1, dse2

    # dz = W2^T dh3 ; du2 = gelu'(u2) * (s*dz + dpool) (in place) with stats (sum du2, sum du2*h2)
    W2k = pack_wt(w2, transpose=False)                     # [k=co 128][out=c 256]
    # ... with the SE / GELU backward (du2 = gelu'(u2) * (s*dz + dpool)) and its statistics fused into the
    # GEMM epilogue: dz itself is never written
    # with the magnitude bookkeeping of dy's producer and of the forward pw2 GEMM at hand: two scaled fp16 parts (three products)
    du2, part2 = pw_gemm(dy, W2k, N, C, Ch, P, pro=PRO_NORMBWD, k=k3, x2=h3, epi=3, aux=h2,
                         ek=(n2.A, n2.B, sv["s"], dpool), in_amax=dy_amax, in2_amax=sv.get("h3_amax"))
    b2 = norm_bwd(part2, N, Ch, P, n2, p["n2w"])
    g["n2w"], g["n2b"] = b2.dgamma, b2.dbeta

    # depthwise backward
    du1 = _act((N, Ch, H, W), dev, dt)
    slots = hb.query("uncr_dw_slots_bwd", H) if geom is None else _dw_any_slots(geom, True)
    part1 = Part(_f32((N * Ch, slots, 2), dev), slots)
    dw_part = _f32((N * Ch, slots, 9), dev)
    wdw = p["wdw"].reshape(Ch, 9).contiguous()
    # statistics for the norm-1 backward in centred form (sum du1*(h1 - mean1)): h1 is the raw pw1 output, whose
    # channel means can be many standard deviations from zero
    # with the bound on |h1| at hand (NormFwd.hb), the row kernel also leaves max |du1| per slot: the two operands of the dx GEMM are
    # then bounded and it multiplies in two scaled fp16 parts
    du1_amax = None
    if geom is not None:
        hb.call("uncr_dw_bwd_any", du2, h2, h1, b2.c1, b2.c2, b2.c3, b2.mu, n1.A, n1.B, wdw, du1, part1.buf, dw_part, n1.mean,
                n1.groups if n1.kind == NORM_GROUP else 0, N, Ch, geom.H, geom.W, geom.Pc, _stream())
    else:
        if _H2_BWD and _H2_DX and n1.hb is not None and hb.query("uncr_dw_bwd_emits_amax", H, W, dt, _DW_VARIANT) == 1:
            du1_amax = _f32((N, Ch * slots), dev)
        hb.call("uncr_dw_bwd", du2, h2, h1, b2.c1, b2.c2, b2.c3, b2.mu, n1.A, n1.B, wdw, du1, part1.buf, dw_part, n1.mean,
                n1.groups if n1.kind == NORM_GROUP else 0, N, Ch, H, W, dt, _DW_VARIANT, du1_amax, _stream())
    dwdw = _f32((Ch, 9), dev)
    with side_chain(dw_part):          # feeds parameter gradients only: next to the pw1 weight-gradient GEMM
        hb.call("uncr_mbconv_param_grads", G, N, C, Ch, R, sv["s"], sv["pooled"], sv["hid_pre"], ds_pre, dhid_pre, dW2, dse1, dse2,
                dw_part, Ch, slots, dwdw, _stream())
    g["wdw"] = dwdw.view_as(p["wdw"])
    b1 = norm_bwd(part1, N, Ch, P, n1, p["n1w"], centered=True)
    g["n1w"], g["n1b"] = b1.dgamma, b1.dbeta
    k1 = b1.k

    W1k = pack_wt(p["w1"].reshape(Ch, C), transpose=False)  # [k=co 256][out=ci 128]
    if need_dx and _FUSED_DX and C % 32 == 0 and Ch % 8 == 0 and hb.query("uncr_pw_gemm_dx_supported", Ch, C) == 1:
        # pw1 backward without a pass over da: the weight-gradient GEMM runs first, on the UN-normalised x and per frame
        # (R[n,k,c] = sum_p du1n*x); da = W1^T du1n is linear, so (sum da, sum da*x) and dW1 follow from R and from sums
        # that exist already (uncr_prenorm_bwd_finish).  The PreNorm backward coefficients are therefore known before the
        # data GEMM, whose epilogue writes dx = dy + c1*da + c2*x + c3 (and the producing block's norm-3 statistics).
        one, zero = _const_planes(dev, N * C)
        # InstanceNorm (one group per plane): the products are taken on x - mean, so that a plane far from zero -- or a CONSTANT one, a
        # zero-padded date -- leaves no difference of two separately rounded sums behind (uncr_prenorm_bwd_finish, xmu)
        xmu = negmu = None
        if _CENTRED_PW1 and n0.kind == NORM_GROUP and n0.groups == C and n0.mean.numel() == N * C:
            xmu, negmu = n0.mean, plane_means(n0, N, C, -1.0)
        wpart, nbx, cop, cip = pw_wgrad(du1, x, N, Ch, C, P, pro_d=PRO_NORMBWD, dk=k1, d2=h1, pro_x=PRO_AFFINE,
                                        xk=(one, zero if negmu is None else negmu, None), partials=True)
        part0 = Part(_f32((N * C, Ch // 2, 2), dev), Ch // 2)
        dW1 = _f32((Ch, C), dev)
        pf = sv.get("part1f")
        hb.call("uncr_prenorm_bwd_finish", wpart, nbx, cop, cip, p["w1"].reshape(Ch, C).contiguous(), part1.buf, part1.slots,
                pf.buf if pf is not None else None, pf.slots if pf is not None else 0, k1[0], k1[1], k1[2],
                k1[3] if pf is not None else None, n0.A, n0.B, part0.buf, dW1, N, Ch, C, _pcount(P), xmu, _stream())
        g["w1"] = dW1.view_as(p["w1"])
        b0 = norm_bwd(part0, N, C, P, n0, p["n0w"], centered=xmu is not None)
        g["n0w"], g["n0b"] = b0.dgamma, b0.dbeta
        dx = _act((N, C, H, W), dev, dt)
        x_h3 = sv.get("x_h3")
        relu = sv.get("x_relu")       # x = relu(A*c0 + B) of in_conv: its ReLU backward and norm statistics ride along
        ra = rb = rmu = None
        if relu is not None:
            x_h3, ra, rb = relu[:3]   # (c0, A, B, mean per plane); or (None, A, None): the mask is [x > 0] itself, c0 was never stored (inconv_forward)
            rmu = relu[3] if (len(relu) > 3 and rb is not None) else None
        dx_part = None
        if x_h3 is not None or relu is not None:
            slots = hb.query("uncr_pw_stat_slots", N, C, P)
            dx_part = Part(_f32((N * C, slots, 2), dev), slots, masked=relu is not None, centered=rmu is not None)
            if relu is None and dt == F32 and _H2_BWD:      # dx is the next backward's dy: leave its per-block maxima for that dz GEMM
                dx_part.amax = _f32((N, slots), dev)
        hb.call("uncr_pw_gemm_dx", du1, h1, W1k, dx, k1[0], k1[1], k1[2], k1[3], dy, x, x_h3, b0.c1, b0.c2, b0.c3, b0.mu, ra, rb, rmu,
                dx_part.buf if dx_part is not None else None, N, Ch, C, P, dt,
                dx_part.amax if dx_part is not None else None, du1_amax, du1_amax.shape[1] if du1_amax is not None else 0,
                n1.hb if du1_amax is not None else None, Ch if du1_amax is not None else 0, _pcount(P), _stream())
        if geom is not None:      # (statistics partner: the producing block's h3, or x itself where the mask is [x > 0])
            fix_tail(dx, dx_part, 1, N * C, hb.query("uncr_pw_tile_px", C), x_h3 if x_h3 is not None else x, rmu)
        join_side()
        return dx, g, dx_part

    # pw1: weight gradient and data gradient (block widths the fused backward does not take, e.g. 64)
    dW1, _ = pw_wgrad(du1, x, N, Ch, C, P, pro_d=PRO_NORMBWD, dk=k1, d2=h1, pro_x=PRO_AFFINE, xk=(n0.A, n0.B, None))
    g["w1"] = dW1.view_as(p["w1"])
    da, part0 = pw_gemm(du1, W1k, N, Ch, C, P, pro=PRO_NORMBWD, k=k1, x2=h1, epi=2, aux=x)
    b0 = norm_bwd(part0, N, C, P, n0, p["n0w"])
    g["n0w"], g["n0b"] = b0.dgamma, b0.dbeta

    dx, dx_part = None, None
    if need_dx:
        dx = _act((N, C, H, W), dev, dt)
        x_h3 = sv.get("x_h3")     # h3 of the block that produced x: emit its norm-3 backward statistics here
        _, dx_part = ew(EW_PASSE, dy, b=da, c=x, aux=x_h3, out=dx, k=b0.k,
                        want_part=x_h3 is not None, planes=N * C, P=P)
    join_side()
    return dx, g, dx_part


# ------------------------------------------------------------------------------------------------
# ResidualConvBlock (uncrtaints.py:24-69): x + CL3(CL2(CL1(x))), CL = dense conv3x3 (reflect, bias) -> norm -> ReLU.
# The dense 3x3 convolution runs as nine accumulating pointwise GEMMs on the padded grid (csrc/conv3.hip).
# ------------------------------------------------------------------------------------------------
EW_RESIDUAL_RELU = 10
RES_KEYS = tuple(f"{n}{i}" for i in (1, 2, 3) for n in ("w", "b", "g", "be"))     # conv weight, bias, norm weight, bias


class _Padded:
    """[planes][S_p] padded planes inside a flat buffer with `margin` zero floats of slack on both sides, so that a
    tap's shifted view (flat[margin + off : ...]) stays inside the allocation."""

    def __init__(self, N: int, C: int, H: int, W: int, dev, geom=None):
        self.N, self.C, self.H, self.W = N, C, H, W
        self.geom = geom          # any-size image: the un-padded side lives on dense planes of stride geom.Pc (csrc/anysize.hip)
        self.Sp = hb.query("uncr_conv3_plane_stride", H, W)
        self.margin = hb.query("uncr_conv3_margin", W)
        self.n = N * C * self.Sp
        self.flat = torch.zeros(self.n + 2 * self.margin, device=dev, dtype=torch.float32)

    def view(self, off: int = 0) -> Tensor:
        return self.flat[self.margin + off: self.margin + off + self.n]


def _taps(W: int):
    return [(ky, kx, (ky - 1) * (W + 2) + (kx - 1)) for ky in range(3) for kx in range(3)]


def conv3x3_forward(xp: _Padded, w: Tensor, b: Tensor, want_stats: bool):
    """xp: reflect-padded input planes.  -> raw conv output c [N,Co,H,W], statistics partials of c."""
    N, Ci, H, W = xp.N, xp.C, xp.H, xp.W
    Co = w.shape[0]
    wt = w.permute(2, 3, 0, 1).contiguous()            # [3,3,Co,Ci]: one contiguous matrix per tap (layout copy)
    outp = _Padded(N, Co, H, W, w.device)
    for i, (ky, kx, off) in enumerate(_taps(W)):
        pw_gemm(xp.view(off), pack_wt(wt[ky, kx], transpose=True), N, Ci, Co, xp.Sp, bias=b.contiguous() if i == 0 else None,
                epi=0 if i == 0 else 4, out=outp.view().view(N, Co, xp.Sp), dense=True)
    g = xp.geom
    c = _f32((N, Co, 1, g.Pc) if g is not None else (N, Co, H, W), w.device)
    part = None
    if want_stats:
        slots = hb.query("uncr_ew_slots", g.Pc if g is not None else H * W)
        part = Part(_f32((N * Co, slots, 2), w.device), slots)
    if g is not None:
        hb.call("uncr_unpad2d_strided", outp.view(), c, part.buf if part else None, N * Co, H, W, g.Pc, _stream())
    else:
        hb.call("uncr_unpad2d", outp.view(), c, part.buf if part else None, N * Co, H, W, _stream())
    return c, part


def conv3x3_backward(du: Tensor, c: Tensor, kk, xp: _Padded, w: Tensor, need_dx: bool):
    """du: gradient after the ReLU mask; kk: norm-backward coefficients (dc = C1*du + C2*c + C3).
    -> (gradient wrt the un-padded conv input or None, dW [Co,Ci,3,3], db [Co])"""
    N, Ci, H, W = xp.N, xp.C, xp.H, xp.W
    Co = w.shape[0]
    dev = w.device
    g = xp.geom
    dcp = _Padded(N, Co, H, W, dev, g)
    if g is not None:
        hb.call("uncr_pad2d_strided", du.contiguous(), c, dcp.view(), kk[0], kk[1], kk[2], kk[3] if len(kk) > 3 else None, PRO_NORMBWD,
                1, N * Co, H, W, g.Pc, _stream())
    else:
        hb.call("uncr_pad2d", du.contiguous(), c, dcp.view(), kk[0], kk[1], kk[2], kk[3] if len(kk) > 3 else None, PRO_NORMBWD, 1,
                N * Co, H, W, _stream())
    dWt = []
    db = None
    for i, (ky, kx, off) in enumerate(_taps(W)):
        dW_t, rs = pw_wgrad(dcp.view().view(N, Co, xp.Sp), xp.view(off).view(N, Ci, xp.Sp), N, Co, Ci, xp.Sp, rowsum=(i == 0), dense=True)
        dWt.append(dW_t)
        if i == 0:
            db = rs
    dW = torch.stack(dWt, dim=0).view(3, 3, Co, Ci).permute(2, 3, 0, 1).contiguous()
    dx = None
    if need_dx:
        wt = w.permute(2, 3, 0, 1).contiguous()
        dxp = _Padded(N, Ci, H, W, dev)
        for i, (ky, kx, off) in enumerate(_taps(W)):
            pw_gemm(dcp.view(-off), pack_wt(wt[ky, kx], transpose=False), N, Co, Ci, xp.Sp, epi=0 if i == 0 else 4,
                    out=dxp.view().view(N, Ci, xp.Sp), dense=True)
        if g is not None:
            dx = _f32((N, Ci, 1, g.Pc), dev)
            hb.call("uncr_unpad2d_reflect_adjoint_strided", dxp.view(), dx, N * Ci, H, W, g.Pc, _stream())
        else:
            dx = _f32((N, Ci, H, W), dev)
            hb.call("uncr_unpad2d_reflect_adjoint", dxp.view(), dx, N * Ci, H, W, _stream())
    return dx, dW, db


def residual_forward(x: Tensor, p: Dict[str, Tensor], spec: NormSpec, training: bool, buffers=None):
    """ResidualConvBlock.forward.  p: RES_KEYS; buffers: {rm1, rv1, ...} BatchNorm running statistics."""
    N, C, H, W = _check4(x)
    P = H * W                 # pixels per stored plane (any-size: the padded count Pc; the finalisations divide by the true count)
    geom = _geom_for(P)
    if geom is not None:
        if _dt(x) != F32:
            raise NotImplementedError("any-size planes are built for fp32 storage")
        H, W = geom.H, geom.W          # the 3x3 taps run on the true image; only the un-padded side has the dense planes' stride
    dev = x.device
    sv = dict(x=x, xp=[], c=[], nf=[], dims=(N, C, H, W), geom=geom)
    src, pro, k = x, PRO_NONE, (None, None)
    for i in (1, 2, 3):
        xp = _Padded(N, C, H, W, dev, geom)
        if geom is not None:
            hb.call("uncr_pad2d_strided", src, None, xp.view(), k[0], k[1], None, None, pro, 0, N * C, H, W, geom.Pc, _stream())
        else:
            hb.call("uncr_pad2d", src, None, xp.view(), k[0], k[1], None, None, pro, 0, N * C, H, W, _stream())
        c, part = conv3x3_forward(xp, p[f"w{i}"], p[f"b{i}"], spec.needs_stats(training))
        rm = buffers.get(f"rm{i}") if buffers else None
        rv = buffers.get(f"rv{i}") if buffers else None
        nf = norm_fwd(part, N, C, P, spec, training, p[f"g{i}"], p[f"be{i}"], rm, rv)
        sv["xp"].append(xp); sv["c"].append(c); sv["nf"].append(nf)
        src, pro, k = c, PRO_AFFINE_RELU, (nf.A, nf.B)
    y = torch.empty_like(x)
    ew(EW_RESIDUAL_RELU, x, b=src, out=y, k=(k[0], k[1], None, None), planes=N * C, P=P)
    return y, sv


def residual_backward(dy: Tensor, sv: dict, p: Dict[str, Tensor], need_dx: bool = True):
    geom = sv.get("geom")
    if geom is not None and _geom() != geom:      # re-enter the forward's any-size geometry
        with geom_scope(geom):
            return residual_backward(dy, sv, p, need_dx)
    N, C, H, W = sv["dims"]
    P = H * W if geom is None else geom.Pc
    dev = dy.device
    g: Dict[str, Tensor] = {}
    da = dy.contiguous()
    for i in (3, 2, 1):
        c, nf, xp = sv["c"][i - 1], sv["nf"][i - 1], sv["xp"][i - 1]
        du = torch.empty_like(c)       # (a zero tail of `da` gives a zero tail here: the mask multiplies)
        _, part = ew(EW_RELU_BWD, da, b=c, out=du, k=(nf.A, nf.B, plane_means(nf, N, C), None), want_part=True, planes=N * C, P=P)
        nb = norm_bwd(part, N, C, P, nf, p[f"g{i}"], centered=True)
        g[f"g{i}"], g[f"be{i}"] = nb.dgamma, nb.dbeta
        da, g[f"w{i}"], g[f"b{i}"] = conv3x3_backward(du, c, nb.k, xp, p[f"w{i}"],
                                                      need_dx or i > 1)
    dx = None
    if need_dx:
        dx = torch.empty_like(da)
        hb.call("uncr_add", dy.contiguous(), da, dx, dx.numel(), _stream())
    return dx, g


# ------------------------------------------------------------------------------------------------
# in_conv: Conv2d(15->128,k1,bias) + GroupNorm(4) + ReLU   (utae.py:453-520, uncrtaints.py:310-314)
# ------------------------------------------------------------------------------------------------

def _inconv_moments_ok(x: Tensor, N: int, Cin: int, Cout: int, spec: NormSpec, gw, P: int, training: bool = True) -> bool:
    """The moment path is taken in the FORWARD and leaves no c0 behind, so everything its backward needs is checked here: the limits
    of uncr_inconv_moments (P % 4) and of uncr_inconv_bwd_finish, which holds [N][Cout / G] partial pairs of a group in 60 KB of LDS
    (csrc/inconv.hip: (2 N Cg + 4 N + 288) doubles) -- at width 256 (Cg = 64) that is N = B*T <= 56; beyond it the generic path.
    GroupNorm: either storage type.  InstanceNorm (one channel per group) and train-mode BatchNorm (statistics over all frames; not
    with synchronised statistics): fp32 storage (round 6)."""
    if not (_INCONV_MOMENTS and Cin + 1 <= 16 and 64 < Cout <= 256 and N <= 64 and P % 4 == 0):
        return False
    if spec.kind == "group":
        return gw is not None and Cout % spec.groups == 0 and (2 * N * (Cout // spec.groups) + 4 * N + 288) * 8 <= 60 * 1024
    if _dt(x) != F32:
        return False
    if spec.kind == "instance":
        return (2 * N + 4 * N + 288) * 8 <= 60 * 1024
    if spec.kind == "batch":
        return training and gw is not None and _SYNC_BN is None
    return False


def inconv_forward(x: Tensor, w: Tensor, b: Tensor, gw: Tensor, gb: Tensor, spec: NormSpec, training: bool,
                   buffers: Optional[Dict[str, Tensor]] = None):
    N, Cin, H, W = _check4(x)
    P = H * W
    Cout = w.shape[0]
    buffers = buffers or {}
    need = spec.needs_stats(training)
    Wt = pack_wt(w.reshape(Cout, Cin), transpose=True)
    if _inconv_moments_ok(x, N, Cin, Cout, spec, gw, P, training):
        # c0 = W x + b is never written: its norm statistics are a quadratic form in the frames' 16 x 16 augmented moment matrices (fp64),
        # and the GEMM's epilogue stores relu(A*c0 + B) straight from the accumulators (csrc/inconv.hip)
        dev = x.device
        Pv = _pcount(P)                # (padded planes of an any-size image: the valid pixels count, the stride is P)
        nblk = hb.query("uncr_inconv_moment_blocks", Pv)
        mpart = torch.empty((N, nblk, 256), device=dev, dtype=torch.float64)
        hb.call("uncr_inconv_moments", x, N, Cin, Pv, mpart, _dt(x), P, _stream())
        A, B = _f32((N * Cout,), dev), _f32((N * Cout,), dev)
        mom = torch.empty((N, 256), device=dev, dtype=torch.float64)
        w2d, bc = w.reshape(Cout, Cin).contiguous(), b.contiguous()
        if spec.kind == "batch":       # train mode: statistics per channel over all frames
            mean, rstd = _f32((Cout,), dev), _f32((Cout,), dev)
            momtot = torch.empty((256,), device=dev, dtype=torch.float64)
            hb.call("uncr_inconv_bn_from_moments", mpart, nblk, N, Cin, Cout, w2d, bc, gw, gb, buffers.get("rm"), buffers.get("rv"), 0.1,
                    1e-5, A, B, mean, rstd, mom, momtot, _stream())
            nf = NormFwd(A, B, mean, rstd, NORM_BATCH_TRAIN, 1)
            mom = momtot
        else:
            G = Cout if spec.kind == "instance" else spec.groups
            if gw is None:
                gw_, gb_ = _const_planes(dev, Cout)
            else:
                gw_, gb_ = gw, gb
            mean, rstd = _f32((N * G,), dev), _f32((N * G,), dev)
            hb.call("uncr_inconv_norm_from_moments", mpart, nblk, N, Cin, Cout, G, w2d, bc, gw_, gb_, 1e-5, A, B, mean, rstd, mom, _stream())
            nf = NormFwd(A, B, mean, rstd, NORM_GROUP, G)
        a0, parta = pw_gemm(x, Wt, N, Cin, Cout, P, bias=bc, epi=9, ek=(A, B, None, None))
        a0 = a0.view(N, Cout, H, W)
        return a0, dict(x=x, c0=None, a0=a0, nf=nf, mom=mom, b=bc, dims=(N, Cin, Cout, H, W), geom=_geom_for(P)), parta
    geom = _geom_for(P)
    c0, part = pw_gemm(x, Wt, N, Cin, Cout, P, bias=b.contiguous(), epi=1 if need else 0)
    nf = norm_fwd(part, N, Cout, P, spec, training, gw, gb, buffers.get("rm"), buffers.get("rv"), src=c0)
    a0 = _act((N, Cout, H, W), x.device, _dt(x))
    _, parta = ew(EW_AFFINE_RELU, c0, out=a0, k=(nf.A, nf.B, None, None), want_part=True, planes=N * Cout, P=P)
    # per-plane means of the norm: pivots of the backward's second statistic, sum du0*(c0 - mean) -- c0 = W x + b of non-negative
    # inputs sits several standard deviations from zero, and the raw sum du0*c0 - mean*sum du0 cancels in fp32 slots
    mu = plane_means(nf, N, Cout) if (need and training and _CENTRED_INCONV) else None
    return a0, dict(x=x, c0=c0, nf=nf, mu=mu, dims=(N, Cin, Cout, H, W), geom=geom), parta


def inconv_backward(da0: Tensor, sv: dict, w: Tensor, gw: Tensor, need_dx: bool, masked_part: Optional[Part] = None):
    """masked_part: da0 is already du0 = da0 * [relu mask] and these are its (sum du0, sum du0*c0) partials (the consumer's
    backward GEMM applied the mask in its epilogue, uncr_pw_gemm_dx).  Without c0 (inconv_forward's moment path) the second
    component is not used: sum du0*c0 follows from the per-frame products R = sum_p du0 x^T."""
    N, Cin, Cout, H, W = sv["dims"]
    P = H * W
    geom = sv.get("geom")
    if geom is not None and _geom() != geom:
        with geom_scope(geom):
            return inconv_backward(da0, sv, w, gw, need_dx, masked_part)
    nf, c0, x = sv["nf"], sv["c0"], sv["x"]
    da0 = cast(da0.contiguous(), _dt(x))
    if c0 is None:
        dev = da0.device
        if masked_part is not None:
            du0, part = da0, masked_part
        else:           # the consumer did not apply the mask: [a0 > 0] is the ReLU's mask
            one, zero = _const_planes(dev, N * Cout)
            du0 = _act((N, Cout, H, W), dev, _dt(x))
            _, part = ew(EW_RELU_BWD, da0, b=sv["a0"], out=du0, k=(one, zero, None, None), want_part=True, planes=N * Cout, P=P)
        R, _ = pw_wgrad(du0, x, N, Cout, Cin, P, per_frame=True)            # [N, Cout, Cin] = sum_p du0 x^T
        dW, db, dg, dbeta = _f32((Cout, Cin), dev), _f32((Cout,), dev), _f32((Cout,), dev), _f32((Cout,), dev)
        w2d = w.reshape(Cout, Cin).contiguous()
        if nf.kind == NORM_BATCH_TRAIN:
            hb.call("uncr_inconv_bwd_finish_bn", R.contiguous(), part.buf, part.slots, sv["mom"], w2d, sv["b"], gw, nf.mean, nf.rstd, N,
                    Cin, Cout, dW, db, dg, dbeta, _stream())
        else:
            hb.call("uncr_inconv_bwd_finish", R.contiguous(), part.buf, part.slots, sv["mom"], w2d, sv["b"],
                    gw if gw is not None else _const_planes(dev, Cout)[0], nf.mean, nf.rstd, N, Cin, Cout, nf.groups, dW, db, dg, dbeta,
                    _stream())
        dx = None
        if need_dx:
            # the gradient w.r.t. the model input needs c0 per pixel: recomputed (a training run's input carries no gradient)
            Wt = pack_wt(w2d, transpose=True)
            c0r, _ = pw_gemm(x, Wt, N, Cin, Cout, P, bias=sv["b"])
            nb = norm_bwd(stats_aux(du0, c0r, N * Cout, P), N, Cout, P, nf, gw)
            Wk = pack_wt(w2d, transpose=False)
            dx, _ = pw_gemm(du0, Wk, N, Cout, Cin, P, pro=PRO_NORMBWD, k=nb.k, x2=c0r, out_dt=F32)
            dx = dx.view(N, Cin, H, W)
        return dx, dW.view_as(w), db, dg, dbeta
    if masked_part is not None:
        du0, part = da0, masked_part
    else:
        du0 = _act((N, Cout, H, W), da0.device, _dt(x))
        _, part = ew(EW_RELU_BWD, da0, b=c0, out=du0, k=(nf.A, nf.B, sv.get("mu"), None), want_part=True, planes=N * Cout, P=P)
        part.centered = sv.get("mu") is not None
    nb = norm_bwd(part, N, Cout, P, nf, gw, centered=part.centered)
    kk = nb.k
    dW, db = pw_wgrad(du0, x, N, Cout, Cin, P, pro_d=PRO_NORMBWD, dk=kk, d2=c0, rowsum=True)
    dx = None
    if need_dx:
        Wk = pack_wt(w.reshape(Cout, Cin), transpose=False)   # [k=128][out=15]
        dx, _ = pw_gemm(du0, Wk, N, Cout, Cin, P, pro=PRO_NORMBWD, k=kk, x2=c0, out_dt=F32)     # the model input's gradient: fp32
        dx = dx.view(N, Cin, H, W)
    return dx, dW.view_as(w), db, nb.dgamma, nb.dbeta


# ------------------------------------------------------------------------------------------------
# L-TAE stage: max-pool + LTAE2dtiny attention + temporal aggregation (uncrtaints.py:402-412)
# ------------------------------------------------------------------------------------------------

def pad_mask_of(x: Tensor, pad_value: float) -> Tensor:
    """[B,T,...] -> int32 [B,T]: 1 where the whole frame equals pad_value (uncrtaints.py:392-394)."""
    B, T = x.shape[:2]
    mask = torch.empty((B, T), device=x.device, dtype=torch.int32)
    hb.call("uncr_pad_mask", x, B * T, x[0, 0].numel(), float(pad_value), mask, _stream())
    return mask


def maxpool_forward(e: Tensor, OH: int, OW: int):
    """e [..., H, W] (planes = product of leading dims) -> pooled [..., OH, OW], argmax int32."""
    H, W = e.shape[-2:]
    planes = e.numel() // (H * W)
    lead = tuple(e.shape[:-2])
    down = _f32(lead + (OH, OW), e.device)
    idx = torch.empty(lead + (OH, OW), device=e.device, dtype=torch.int32)
    g = _geom_for(H * W)
    if g is not None:       # padded planes of an any-size image: the window arithmetic of the true H x W
        hb.call("uncr_maxpool_fwd_strided", e, down, idx, planes, g.H, g.W, g.Pc, OH, OW, _stream())
        return down, idx
    hb.call("uncr_maxpool_fwd", e, down, idx, planes, H, W, OH, OW, _dt(e), _stream())
    return down, idx


def maxpool_backward_into(ddown: Tensor, idx: Tensor, de: Tensor, H: int, W: int, OH: int, OW: int):
    """de[plane][argmax] += ddown (in place on `de`)."""
    planes = ddown.numel() // (OH * OW)
    g = _geom_for(H * W)
    if g is not None:
        hb.call("uncr_maxpool_bwd_strided", ddown.contiguous(), idx, de, planes, g.H, g.W, g.Pc, OH, OW, _stream())
        return
    hb.call("uncr_maxpool_bwd", ddown.contiguous(), idx, de, planes, H, W, OH, OW, _dt(de), _stream())


def ltae_attention_forward(down: Tensor, dates: Optional[Tensor], pad: Optional[Tensor], p: Dict[str, Tensor],
                           denom: Optional[Tensor], n_head: int, d_k: int):
    """LTAE2dtiny (ltae.py:197-239): down [B,T,C,h,w] -> att [n_head,B,T,h,w].
    p: in_norm_w/b [C], inconv_w [D,C,1], inconv_b [D], fc_w [nh*dk,D], fc_b, Q [nh,dk]."""
    B, T, C, ah, aw = down.shape
    S = ah * aw
    if S % 256:
        raise RuntimeError("L-TAE resolution must be a multiple of 256 pixels")
    D = p["inconv_w"].shape[0]
    HK = n_head * d_k
    dev = down.device
    NF = B * T
    use_pe = denom is not None
    xn, mean, rstd = _f32((NF, C, S), dev), _f32((B, n_head, S), dev), _f32((B, n_head, S), dev)
    hb.call("uncr_ltae_gn_fwd", down, p["in_norm_w"], p["in_norm_b"], 1e-5, xn, mean, rstd, B, T, C, n_head, S,
            _stream())
    bias1 = _f32((NF, D), dev)
    hb.call("uncr_ltae_posbias", dates.reshape(-1).contiguous().float() if use_pe else None,
            denom if use_pe else None, denom.numel() if use_pe else 0, p["inconv_b"], bias1, NF, D,
            1 if use_pe else 0, _stream())
    Wit = pack_wt(p["inconv_w"].reshape(D, C), transpose=True)
    y1, _ = pw_gemm(xn, Wit, NF, C, D, S, bias=bias1, bias_per_frame=True)
    Wkt = pack_wt(p["fc_w"], transpose=True)
    k, _ = pw_gemm(y1, Wkt, NF, D, HK, S, bias=p["fc_b"].contiguous())
    att = _f32((n_head, B, T, ah, aw), dev)
    hb.call("uncr_ltae_softmax_fwd", k, p["Q"].contiguous(), pad, att, B, T, n_head, d_k, S, _stream())
    saved = dict(down=down, xn=xn, mean=mean, rstd=rstd, y1=y1, k=k, att=att, pad=pad, dims=(B, T, C, S, D, HK))
    return att, saved


_FUSED_LTAE = True     # dev_options(fused_ltae=False): the unfused L-TAE kernels (tests, A/B runs)


def ltae_fused_ok(T: int, C: int, n_head: int, S: int) -> bool:
    return _FUSED_LTAE and hb.query("uncr_ltae_fused_supported", T, C, n_head, S) == 1 and n_head in (4, 8, 16, 32)


def ltae_attention_forward_fused(down: Tensor, dates: Optional[Tensor], pad: Optional[Tensor], p: Dict[str, Tensor],
                                 denom: Optional[Tensor], n_head: int, d_k: int):
    """LTAE2dtiny (ltae.py:197-239) as one fused kernel: the score is a linear functional of the group-normalised input
    (csrc/ltae_fused.hip); A', B' are composed from the parameters first.  down [B,T,C,h,w] -> att [n_head,B,T,h,w]."""
    B, T, C, ah, aw = down.shape
    S = ah * aw
    D = p["inconv_w"].shape[0]
    HK = n_head * d_k
    dev = down.device
    NF = B * T
    use_pe = denom is not None
    bias1 = _f32((NF, D), dev)
    Ap, Bp, M, U = _f32((n_head, C), dev), _f32((n_head, NF), dev), _f32((HK, C), dev), _f32((HK, NF), dev)
    Wi, Wk = p["inconv_w"].reshape(D, C).contiguous(), p["fc_w"].contiguous()
    hb.call("uncr_ltae_compose", p["Q"].contiguous(), Wk, p["fc_b"].contiguous(), Wi, p["inconv_b"],
            dates.reshape(-1).contiguous().float() if use_pe else None, denom if use_pe else None,
            denom.numel() if use_pe else 0, 1 if use_pe else 0, p["in_norm_w"], p["in_norm_b"], n_head, d_k, D, C, NF, bias1, Ap, Bp,
            M, U, _stream())
    att = _f32((n_head, B, T, ah, aw), dev)
    mean, rstd = _f32((B, n_head, S), dev), _f32((B, n_head, S), dev)
    hb.call("uncr_ltae_fused_fwd", down, Ap, Bp, pad, 1e-5, att, mean, rstd, B, T, C, n_head, S, _stream())
    saved = dict(fused=True, down=down, att=att, mean=mean, rstd=rstd, Ap=Ap, M=M, U=U, bias1=bias1, pad=pad,
                 dims=(B, T, C, S, D, HK))
    return att, saved


def ltae_attention_backward_fused(datt: Tensor, sv: dict, p: Dict[str, Tensor], n_head: int, d_k: int):
    """-> d(down) [B*T,C,S], {param grads}"""
    B, T, C, S, D, HK = sv["dims"]
    dev = datt.device
    NF = B * T
    nblk = S * n_head // 256          # blocks per sample (256 / n_head pixels each)
    ddown = _f32((NF, C, S), dev)
    partA, partB = _f32((B * nblk, n_head, C), dev), _f32((B * nblk, n_head, T), dev)
    hb.call("uncr_ltae_fused_bwd", datt.contiguous(), sv["att"], sv["down"], sv["Ap"], sv["pad"], sv["mean"], sv["rstd"], ddown,
            partA, partB, B, T, C, n_head, S, _stream())
    # everything from here on only feeds parameter gradients: a side chain next to the pooled-gradient scatter (ltae_stage_backward
    # joins it).  Outputs are allocated first, on the current stream.
    dQ = _f32((n_head, d_k), dev)
    scratch = _f32((n_head * C + n_head + NF * n_head + n_head * 2 * C,), dev)
    dWk, dbk, dWi, dbi = _f32((HK, D), dev), _f32((HK,), dev), _f32((D, C), dev), _f32((D,), dev)
    gb = _f32((2 * C,), dev)
    Wi, Wk, Q = p["inconv_w"].reshape(D, C).contiguous(), p["fc_w"].contiguous(), p["Q"].contiguous()
    with side_chain(partA, partB, Wi, Wk, Q):
        hb.call("uncr_ltae_compose_bwd", Q, Wk, Wi, sv["bias1"], p["in_norm_w"], p["in_norm_b"], sv["M"], sv["U"], partA, partB,
                nblk, n_head, d_k, D, C, NF, T, scratch, dQ, dWk, dbk, dWi, dbi, gb, _stream())
    g = dict(Q=dQ, fc_w=dWk, fc_b=dbk, inconv_w=dWi.view_as(p["inconv_w"]), inconv_b=dbi, in_norm_w=gb[:C], in_norm_b=gb[C:])
    return ddown, g


def ltae_attention_backward(datt: Tensor, sv: dict, p: Dict[str, Tensor], n_head: int, d_k: int,
                            dy1_extra: Optional[Tensor] = None):
    """-> d(down) [B*T,C,S], {param grads}.  dy1_extra: gradient reaching y1 through the values (use_v)."""
    if sv.get("fused"):
        assert dy1_extra is None
        return ltae_attention_backward_fused(datt, sv, p, n_head, d_k)
    B, T, C, S, D, HK = sv["dims"]
    NF = B * T
    dev = datt.device
    g: Dict[str, Tensor] = {}
    nchunk = (S + 255) // 256
    dk = _f32((NF, HK, S), dev)
    dq_part = _f32((B * nchunk, HK), dev)
    hb.call("uncr_ltae_softmax_bwd", datt.contiguous(), sv["att"], sv["k"], p["Q"].contiguous(), sv["pad"], dk,
            dq_part, B, T, n_head, d_k, S, _stream())
    dQ = _f32((HK,), dev)
    hb.call("uncr_colsum", dq_part, B * nchunk, HK, dQ, _stream())
    g["Q"] = dQ.view(n_head, d_k)
    dWk, dbk = pw_wgrad(dk, sv["y1"], NF, HK, D, S, rowsum=True)
    g["fc_w"], g["fc_b"] = dWk, dbk
    Wkk = pack_wt(p["fc_w"], transpose=False)                     # [k=64][out=256]
    dy1, _ = pw_gemm(dk, Wkk, NF, HK, D, S)
    if dy1_extra is not None:
        hb.call("uncr_add", dy1, dy1_extra.contiguous(), dy1, dy1.numel(), _stream())
    dWi, dbi = pw_wgrad(dy1, sv["xn"], NF, D, C, S, rowsum=True)
    g["inconv_w"], g["inconv_b"] = dWi.view_as(p["inconv_w"]), dbi
    Wik = pack_wt(p["inconv_w"].reshape(D, C), transpose=False)   # [k=256][out=128]
    dxn, _ = pw_gemm(dy1, Wik, NF, D, C, S)
    ddown = _f32((NF, C, S), dev)
    gb_part = _f32((B * nchunk, C, 2), dev)
    hb.call("uncr_ltae_gn_bwd", dxn, sv["down"], p["in_norm_w"], sv["mean"], sv["rstd"], ddown, gb_part, B, T, C,
            n_head, S, _stream())
    gb = _f32((C * 2,), dev)
    hb.call("uncr_colsum", gb_part, B * nchunk, C * 2, gb, _stream())
    gb = gb.view(C, 2)
    g["in_norm_w"], g["in_norm_b"] = gb[:, 0].contiguous(), gb[:, 1].contiguous()
    return ddown, g


def aggregate_forward(e: Tensor, att: Tensor, pad: Optional[Tensor], training: bool, p_drop: float, seed,
                      dmask: Optional[Tensor] = None, want_stats: bool = True, shared_mask: bool = False):
    # `seed` is an int, or (int, device int64 tensor): the tensor is a step counter read by the kernel, so that a
    # captured HIP graph draws a fresh dropout mask on every replay
    """Compact_Temporal_Aggregator 'att_group' (uncrtaints.py:156-221): e [B,T,C,H,W], att [nh,B,T,ah,aw]."""
    B, T, C, H, W = e.shape
    n_head, _, _, ah, aw = att.shape
    dev = e.device
    geom = _geom_for(H * W)
    odd_heads = None
    if geom is None and C % n_head == 0 and C // n_head not in (2, 4, 6, 8, 16, 32) and not (H <= aw and (H, W) != (ah, aw)):
        # channels per head outside the streaming kernels' list (e.g. 96 channels in 8 heads): the scalar kernels below take any count;
        # a dense plane is a padded plane without a tail
        if _dt(e) != F32:
            raise NotImplementedError(f"{C // n_head} channels per head (other than 2, 4, 6, 8, 16, 32) are built for fp32 storage")
        geom = odd_heads = Geom(H, W, H * W)
    if geom is not None:
        # padded planes of an any-size image (csrc/anysize.hip): the scalar kernels on the true H x W (bilinear up-sampling of any ratio)
        if _dt(e) != F32:
            raise NotImplementedError("any-size planes are built for fp32 storage")
        if odd_heads is None and geom.H <= aw and (geom.H, geom.W) != (ah, aw):
            # a feature map not larger than the attention map (inputs below 32 x 32: uncrtaints.py:403-404 pools UP to 32 x 32 and the
            # aggregator takes its AvgPool2d(w // H) branch, uncrtaints.py:197-204): a few hundred pixels per plane -- through the dense
            # kernels of that branch on extracted planes, the result embedded again
            with geom_scope(None):
                gd, svd, _ = aggregate_forward(extract_tail(e, geom), att, pad, training, p_drop, seed, dmask, False, shared_mask)
            g = embed_tail(gd, geom)
            gpart = stats_sq(g, B * C, geom.Pc) if want_stats else None
            return g, dict(small=svd, geom=geom, dims=(B, T, C, H, W, n_head, ah, aw)), gpart
        if geom.H < ah or geom.W < aw:
            raise NotImplementedError(f"feature map {geom.H}x{geom.W} against a {ah}x{aw} attention map")
        g = _f32((B, C, H, W), dev)
        gpart = None
        if want_stats:
            slots = hb.query("uncr_agg_any_slots", geom.Pc, C, n_head)
            gpart = Part(_f32((B * C, slots, 2), dev), slots)
        use_mask = dmask if training else None
        pd = float(p_drop) if (training and dmask is None) else 0.0
        seed_val, seed_dev = seed if isinstance(seed, tuple) else (seed, None)
        hb.call("uncr_aggregate_any_fwd", e, att, pad, use_mask, seed_val, seed_dev, pd, 1 if shared_mask else 0, g,
                gpart.buf if gpart else None, B, T, C, n_head, geom.H, geom.W, geom.Pc, ah, aw, _stream())
        sv = dict(e=e, att=att, pad=pad, dmask=use_mask, pd=pd, seed=seed_val, seed_dev=seed_dev, shared=1 if shared_mask else 0,
                  dims=(B, T, C, H, W, n_head, ah, aw))
        if odd_heads is not None:
            sv["any"] = odd_heads              # (no geometry scope to re-enter, no tail to zero)
        else:
            fix_tail(g, None, 2, B * C)
            sv["geom"] = geom
        return g, sv, gpart
    if H <= aw and (H, W) != (ah, aw):
        # the reference's AvgPool branch (uncrtaints.py:197-204): the attention is pooled down to the feature map, no dropout.
        # (At equal size the pooling is the identity and the streaming kernel below serves it.)
        k = aw // H
        if k <= 0 or ah // k != H or aw // k != W:
            raise ValueError(f"AvgPool2d(kernel_size={k}) of a {ah}x{aw} attention map does not give the feature map's {H}x{W}")
        if _dt(e) != F32:
            raise NotImplementedError("the AvgPool branch of the aggregator is built for fp32 storage")
        g = _f32((B, C, H, W), dev)
        hb.call("uncr_aggregate_pool_fwd", e, att, pad, g, B, T, C, n_head, H, W, ah, aw, k, _stream())
        return g, dict(e=e, att=att, pad=pad, pool_k=k, dims=(B, T, C, H, W, n_head, ah, aw)), None
    if (H * W) % 1024 or W % 4:
        raise RuntimeError(f"unsupported spatial size {H}x{W}")
    if H < ah or W < aw:
        raise NotImplementedError(f"feature map {H}x{W} against a {ah}x{aw} attention map")
    g = _act((B, C, H, W), dev, _dt(e))
    gpart = None
    if want_stats:
        slots = hb.query("uncr_agg_slots", H * W)
        gpart = Part(_f32((B * C, slots, 2), dev), slots)
    use_mask = dmask if training else None
    pd = float(p_drop) if (training and dmask is None) else 0.0
    seed_val, seed_dev = seed if isinstance(seed, tuple) else (seed, None)
    hb.call("uncr_aggregate_fwd", e, att, pad, use_mask, seed_val, seed_dev, pd, 1 if shared_mask else 0, g,
            gpart.buf if gpart else None, B, T, C, n_head, H, W, ah, aw, _dt(e), _stream())
    saved = dict(e=e, att=att, pad=pad, dmask=use_mask, pd=pd, seed=seed_val, seed_dev=seed_dev,
                 shared=1 if shared_mask else 0, dims=(B, T, C, H, W, n_head, ah, aw))
    return g, saved, gpart


def aggregate_backward(dg: Tensor, sv: dict):
    """-> de [B,T,C,H,W] (freshly written), datt [nh,B,T,ah,aw]"""
    B, T, C, H, W, n_head, ah, aw = sv["dims"]
    dev = dg.device
    geom = sv.get("geom") or sv.get("any")
    if "small" in sv:         # the AvgPool branch of a padded-plane feature map: dense kernels on extracted planes (aggregate_forward)
        with geom_scope(None):
            ded, datt = aggregate_backward(extract_tail(dg.contiguous().float(), geom), sv["small"])
        return embed_tail(ded, geom), datt
    if geom is not None:
        de = _f32((B, T, C, H, W), dev)
        datt_up = _f32((n_head * B * T, geom.Pc), dev)          # (scratch: the gradient of the up-sampled attention, plane stride Pc)
        datt = _f32((n_head, B, T, ah, aw), dev)
        hb.call("uncr_aggregate_any_bwd", dg.contiguous().float(), sv["e"], sv["att"], sv["pad"], sv["dmask"], sv["seed"], sv["seed_dev"],
                sv["pd"], sv["shared"], de, datt_up, datt, B, T, C, n_head, geom.H, geom.W, geom.Pc, ah, aw, _stream())
        if "geom" in sv:
            with geom_scope(geom):
                fix_tail(de.view(B * T, C, H, W), None, 2, B * T * C)
        return de, datt
    if "pool_k" in sv:
        de, datt = _f32((B, T, C, H, W), dev), _f32((n_head, B, T, ah, aw), dev)
        hb.call("uncr_aggregate_pool_bwd", dg.contiguous().float(), sv["e"], sv["att"], sv["pad"], de, datt, B, T, C, n_head, H, W,
                ah, aw, sv["pool_k"], _stream())
        return de, datt
    dt = _dt(sv["e"])
    de = _act((B, T, C, H, W), dev, dt)
    datt_up = _f32((n_head, B, T, H * W), dev)
    datt = _f32((n_head, B, T, ah, aw), dev)
    hb.call("uncr_aggregate_bwd", cast(dg.contiguous(), dt), sv["e"], sv["att"], sv["pad"], sv["dmask"], sv["seed"],
            sv["seed_dev"], sv["pd"], sv["shared"], de, datt_up, datt, B, T, C, n_head, H, W, ah, aw, dt, _stream())
    return de, datt


def aggregate_backward_datt(dg: Tensor, sv: dict) -> Tensor:
    """Pass 1 of the two-pass backward (csrc/aggregate.hip): -> datt [nh,B,T,ah,aw]; nothing is written at full resolution."""
    B, T, C, H, W, n_head, ah, aw = sv["dims"]
    dev = dg.device
    dt = _dt(sv["e"])
    datt_up = _f32((n_head, B, T, H * W), dev)
    datt = _f32((n_head, B, T, ah, aw), dev)
    hb.call("uncr_aggregate_bwd_datt", cast(dg.contiguous(), dt), sv["e"], sv["att"], sv["pad"], sv["dmask"], sv["seed"], sv["seed_dev"],
            sv["pd"], sv["shared"], datt_up, datt, B, T, C, n_head, H, W, ah, aw, dt, _stream())
    return datt


def aggregate_backward_de(dg: Tensor, sv: dict, ddown: Optional[Tensor], idx: Optional[Tensor], att_down: int,
                          e_h3: Optional[Tensor]):
    """Pass 2: de [B,T,C,H,W] = a * dg + the pooled gradient scattered to the arg-max, written once, with the (sum de, sum de*h3)
    partials (and per-block maxima) of the block that produced e.  -> (de, Part or None)"""
    B, T, C, H, W, n_head, ah, aw = sv["dims"]
    dev = dg.device
    dt = _dt(sv["e"])
    de = _act((B, T, C, H, W), dev, dt)
    part = None
    if e_h3 is not None:
        slots = hb.query("uncr_ew_slots", H * W)
        part = Part(_f32((B * T * C, slots, 2), dev), slots)
        if dt == F32 and _H2_BWD:
            part.amax = _f32((B * T, C * slots), dev)     # per-block max |de|, one row per frame
    hb.call("uncr_aggregate_bwd_de", cast(dg.contiguous(), dt), sv["att"], sv["pad"], sv["dmask"], sv["seed"], sv["seed_dev"], sv["pd"],
            sv["shared"], de, ddown.contiguous() if ddown is not None else None, idx if ddown is not None else None, e_h3,
            part.buf if part is not None else None, part.amax if part is not None else None, B, T, C, n_head, H, W, ah, aw,
            att_down, att_down, dt, _stream())
    return de, part


# the aggregation backward in two passes (attention gradient | de + pooled-gradient scatter + the encoder block's statistics): de is written
# once and never re-read; False: one pass + uncr_pool_scatter_stats (A/B runs, bisecting)
_AGG_TWO_PASS = True


def head_mean_attention(att: Tensor) -> Tensor:
    """'att_mean' (uncrtaints.py:180,212): average the attention over heads, then give every head that average."""
    NH = att.shape[0]
    n = att[0].numel()
    m = _f32(att.shape[1:], att.device)
    scratch = _f32(att.shape[1:], att.device)
    hb.call("uncr_ensemble_combine", att, None, NH, n, 2, m, scratch, _stream())      # mean over the leading axis
    out = torch.empty_like(att)
    hb.call("uncr_bcast_scale", m, NH, n, 1.0, out, _stream())
    return out


def head_mean_attention_backward(datt_e: Tensor) -> Tensor:
    NH = datt_e.shape[0]
    n = datt_e[0].numel()
    s = _f32((n,), datt_e.device)
    hb.call("uncr_colsum", datt_e.contiguous(), NH, n, s, _stream())
    out = torch.empty_like(datt_e)
    hb.call("uncr_bcast_scale", s, NH, n, 1.0 / NH, out, _stream())
    return out


def mean_mode_weights(pad: Optional[Tensor], n_head: int, B: int, T: int, ah: int, aw: int, dev) -> Tensor:
    out = _f32((n_head, B, T, ah, aw), dev)
    hb.call("uncr_mean_weights", pad, n_head, B, T, ah * aw, out, _stream())
    return out


def _value_layers(p: Dict[str, Tensor]):
    """[(W, b, bn_w, bn_b), ...]: the value MLP's layers out of its parameter dict (keys of layer i > 0 carry the index)"""
    out, i = [], 0
    while ("mlp_w" + ("" if i == 0 else str(i))) in p:
        sfx = "" if i == 0 else str(i)
        out.append((p["mlp_w" + sfx], p["mlp_b" + sfx], p["bn_w" + sfx], p["bn_b" + sfx], sfx))
        i += 1
    return out


def ltae_values_forward(sv_att: dict, pad: Optional[Tensor], p: Dict[str, Tensor], n_head: int, training: bool,
                        bn_buffers, p_drop: float, seed):
    """LTAE2d values (ltae.py:122-133): attention-weighted sum over T of the projected features (per head), (Linear +
    BatchNorm1d + ReLU) per MLP layer, dropout, GroupNorm.  -> v [B,C,S] (C = mlp[-1]), saved.
    p: mlp_w [C,D], mlp_b, bn_w, bn_b (further layers: mlp_w1, ...), on_w, on_b; bn_buffers: (running_mean, running_var) of the one
    layer, or a list of such pairs.  The attention-weighted sum IS the temporal aggregation at the attention's own resolution, so it
    runs on the aggregate kernels."""
    B, T, Cin, S, D, HK = sv_att["dims"]
    att, y1 = sv_att["att"], sv_att["y1"]
    ah, aw = att.shape[-2:]
    dev = y1.device
    vh, sv_agg, _ = aggregate_forward(y1.view(B, T, D, ah, aw), att, pad, False, 0.0, 0, None, want_stats=False)
    spec = NormSpec("batch", 1)
    if bn_buffers is None or (isinstance(bn_buffers, tuple) and len(bn_buffers) == 2 and not isinstance(bn_buffers[0], tuple)):
        bn_buffers = [bn_buffers]
    layers, src, Cprev = [], vh.view(B, D, S), D
    for li, (w, b, bw, bb, _) in enumerate(_value_layers(p)):
        C = w.shape[0]
        m1, part = pw_gemm(src, pack_wt(w, transpose=True), B, Cprev, C, S, bias=b.contiguous(), epi=1)
        rm, rv = bn_buffers[li] if (li < len(bn_buffers) and bn_buffers[li] is not None) else (None, None)
        nf = norm_fwd(part, B, C, S, spec, training, bw, bb, rm, rv, src=m1)
        r = _f32((B, C, S), dev)
        ew(EW_AFFINE_RELU, m1, out=r, k=(nf.A, nf.B, None, None), want_part=False, planes=B * C, P=S)
        layers.append(dict(x=src, m1=m1, nf=nf, r=r, dims=(Cprev, C)))
        src, Cprev = r, C
    C, r = Cprev, src
    pd = float(p_drop) if training else 0.0
    seed_val, seed_dev = seed if isinstance(seed, tuple) else (seed, None)
    rd = r
    if pd > 0.0:
        rd = _f32((B, C, S), dev)
        hb.call("uncr_dropout", r, rd, r.numel(), seed_val, seed_dev, pd, _stream())
    v, mean, rstd = _f32((B, C, S), dev), _f32((B, n_head, S), dev), _f32((B, n_head, S), dev)
    hb.call("uncr_ltae_gn_fwd", rd, p["on_w"], p["on_b"], 1e-5, v, mean, rstd, B, 1, C, n_head, S, _stream())
    last = layers[-1]
    saved = dict(agg=sv_agg, vh=vh, layers=layers, m1=last["m1"], nf=last["nf"], r=r, rd=rd, gn=(mean, rstd), pd=pd, seed=seed_val,
                 seed_dev=seed_dev, dims=(B, T, C, S, D))
    return v, saved


def ltae_values_backward(dv: Tensor, sv: dict, p: Dict[str, Tensor], n_head: int):
    """-> (dy1 contribution [B*T,D,S], d attention [nh,B,T,ah,aw], {param grads})"""
    B, T, C, S, D = sv["dims"]
    dev = dv.device
    g: Dict[str, Tensor] = {}
    nchunk = (S + 255) // 256
    mean, rstd = sv["gn"]
    drd = _f32((B, C, S), dev)
    gb_part = _f32((B * nchunk, C, 2), dev)
    hb.call("uncr_ltae_gn_bwd", dv.contiguous(), sv["rd"], p["on_w"], mean, rstd, drd, gb_part, B, 1, C, n_head, S,
            _stream())
    gb = _f32((C * 2,), dev)
    hb.call("uncr_colsum", gb_part, B * nchunk, C * 2, gb, _stream())
    gb = gb.view(C, 2)
    g["on_w"], g["on_b"] = gb[:, 0].contiguous(), gb[:, 1].contiguous()
    dr = drd
    if sv["pd"] > 0.0:
        dr = _f32((B, C, S), dev)
        hb.call("uncr_dropout", drd, dr, drd.numel(), sv["seed"], sv["seed_dev"], sv["pd"], _stream())
    vl = _value_layers(p)
    for li in range(len(vl) - 1, -1, -1):            # the MLP's layers back to front: dr is the gradient behind layer li's ReLU
        w, _, bw, _, sfx = vl[li]
        lay = sv["layers"][li]
        Cx, Cl = lay["dims"]
        nf, m1 = lay["nf"], lay["m1"]
        du = _f32((B, Cl, S), dev)
        _, part = ew(EW_RELU_BWD, dr, b=m1, out=du, k=(nf.A, nf.B, plane_means(nf, B, Cl), None), want_part=True, planes=B * Cl, P=S)
        nb = norm_bwd(part, B, Cl, S, nf, bw, centered=True)
        g["bn_w" + sfx], g["bn_b" + sfx] = nb.dgamma, nb.dbeta
        kk = nb.k
        dWm, dbm = pw_wgrad(du, lay["x"].reshape(B, Cx, S), B, Cl, Cx, S, pro_d=PRO_NORMBWD, dk=kk, d2=m1, rowsum=True)
        g["mlp_w" + sfx], g["mlp_b" + sfx] = dWm, dbm
        dr, _ = pw_gemm(du, pack_wt(w, transpose=False), B, Cl, Cx, S, pro=PRO_NORMBWD, k=kk, x2=m1)      # W as [k = Cl][out = Cx]
    ah, aw = sv["agg"]["dims"][6:8]
    dy1, datt = aggregate_backward(dr.view(B, D, ah, aw), sv["agg"])
    return dy1.view(B * T, D, S), datt, g


def include_v_forward(gagg: Tensor, v: Tensor, w: Tensor, b: Tensor, want_stats: bool):
    """uncrtaints.py:414-417: include_v(cat(g, up(v))) = Wa*g + up(Wv*v + b)  (the 1x1 convolution commutes with the
    bilinear up-sampling, whose weights sum to one).  gagg [B,C,H,W], v [B,Cv,ah,aw], w [C,C+Cv,1,1]."""
    B, C, H, W = gagg.shape
    Cv, ah, aw = v.shape[1:]
    P, S = H * W, ah * aw
    geom = _geom_for(P)        # padded planes of an any-size image: P = Pc, zero tails
    w2 = w.reshape(C, C + Cv)
    wa, wv = w2[:, :C].contiguous(), w2[:, C:].contiguous()
    z, _ = pw_gemm(v.reshape(B, Cv, S), pack_wt(wv, transpose=True), B, Cv, C, S, bias=b.contiguous())
    t, _ = pw_gemm(gagg.view(B, C, P), pack_wt(wa, transpose=True), B, C, C, P)          # no bias: a zero tail stays zero
    out = _f32((B, C, H, W), gagg.device)
    part = None
    if want_stats:
        # (uncr_add_upsampled_any keeps the scalar kernels' slot count: one channel per "head" selects it)
        slots = hb.query("uncr_agg_any_slots", geom.Pc, 1, 1) if geom is not None else hb.query("uncr_agg_slots", P)
        part = Part(_f32((B * C, slots, 2), gagg.device), slots)
    if geom is not None:
        hb.call("uncr_add_upsampled_any", t, z, out, part.buf if part else None, B * C, geom.H, geom.W, geom.Pc, ah, aw, _stream())
    else:
        hb.call("uncr_add_upsampled", t, z, out, part.buf if part else None, B * C, H, W, ah, aw, _stream())
    return out, dict(g=gagg, v=v, wa=wa, wv=wv, dims=(B, C, Cv, H, W, ah, aw), geom=geom), part


def include_v_backward(dout: Tensor, sv: dict):
    """-> (d gagg [B,C,H,W], d v [B,Cv,ah,aw], dW [C,C+Cv,1,1], db [C])"""
    B, C, Cv, H, W, ah, aw = sv["dims"]
    P, S = H * W, ah * aw
    dout = dout.contiguous()
    dz = _f32((B, C, S), dout.device)
    if sv.get("geom") is not None:
        gm = sv["geom"]
        hb.call("uncr_bilinear_adjoint_any", dout, dz, B * C, gm.H, gm.W, gm.Pc, ah, aw, _stream())
    else:
        hb.call("uncr_bilinear_adjoint", dout, dz, B * C, H, W, ah, aw, _stream())
    dWa, db = pw_wgrad(dout.view(B, C, P), sv["g"].view(B, C, P), B, C, C, P, rowsum=True)     # db = sum dout (= sum dz)
    dWv, _ = pw_wgrad(dz, sv["v"].reshape(B, Cv, S), B, C, Cv, S)
    dg, _ = pw_gemm(dout.view(B, C, P), pack_wt(sv["wa"], transpose=False), B, C, C, P)
    dv, _ = pw_gemm(dz, pack_wt(sv["wv"], transpose=False), B, C, Cv, S)
    dW = torch.cat((dWa, dWv), dim=1).view(C, C + Cv, 1, 1)
    return dg.view(B, C, H, W), dv.view(B, Cv, ah, aw), dW, db


def ltae_stage_forward(e: Tensor, dates: Optional[Tensor], pad: Optional[Tensor], p: Dict[str, Tensor],
                       denom: Optional[Tensor], n_head: int, d_k: int, att_down: int, training: bool, p_drop: float,
                       seed: int, dmask: Optional[Tensor] = None, want_stats: bool = True, mode: str = "att_group",
                       values: Optional[dict] = None, pooled=None):
    """Fused stage used by UNCRTAINTS.forward: e [B,T,C,H,W] -> g [B,C,H,W] (+ stats partials of g).
    pooled: (down, idx) of AdaptiveMaxPool2d((att_down, att_down))(e) if the producer of e already took it.
    mode: 'att_group' | 'att_mean' | 'mean' (uncrtaints.py:156-221).
    values (use_v): dict(p=value-branch params, include_w, include_b, bn_buffers, p_drop, seed): the LTAE2d values are
    up-sampled and merged through include_v (uncrtaints.py:414-417)."""
    B, T = e.shape[:2]
    if _LTAE_REPLAY == "replay":
        g_, sv_, gp_, att_ = _LTAE_STORE["fwd"]
        return g_.detach(), sv_, gp_, att_         # (a fresh alias: the caller hands it to autograd as a new output)
    if _LTAE_REPLAY == "record":
        with dev_options(ltae_replay=None):
            _LTAE_STORE["fwd"] = ltae_stage_forward(e, dates, pad, p, denom, n_head, d_k, att_down, training, p_drop, seed, dmask,
                                                    want_stats, mode, values, pooled)
        return _LTAE_STORE["fwd"]
    if pooled is not None:
        down, idx = (v.view(B, T, e.shape[2], att_down, att_down) for v in pooled)
    else:
        down, idx = maxpool_forward(e, att_down, att_down)
    _, _, Cc, ah_, aw_ = down.shape
    if values is None and ltae_fused_ok(T, Cc, n_head, ah_ * aw_):
        att, sv_att = ltae_attention_forward_fused(down.contiguous(), dates, pad, p, denom, n_head, d_k)
    else:
        att, sv_att = ltae_attention_forward(down, dates, pad, p, denom, n_head, d_k)
    if mode == "att_group":
        w_att, shared = att, False
        sg = _geom_for(e.shape[-2] * e.shape[-1])
        true_h = sg.H if sg is not None else e.shape[-2]
        if true_h <= att_down:      # feature map not larger than the attention map: the reference takes its AvgPool
            p_drop, dmask = 0.0, None    # branch (kernel 1 = identity at equal size), which has NO dropout (uncrtaints.py:197-204)
    elif mode == "att_mean":
        w_att, shared = head_mean_attention(att), True
    elif mode == "mean":
        w_att, shared = mean_mode_weights(pad, n_head, B, T, att_down, att_down, e.device), False
        p_drop = 0.0          # no dropout in this mode (uncrtaints.py:189-192)
        dmask = None
    else:
        raise NotImplementedError(mode)
    if values is None:
        g, sv_agg, gpart = aggregate_forward(e, w_att, pad, training, p_drop, seed, dmask, want_stats, shared)
        return g, dict(att=sv_att, agg=sv_agg, idx=idx, att_down=att_down, mode=mode), gpart, att
    # use_v with any aggregation mode (uncrtaints.py:324-338,414-417): the aggregate takes the mode's weights (heads averaged for
    # 'att_mean', uniform over the unpadded dates for 'mean'), the values always take the attention itself
    g0, sv_agg, _ = aggregate_forward(e, w_att, pad, training, p_drop, seed, dmask, False, shared)
    v, sv_val = ltae_values_forward(sv_att, pad, values["p"], n_head, training, values.get("bn_buffers"),
                                    values.get("p_drop", 0.0), values.get("seed", 0))
    ah, aw = att.shape[-2:]
    g, sv_inc, gpart = include_v_forward(g0, v.view(B, -1, ah, aw), values["include_w"], values["include_b"], want_stats)
    return g, dict(att=sv_att, agg=sv_agg, idx=idx, att_down=att_down, mode=mode, val=sv_val, inc=sv_inc,
                   vp=values["p"]), gpart, att


def _pool_scatter(ddown: Tensor, sv: dict, de: Tensor, e_h3: Optional[Tensor]) -> Optional[Part]:
    """de[argmax] += d(pooled).  With the h3 of the block that produced the pooled tensor this also returns the (sum de, sum de*h3)
    partials that block's backward needs (one fused pass instead of a sparse scatter + a statistics pass)."""
    H, W = de.shape[-2:]
    ad = sv["att_down"]
    planes = de.numel() // (H * W)
    if e_h3 is not None and e_h3.numel() == de.numel() and e_h3.dtype == de.dtype \
            and hb.query("uncr_pool_scatter_stats_supported", H, W, ad, ad) == 1:
        slots = hb.query("uncr_ew_slots", H * W)
        part = Part(_f32((planes, slots, 2), de.device), slots)
        C = de.shape[-3]
        if _dt(de) == F32 and _H2_BWD and planes % C == 0:
            part.amax = _f32((planes // C, C * slots), de.device)     # per-block max |de|, one row per frame
        hb.call("uncr_pool_scatter_stats", ddown.contiguous(), sv["idx"], de, e_h3, part.buf, planes, H, W, ad, ad, _dt(de),
                part.amax, _stream())
        return part
    maxpool_backward_into(ddown, sv["idx"], de, H, W, ad, ad)
    return None


def ltae_stage_backward(dg: Tensor, sv: dict, p: Dict[str, Tensor], n_head: int, d_k: int, e_h3: Optional[Tensor] = None):
    """-> (de, {param grads}, partials (sum de, sum de*h3) or None).  e_h3: h3 of the block that produced e."""
    if _LTAE_REPLAY == "replay":
        de_, g_, part_ = _LTAE_STORE["bwd"]
        # fresh aliases: autograd may then adopt a gradient as .grad instead of cloning it (7 copy nodes in the captured graph otherwise)
        return de_.detach(), {k: v.detach() for k, v in g_.items()}, part_
    if _LTAE_REPLAY == "record":
        with dev_options(ltae_replay=None):
            _LTAE_STORE["bwd"] = ltae_stage_backward(dg, sv, p, n_head, d_k, e_h3)
        return _LTAE_STORE["bwd"]
    sgeom = sv["agg"].get("geom") if isinstance(sv.get("agg"), dict) else None
    if sgeom is not None and _geom() != sgeom:      # re-enter the forward's any-size geometry (csrc/anysize.hip)
        with geom_scope(sgeom):
            return ltae_stage_backward(dg, sv, p, n_head, d_k, e_h3)
    if "val" in sv:     # use_v: include_v -> (aggregation, values) -> attention
        dg0, dv, dWinc, dbinc = include_v_backward(dg, sv["inc"])
        de, datt = aggregate_backward(dg0, sv["agg"])
        dy1, datt_v, gv = ltae_values_backward(dv.reshape(dv.shape[0], dv.shape[1], -1), sv["val"], sv["vp"], n_head)
        vmode = sv.get("mode", "att_group")
        if vmode == "mean":          # the aggregate's weights are constants: the attention is reached through the values alone
            datt = datt_v.contiguous().view_as(datt)
        else:
            if vmode == "att_mean":
                datt = head_mean_attention_backward(datt)
            hb.call("uncr_add", datt, datt_v, datt, datt.numel(), _stream())
        ddown, g = ltae_attention_backward(datt, sv["att"], p, n_head, d_k, dy1_extra=dy1)
        part = _pool_scatter(ddown, sv, de, e_h3)
        g.update(gv)
        g["include_w"], g["include_b"] = dWinc, dbinc
        join_side()
        return de, g, part
    mode = sv.get("mode", "att_group")
    agg = sv["agg"]
    # (fp32 storage: 342 -> 304 us for the pair of full-resolution launches; bf16 storage keeps the one-pass kernel + the four-chunk
    # scatter: its 8-byte rows leave the second pass short of bytes in flight, 214 -> 302 us)
    if (_AGG_TWO_PASS and mode in ("att_group", "att_mean") and "pool_k" not in agg and "geom" not in agg and "any" not in agg and e_h3 is not None
            and _dt(agg["e"]) == F32
            and e_h3.numel() == agg["e"].numel() and e_h3.dtype == agg["e"].dtype and e_h3.is_contiguous()
            and hb.query("uncr_aggregate_bwd_de_supported", agg["dims"][3], agg["dims"][4], sv["att_down"], sv["att_down"]) == 1):
        datt = aggregate_backward_datt(dg, agg)
        if mode == "att_mean":
            datt = head_mean_attention_backward(datt)
        ddown, g = ltae_attention_backward(datt, sv["att"], p, n_head, d_k)
        de, part = aggregate_backward_de(dg, agg, ddown, sv["idx"], sv["att_down"], e_h3)
        join_side()          # the attention's parameter-gradient chain ran next to the second pass
        return de, g, part
    de, datt = aggregate_backward(dg, sv["agg"])
    if mode == "mean":
        # the attention does not reach the output: no gradient to the temporal encoder (returned as zeros)
        g = {k: torch.zeros_like(v) for k, v in p.items()}
        return de, g, None
    if mode == "att_mean":
        datt = head_mean_attention_backward(datt)
    ddown, g = ltae_attention_backward(datt, sv["att"], p, n_head, d_k)
    part = _pool_scatter(ddown, sv, de, e_h3)
    join_side()          # the attention's parameter-gradient chain ran next to the scatter
    return de, g, part


# ------------------------------------------------------------------------------------------------
# the attention classes called on their own (ltae.py:244-307, 312-385, 388-458): pixel-major rows [m, T, d]
# ------------------------------------------------------------------------------------------------

def _rows_to_planes(x2d: Tensor, Rp: int) -> Tensor:
    """[R, D] rows -> [1, D, Rp] channel-major planes (zero-padded to the GEMM's pixel tile): the layout of pw_gemm / pw_wgrad"""
    R, D = x2d.shape
    out = _f32((1, D, Rp), x2d.device)
    hb.call("uncr_transpose2d", x2d.contiguous(), out, R, D, Rp, _stream())
    return out


def _planes_to_rows(xT: Tensor, R: int) -> Tensor:
    _, D, Rp = xT.shape
    out = _f32((Rp, D), xT.device)
    hb.call("uncr_transpose2d", xT, out, D, Rp, D, _stream())
    return out[:R]


def linear_rows_forward(x2d: Tensor, W: Tensor, b: Optional[Tensor]):
    """nn.Linear on rows: y [R, Dout] = x [R, Din] W^T + b, as a 1x1 convolution on the transposed tensor (MFMA GEMM)."""
    R, Din = x2d.shape
    Dout = W.shape[0]
    Rp = (R + 255) // 256 * 256
    xT = _rows_to_planes(x2d.float(), Rp)
    yT, _ = pw_gemm(xT, pack_wt(W, transpose=True), 1, Din, Dout, Rp, bias=b.contiguous() if b is not None else None)
    return _planes_to_rows(yT, R), dict(xT=xT, dims=(R, Rp, Din, Dout))


def linear_rows_backward(dy: Tensor, sv: dict, W: Tensor, need_dx: bool = True):
    """-> (dx [R, Din] or None, dW [Dout, Din], db [Dout]); the zero padding of the transposed operands contributes nothing."""
    R, Rp, Din, Dout = sv["dims"]
    dyT = _rows_to_planes(dy.reshape(R, Dout).float(), Rp)
    dW, db = pw_wgrad(dyT, sv["xT"], 1, Dout, Din, Rp, rowsum=True)
    dx = None
    if need_dx:
        dxT, _ = pw_gemm(dyT, pack_wt(W, transpose=False), 1, Dout, Din, Rp)
        dx = _planes_to_rows(dxT, R)
    return dx, dW, db


def sdpa_rows_forward(q: Tensor, k: Tensor, v: Optional[Tensor], pad: Optional[Tensor], temperature: float, p_drop: float,
                      seed, want_out: bool, want_comp: bool):
    """q [m, dk] (or [q_rows, dk] shared by consecutive row groups), k [m, T, dk], v [m, T, dv], pad [m, T] int32.
    -> attention as returned (after dropout) [m, T], attn @ v [m, dv] or None, masked scaled scores [m, T] or None, saved"""
    m, T, dk = k.shape
    dev = k.device
    q, k = q.contiguous().float(), k.contiguous().float()
    dv = v.shape[-1] if v is not None else 0
    if v is not None:
        v = v.contiguous().float()
    attn_sm = _f32((m, T), dev)
    attn_out = _f32((m, T), dev) if p_drop > 0.0 else None
    out = _f32((m, dv), dev) if want_out else None
    comp = _f32((m, T), dev) if want_comp else None
    seed_val, seed_dev = seed if isinstance(seed, tuple) else (seed, None)
    hb.call("uncr_sdpa_rows_fwd", q, q.shape[0], k, v, pad, float(temperature), attn_sm, attn_out, out, comp, m, T, dk, dv,
            float(p_drop), seed_val, seed_dev, _stream())
    sv = dict(q=q, k=k, v=v, pad=pad, attn_sm=attn_sm, temperature=float(temperature), p_drop=float(p_drop), seed=seed_val,
              seed_dev=seed_dev, dims=(m, T, dk, dv))
    return (attn_out if attn_out is not None else attn_sm), out, comp, sv


def sdpa_rows_backward(dattn: Optional[Tensor], dout: Optional[Tensor], dcomp: Optional[Tensor], sv: dict, need_dv: bool = True):
    """-> dq (shape of q), dk [m, T, dk], dv [m, T, dv] or None"""
    m, T, dk, dv = sv["dims"]
    q = sv["q"]
    dev = q.device
    dq_rows, dkk = _f32((m, dk), dev), _f32((m, T, dk), dev)
    dvv = _f32((m, T, dv), dev) if (need_dv and sv["v"] is not None) else None
    c = lambda t: t.contiguous().float() if t is not None else None
    hb.call("uncr_sdpa_rows_bwd", c(dattn), c(dout) if sv["v"] is not None else None, c(dcomp), q, q.shape[0], sv["k"], sv["v"],
            sv["pad"], sv["attn_sm"], sv["temperature"], dq_rows, dkk, dvv, m, T, dk, dv, sv["p_drop"], sv["seed"],
            sv["seed_dev"], _stream())
    if q.shape[0] != m:      # one query per group of m / q_rows consecutive rows: sum each group
        dq = _f32((q.shape[0], dk), dev)
        hb.call("uncr_colsum_batched", dq_rows, q.shape[0], m // q.shape[0], dk, dq, _stream())
    else:
        dq = dq_rows
    return dq, dkk, dvv


# ------------------------------------------------------------------------------------------------
# out_conv + output nonlinearities (uncrtaints.py:381,432-445)
# ------------------------------------------------------------------------------------------------

_HEAD_OPS = {"softplus": (8, 9), "elu": (11, 12), "identity": (13, 14)}     # (forward, backward) element-wise op codes


def head_forward(y: Tensor, w: Tensor, b: Tensor, n_mean: int, mean_sigmoid: bool, scale: float, eps: float,
                 var_mode: str = "softplus"):
    N, C, H, W = _check4(y)
    P = H * W
    Co = w.shape[0]
    Wt = pack_wt(w.reshape(Co, C), transpose=True)
    out = _f32((N, Co, H, W), y.device)
    nm = n_mean if mean_sigmoid else -n_mean
    vm = {"softplus": 0, "elu": 1, "identity": 2}[var_mode]
    if Co <= 64:       # convolution + nonlinearities in ONE kernel (the pre-activation is a second output, for the backward)
        o = _f32((N, Co, H, W), y.device)
        hb.call("uncr_head_fwd", y, Wt, b.contiguous(), out, o, N, C, Co, P, nm, float(scale), float(eps), vm, _dt(y),
                _stream())
        from_out = False
    else:
        if _dt(y) == BF16:
            raise NotImplementedError("bf16 activations: out_conv wider than 64 channels is not built")
        o, _ = pw_gemm(y, Wt, N, C, Co, P, bias=b.contiguous())
        ew(_HEAD_OPS[var_mode][0], o, out=out, planes=N * Co, P=P, C=Co, n_mean=nm, scale=scale, eps=eps)
        from_out = False
    return out, dict(y=y, o=o, nm=nm, scale=scale, eps=eps, from_out=from_out, dims=(N, C, Co, H, W),
                     y_h3=getattr(y, "_uncr_h3", None), var_mode=var_mode)


def head_backward(dout: Tensor, sv: dict, w: Tensor, need_dy: bool = True):
    N, C, Co, H, W = sv["dims"]
    P = H * W
    dout = dout.contiguous()
    # the gradient w.r.t. the head's pre-activation is an activation gradient: stored like the decoder's activations
    do = _act((N, Co, H, W), dout.device, _dt(sv["y"]))
    ew(_HEAD_OPS[sv.get("var_mode", "softplus")][1], dout, b=sv["o"], out=do, planes=N * Co, P=P,
       C=-Co if sv.get("from_out") else Co, n_mean=sv["nm"], scale=sv["scale"], eps=sv.get("eps", 0.0))
    dW, db = pw_wgrad(do, sv["y"], N, Co, C, P, rowsum=True)
    dy, dy_part = None, None
    if need_dy:
        Wk = pack_wt(w.reshape(Co, C), transpose=False)        # [k=26][out=128]
        y_h3 = sv.get("y_h3")       # last decoder block's h3: emit (sum dy, sum dy*h3) in the GEMM epilogue
        dy, dy_part = pw_gemm(do, Wk, N, Co, C, P, epi=2 if y_h3 is not None else 0, aux=y_h3, want_amax=True)
        dy = dy.view(N, C, H, W)
    return dy, dW.view_as(w), db, dy_part


# ------------------------------------------------------------------------------------------------
# MGNLL (losses.py:149-218)
# ------------------------------------------------------------------------------------------------
_RED = {"none": 0, "mean": 1, "sum": 2}


ELT_GNLL, ELT_L1, ELT_L2 = 0, 1, 2


def eltloss_forward(kind: int, pred: Tensor, target: Tensor, var: Optional[Tensor], eps: float, full: bool,
                    reduction: str, check_negative: bool = False):
    """Element-wise criteria of get_loss (losses.py:14-32).  -> (loss, clamped variance or None)."""
    pred, target = pred.contiguous().float(), target.contiguous().float()
    n, dev = pred.numel(), pred.device
    red = _RED[reduction]
    nb = hb.query("uncr_eltloss_blocks", n)
    part = _f32((nb,), dev)
    loss_none = torch.empty_like(pred) if red == 0 else None
    loss = _f32((), dev) if red else None
    vclamp = torch.empty_like(var) if var is not None else None
    flag = torch.zeros((1,), device=dev, dtype=torch.int32) if (check_negative and var is not None) else None
    hb.call("uncr_eltloss_fwd", kind, pred, target, var, loss_none, vclamp, part, loss, flag, n, 1, float(eps),
            1 if full else 0, red, _stream())
    if flag is not None and int(flag.item()):        # opt-in host sync (losses.py:110-111)
        raise ValueError("var has negative entry/entries")
    return (loss_none if red == 0 else loss), vclamp


def eltloss_backward(kind: int, gout: Tensor, pred: Tensor, target: Tensor, var: Optional[Tensor], eps: float,
                     reduction: str, need_dpred: bool = True, need_dvar: bool = True):
    red = _RED[reduction]
    dpred = torch.empty_like(pred) if need_dpred else None
    dvar = torch.empty_like(var) if (need_dvar and var is not None) else None
    gout = gout.contiguous().to(torch.float32)
    hb.call("uncr_eltloss_bwd", kind, pred, target, var, gout if red else None, gout if red == 0 else None, dpred,
            dvar, pred.numel(), float(eps), red, _stream())
    return dpred, dvar


def _batch_strided(t: Tensor) -> bool:
    """[B,1,K,H,W] fp32 whose samples are dense [K,H,W] blocks at any distance >= K*H*W: a channel slice of the head's output."""
    if t.dim() != 5 or t.dtype != torch.float32 or t.shape[1] != 1:
        return False
    B, _, K, H, W = t.shape
    st = t.stride()
    return st[4] == 1 and st[3] == W and st[2] == H * W and (B == 1 or st[0] >= K * H * W)


def _bs(t: Tensor):
    """(pointer, batch stride) of a dense or batch-strided tensor for the *_bstride arguments."""
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else t.shape[2] * t.shape[3] * t.shape[4])


def mgnll_forward(pred: Tensor, target: Tensor, var: Tensor, eps: float, reduction: str, check_negative: bool,
                  want_variance: bool = False):
    """-> (loss, clamped per-band variance [B,1,K,H,W] or None).  pred / var may be channel slices of the head's output (read in
    place); anything else must be contiguous."""
    B, T1, K, H, W = pred.shape
    Kv = var.shape[2]
    dev = pred.device
    P = H * W
    if not pred.is_cuda:
        raise RuntimeError("uncrtaints_amd kernels need tensors on the GPU (cuda device); there is no CPU path")
    pred = pred if _batch_strided(pred) else pred.contiguous()
    var = var if _batch_strided(var) else var.contiguous()
    nb = hb.query("uncr_mgnll_blocks", P)
    part = _f32((nb,), dev)
    red = _RED[reduction]
    loss_none = _f32((W, H, B), dev) if red == 0 else None
    loss = _f32((), dev) if red else None
    flag = torch.zeros((1,), device=dev, dtype=torch.int32) if check_negative else None
    vclamp = _f32((B, 1, K, H, W), dev) if want_variance else None
    (pp, sp), (pv, sv) = _bs(pred), _bs(var)
    hb.call("uncr_mgnll_fwd", pp, target, pv, loss_none, vclamp, part, loss, flag, B, K, Kv, H, W, float(eps), red,
            sp, sv, _stream())
    if check_negative and int(flag.item()):      # opt-in host sync (losses.py:199-200)
        raise ValueError("var has negative entry/entries")
    if red == 0:
        return (loss_none[..., 0] if B == 1 else loss_none), vclamp      # reference .squeeze() drops B == 1
    return loss, vclamp


def mgnll_backward(gout: Tensor, pred: Tensor, target: Tensor, var: Tensor, eps: float, reduction: str,
                   need_dpred: bool = True, need_dvar: bool = True):
    """With both gradients requested they are the two channel slices of ONE [B,1,K+Kv,H,W] buffer (written in place by the kernel):
    when pred / var were split off the head's output with losses.split_prediction, its backward hands that buffer on as the
    gradient of the output without a copy."""
    B, T1, K, H, W = pred.shape
    Kv = var.shape[2]
    red = _RED[reduction]
    dpred = dvar = None
    if need_dpred and need_dvar:
        full = _f32((B, 1, K + Kv, H, W), pred.device)
        dpred, dvar = full[:, :, :K], full[:, :, K:]
    elif need_dpred:
        dpred = _f32((B, 1, K, H, W), pred.device)
    elif need_dvar:
        dvar = _f32((B, 1, Kv, H, W), pred.device)
    gout = gout.contiguous().to(torch.float32)
    if red == 0 and B == 1:
        gout = gout.reshape(W, H, 1)
    (pp, sp), (pv, sv) = _bs(pred), _bs(var)
    pdp, sdp = _bs(dpred) if dpred is not None else (None, 0)
    pdv, sdv = _bs(dvar) if dvar is not None else (None, 0)
    hb.call("uncr_mgnll_bwd", pp, target, pv, gout if red else None, gout if red == 0 else None, pdp, pdv, B, K,
            Kv, H, W, float(eps), red, sp, sv, sdp, sdv, _stream())
    return dpred, dvar


def ensemble_combine(means: Tensor, variances: Optional[Tensor], mode: str = "both"):
    """means/variances [M, ...] -> (mean_ens, var_ens) (ensemble_reconstruct.py:116-133)."""
    M = means.shape[0]
    n = means[0].numel()
    narrow = []
    if variances is not None and variances.shape != means.shape:
        # isotropic members carry one variance channel: broadcast it over the bands like the reference's numpy code does
        if variances.dim() == means.dim():
            narrow = [d - 1 for d in range(1, means.dim()) if variances.shape[d] == 1 and means.shape[d] != 1]
        try:
            variances = variances.expand_as(means)
        except RuntimeError:
            raise ValueError(f"ensemble_combine: variances {tuple(variances.shape)} do not broadcast to means "
                             f"{tuple(means.shape)}") from None
    mu, v = torch.empty_like(means[0]), torch.empty_like(means[0])
    code = {"both": 0, "aleatoric": 1, "epistemic": 2}[mode]
    hb.call("uncr_ensemble_combine", means.contiguous(), variances.contiguous() if variances is not None else None, M,
            n, code, mu, v, _stream())
    if mode == "aleatoric":        # the mean of the members' variances keeps THEIR shape (one channel for isotropic members)
        for d in narrow:
            v = v.narrow(d, 0, 1)
    return mu, v
