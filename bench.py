#!/usr/bin/env python
"""Benchmark of the UnCRtainTS hot path on MI355X: one step = forward + MGNLL + backward (+ gradient
all-reduce for N > 1) + Adam step on one batch of synthetic SAR+optical stacks.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line: BASELINE.json's metric (samples/s fwd+bwd at B x T x 15 x 256 x 256, T=3, B=4 per
GPU, fp32), plus `roofline` for the dominant kernel (HIP-event timed inside the timed region) and
`cpu_baseline` (the CPU oracle timed on the host cores; N=1 only)."""
import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_bf16 / _f16 dense peak (what the split GEMMs really run on: 6, 3 or 2 products)

PRO_NORMBWD = 3
def traffic_file(bf16=False):
    """The newest profiles/<tag>_traffic[_bf16].json (tools/measure_traffic.sh <tag> [bf16]) that was measured on the present kernel
    sources (its `_source_sha`), else the newest one (the caller then reports why the number is null)."""
    import glob
    from uncrtaints_amd.build import source_sha
    suffix = "_traffic_bf16.json" if bf16 else "_traffic.json"
    cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*" + suffix))
                   if bf16 or not f.endswith("_traffic_bf16.json"))
    sha = source_sha()
    for f in reversed(cands):
        try:
            with open(f) as fh:
                if json.load(fh).get("_source_sha") == sha:
                    return os.path.basename(f)
        except (OSError, ValueError):
            continue
    return os.path.basename(cands[-1]) if cands else "none_traffic.json"


# entry points that are another profiled kernel plus a consumer-side BatchNorm finalisation (csrc/bn_inline.h): same byte model;
# value = (the kernel they extend, leading int arguments to drop -- the partial-slot counts of the finalisation)
ALIASES = {"uncr_dw_fwd_bn": ("uncr_dw_fwd", 1)}


def profile_tag(name, args):
    """Record name of a launch whose int arguments do not tell its variant: uncr_pw_gemm_dx with the ReLU mask taken from x itself
    (relu_a given, no xh3: the encoder's first block behind in_conv's moment path) reads one operand stream less."""
    if name == "uncr_pw_gemm_dx" and args[10] is None and args[15] is not None:
        return "uncr_pw_gemm_dx:relu_x"
    return name


# activation tensors an uncr_ew launch streams (inputs + output), by op code (include/uncr_hip.h UNCR_EW_*)
EW_TENSORS = {0: 1, 1: 2, 2: 2, 3: 3, 4: 3, 5: 4, 6: 3, 7: 1, 8: 2, 9: 3, 10: 3, 11: 2, 12: 3, 13: 2, 14: 3, 15: 2, 16: 3, 17: 1}


def _alias(name, key):
    if name in ALIASES:
        base, drop = ALIASES[name]
        return base, tuple(key[drop:])
    return name, key


def kernel_model(name, key):
    """-> (label, algorithmic bytes, flops, bf16 MFMA products per fp32-equivalent MAC) of one launch from its int arguments
    (DESIGN.md section 4).  The last int of every profiled entry point is the activation storage code (0 fp32, 1 bf16)."""
    name, key = _alias(name, key)
    if name == "uncr_pw_gemm":
        bias_stride, N, Cin, Cout, P, pro, epi, in_dt, out_dt = key[:9]      # (then the counts of the magnitude arrays)
        bi, bo = (2.0 if in_dt else 4.0), (2.0 if out_dt else 4.0)
        rd = Cin * (2 if pro == PRO_NORMBWD else 1) * bi + (Cout * bo if epi in (2, 3, 10) else 0)
        # products per fp32 MAC on the 16-bit matrix pipe: 2 with bf16 storage, 3 for the fp16 two-part forward GEMMs (norm prologue,
        # statistics epilogue: DESIGN 4.1b), 6 for the exact bf16 split
        # (the call carried magnitude bounds for its activation operand: their counts follow in the key)
        h2 = (not in_dt) and pro in (1, 2) and epi in (0, 1, 10) and len(key) > 9 and key[9] > 0
        # ... and for the dz GEMM of the MBConv backward when both magnitude arrays were passed
        h2 = h2 or ((not in_dt) and pro == PRO_NORMBWD and epi == 3 and len(key) > 10 and key[9] > 0 and key[10] > 0)
        return (f"pw_gemm[{Cin}->{Cout},pro{pro},epi{epi},N{N},P{P}]", N * P * (rd + Cout * bo), 2.0 * N * P * Cin * Cout,
                (2 if in_dt else (3 if h2 else 6)) if Cout > 64 else 0)
    if name in ("uncr_pw_gemm_dx", "uncr_pw_gemm_dx:relu_x"):          # in, in2 (norm-bwd prologue), dy, x, xh3 -> dx
        N, Cin, Cout, P, act = key[:5]        # (then the counts of the two magnitude arrays: both given = two scaled fp16 parts)
        h2 = (not act) and len(key) > 6 and key[5] > 0 and key[6] > 0
        nco = 3 if name.endswith(":relu_x") else 4       # relu_x: no xh3 stream (the mask and the statistics come from x)
        return (f"pw_gemm_dx[{Cin}->{Cout},N{N},P{P}]" + (",relu_x" if nco == 3 else ""),
                (2.0 if act else 4.0) * N * P * (2 * Cin + nco * Cout), 2.0 * N * P * Cin * Cout, 2 if act else (3 if h2 else 6))
    if name == "uncr_residual_pool":       # x, h3 -> y (+ 8x8 max-pool)
        planes, H, W, OH, OW, act = key[-6:]
        return (f"residual_pool[planes{planes},{H}x{W}]", (2.0 if act else 4.0) * planes * H * W * 3, 0.0, 0)
    if name == "uncr_pw_wgrad":
        N, Cd, Cx, P, PXB, pro_d, pro_x, act = key[:8]       # (then the counts of the two magnitude arrays)
        rd = Cd * (2 if pro_d == PRO_NORMBWD else 1) + Cx * (2 if pro_x == PRO_NORMBWD else 1)
        wide = (Cd, Cx) in ((128, 256), (256, 128))
        # products per fp32 MAC: 1 with bf16 storage, 3 when the call carried its operands' magnitude bounds (two row-scaled fp16
        # parts: MBConv's dW2 products), 6 for the exact bf16 split
        h2 = (not act) and len(key) > 9 and key[8] > 0 and key[9] > 0
        return (f"pw_wgrad[{Cd}x{Cx},N{N},P{P}]" + (",fp16x2" if h2 else ""), (2.0 if act else 4.0) * N * P * rd, 2.0 * N * P * Cd * Cx,
                (1 if act else (3 if h2 else 6)) if wide else 0)
    if name == "uncr_dw_fwd":
        N, C, H, W, act = key[:5]
        return (f"dw_fwd[N{N},C{C},{H}x{W}]", (2.0 if act else 4.0) * N * C * H * W * 2, 18.0 * N * C * H * W, 0)
    if name == "uncr_dw_bwd":
        N, C, H, W, act = key[-6:-1]
        return (f"dw_bwd[N{N},C{C},{H}x{W}]", (2.0 if act else 4.0) * N * C * H * W * 4, 36.0 * N * C * H * W, 0)
    if name == "uncr_ew":
        op, planes, P, C, n_mean, act = key[:6]      # (then the valid-pixel count of padded planes)
        tensors = EW_TENSORS[op]
        nbytes = (2.0 if act else 4.0) * planes * P * tensors
        if op in (9, 12, 14):        # head backward: two fp32 inputs, output in the activation storage
            nbytes = planes * P * (8.0 + (2.0 if act else 4.0))
        return (f"ew[op{op},planes{planes},P{P}]", nbytes, 0.0, 0)
    if name == "uncr_aggregate_fwd":
        B, T, C, NH, H, W, AH, AW, act = key[-9:]
        return (f"aggregate_fwd[B{B},T{T}]", (2.0 if act else 4.0) * B * C * H * W * (T + 1), 2.0 * B * T * C * H * W, 0)
    if name == "uncr_aggregate_bwd":
        B, T, C, NH, H, W, AH, AW, act = key[-9:]
        fold = W == 256 and W == 8 * AW and H == 8 * AH      # the up-sampling's adjoint runs inside the kernel: no full-resolution datt
        return (f"aggregate_bwd[B{B},T{T}]", B * H * W * ((2.0 if act else 4.0) * C * (2 * T + 1) + (0.0 if fold else 4.0 * NH * T)),
                4.0 * B * T * C * H * W, 0)
    if name == "uncr_aggregate_bwd_datt":       # pass 1 of the two-pass backward: reads dg and e, writes only low-resolution partials
        B, T, C, NH, H, W, AH, AW, act = key[-9:]
        return (f"aggregate_bwd_datt[B{B},T{T}]", (2.0 if act else 4.0) * B * C * H * W * (T + 1), 2.0 * B * T * C * H * W, 0)
    if name == "uncr_aggregate_bwd_de":         # pass 2: reads dg and h3, writes de (+ scatter + statistics)
        B, T, C, NH, H, W, AH, AW, OH, OW, act = key[-11:]
        return (f"aggregate_bwd_de[B{B},T{T}]", (2.0 if act else 4.0) * B * C * H * W * (2 * T + 1), 3.0 * B * T * C * H * W, 0)
    return (name, 0.0, 0.0, 0)


def written_fraction(name, key):
    """Share of a launch's algorithmic bytes that are WRITES (the rest are reads) -- picks the measured stream roof of that mix."""
    name, key = _alias(name, key)
    if name == "uncr_pw_gemm":
        bias_stride, N, Cin, Cout, P, pro, epi, in_dt, out_dt = key[:9]
        bi, bo = (2.0 if in_dt else 4.0), (2.0 if out_dt else 4.0)
        rd = Cin * (2 if pro == PRO_NORMBWD else 1) * bi + (Cout * bo if epi in (2, 3, 10) else 0)
        return Cout * bo / (rd + Cout * bo)
    if name in ("uncr_pw_gemm_dx", "uncr_pw_gemm_dx:relu_x"):
        N, Cin, Cout, P, act = key[:5]
        return Cout / (2.0 * Cin + (3.0 if name.endswith(":relu_x") else 4.0) * Cout)
    if name == "uncr_residual_pool":
        return 1.0 / 3.0
    if name == "uncr_dw_fwd":
        return 0.5
    if name == "uncr_dw_bwd":
        return 0.25
    if name == "uncr_ew":
        tensors = EW_TENSORS[key[0]]
        return 0.0 if tensors == 1 else 1.0 / tensors
    if name == "uncr_aggregate_fwd":
        T = key[-8]
        return 1.0 / (T + 1)
    if name == "uncr_aggregate_bwd":
        T = key[-8]
        return T / (2.0 * T + 1)
    if name == "uncr_aggregate_bwd_de":
        T = key[-10]
        return T / (2.0 * T + 1)
    return 0.0          # weight gradients and the attention-gradient pass: read-only streams


_STREAM_ROOFS = None


def stream_roof_gbs(wfrac):
    """What a pure stream with this written share reaches on this part: piecewise-linear through the points of
    profiles/r04_stream_roofs.json (tools/stream_roofs.py: float4 lanes, 1 GB per stream, best of plain / non-temporal and of two
    grid sizes): read-only 7.0, 3r:1w 5.6, 2r:1w 5.8, copy 5.6, 1r:2w 5.4, write-only 5.9 TB/s.  None if the file is missing."""
    global _STREAM_ROOFS
    if _STREAM_ROOFS is None:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "r04_stream_roofs.json")))["best_per_mode"]
            w = {"read_only": 0.0, "3r_1w": 0.25, "2r_1w": 1.0 / 3.0, "copy_1r_1w": 0.5, "1r_2w": 2.0 / 3.0, "write_only": 1.0}
            _STREAM_ROOFS = sorted((w[k], v["TBps_total"] * 1e3) for k, v in d.items() if k in w)
        except (OSError, KeyError, ValueError):
            _STREAM_ROOFS = []
    pts = _STREAM_ROOFS
    if len(pts) < 2:
        return None
    for (w0, r0), (w1, r1) in zip(pts, pts[1:]):
        if w0 <= wfrac <= w1:
            return r0 + (r1 - r0) * (wfrac - w0) / max(w1 - w0, 1e-9)
    return pts[-1][1]


def stream_ratio(gbs, probe_gbs):
    """A kernel's rate against the cold pure-stream probe of its read : write mix.  <= 1: a fraction of that roof.  > 1: the launch ran
    above what a cold stream reaches, i.e. part of its operands came from the 256 MB Infinity Cache (its producer had just written
    them) -- reported as cache-assisted with the ratio, never as a roofline fraction."""
    r = gbs / probe_gbs
    if r <= 1.0:
        return {"frac_of_stream_roof": round(r, 4)}
    return {"frac_of_stream_roof": None, "infinity_cache_assisted": True, "ratio_to_cold_stream_probe": round(r, 4)}


PROFILED = ("uncr_pw_gemm", "uncr_pw_gemm_dx", "uncr_residual_pool", "uncr_pw_wgrad", "uncr_dw_fwd", "uncr_dw_bwd", "uncr_ew", "uncr_aggregate_fwd",
            "uncr_aggregate_bwd", "uncr_aggregate_bwd_datt", "uncr_aggregate_bwd_de") + tuple(ALIASES)


def a_step_bytes(T, P=65536, bf16=False):
    """SURVEY 8(d) contract figure: algorithmic HBM bytes of one fwd+bwd step per sample (fp32; bf16 activations halve it)."""
    return 2.5 * (2.0 if bf16 else 4.0) * P * (2206 * T + 9921)


def build_model(device, seed, act_dtype="fp32"):
    from uncrtaints_amd.src.backbones import uncrtaints as U
    from uncrtaints_amd.src.learning.weight_init import weight_init
    torch.manual_seed(seed)
    m = U.UNCRTAINTS(input_dim=15, out_conv=[26], out_nonlin_mean=True, out_nonlin_var="softplus", covmode="diag",
                     scale_by=1.0)
    m.apply(weight_init)
    return m.to(device).train().set_act_dtype(act_dtype)


def synthetic(B, T, H, W, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, T, 15, H, W, generator=g)
    y = torch.rand(B, 1, 13, H, W, generator=g)
    dates = torch.sort(torch.randint(1400, 1800, (B, T), generator=g), dim=1).values.float()
    return x.to(device), y.to(device), dates.to(device)


def cpu_baseline(T, H, W, budget_s=45.0):
    """The CPU oracle (a port, not the reference itself) on the host cores: fwd + MGNLL + bwd, train mode, B=1."""
    from oracle import uncrtaints_oracle as orc
    # 16 threads was the fastest setting measured on the GPU box's 256-thread host (8: 15.3 s, 16: 12.1 s,
    # 32: 13.5 s, 64: 20.5 s per step before the oracle moved to ATen convolutions; torch's default of 128 is 3x
    # slower still).  `cores` in the JSON is the thread count actually used.
    cores = min(16, os.cpu_count() or 1)
    cfg = orc.OracleConfig()
    p = orc.init_params(cfg, seed=1)
    pt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
          for k, v in p.items()}
    x, y, dates = orc.synthetic_batch(1, T, H, W, seed=1)

    def step():
        for v in pt.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        out = orc.forward(pt, x, dates, cfg, training=True)
        orc.loss_from_output(out, y, cfg).backward()

    def timed(threads, n_max, budget):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        step()                                      # warm-up (also sizes the sample)
        warm = time.perf_counter() - t0
        n = max(1, min(n_max, int(budget / max(warm, 1e-3)) - 1))
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], n

    # 1 warm-up + 5 timed iterations, median, at the fastest thread count, plus a k = 8 figure to compare with
    # the survey container's 8-core number (fewer iterations there: the sample stays bounded at about a minute in total)
    med, n = timed(cores, 5, budget_s)
    med8, n8 = timed(min(8, cores), 2, budget_s / 2)
    return {"value": round(1.0 / med, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (torch fp32), B=1 T={T} {H}x{W}, fwd+MGNLL+bwd train mode, 1 warm-up + {n} timed, median",
            "k8": {"value": round(1.0 / med8, 4), "cores": min(8, cores), "sample": f"same, 1 warm-up + {n8} timed, median"}}


class PowerSampler:
    """Socket power and shader clock of this rank's GPU while the timed steps run, from `rocm-smi --showclocks --showpower` polled
    by a background thread (host side only: the timed region is a run of graph replays, nothing here touches a stream).  MI355X
    enforces a socket power cap (`rocm-smi --showmaxpower`: 1400 W); a kernel mix that reaches it is clocked down, and its kernels
    are then bounded by energy per unit of work, not by the HBM or MFMA peaks (profiles/r04_power.json)."""

    def __init__(self, device_index=0, period_s=0.25):
        import threading
        self.idx, self.period, self.samples, self._stop = device_index, period_s, [], False
        self.cap = None
        try:
            m = re.findall(r"GPU\[%d\].*?Power \(W\):\s*([0-9.]+)" % device_index, self._smi("--showmaxpower"))
            self.cap = float(m[0]) if m else None
        except Exception:      # noqa: BLE001 -- no rocm-smi, no power object
            self.cap = None
        self._th = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _smi(*a):
        import subprocess
        return subprocess.run(["rocm-smi", *a], capture_output=True, text=True, timeout=10).stdout

    def _run(self):
        while not self._stop:
            try:
                o = self._smi("--showclocks", "--showpower")
                sclk = re.findall(r"GPU\[%d\].*?sclk clock level.*?\((\d+)Mhz\)" % self.idx, o)
                pw = re.findall(r"GPU\[%d\].*?Power \(W\):\s*([0-9.]+)" % self.idx, o)
                if sclk and pw:
                    self.samples.append((time.perf_counter(), int(sclk[0]), float(pw[0])))
            except Exception:      # noqa: BLE001
                pass
            time.sleep(self.period)

    def start(self):
        self._th.start()

    def stop(self):
        self._stop = True

    def summary(self, t0, t1):
        s = [x for x in self.samples if t0 + 0.5 <= x[0] <= t1]      # the first half second: the governor is still settling
        if not s:
            return {"samples": 0, "power_cap_w": self.cap, "note": "no rocm-smi samples inside the timed region (too short, or no rocm-smi)"}
        pw, ck = [x[2] for x in s], [x[1] for x in s]
        return {"samples": len(s), "avg_power_w": round(sum(pw) / len(pw), 1), "max_power_w": max(pw), "power_cap_w": self.cap,
                "avg_sclk_mhz": round(sum(ck) / len(ck)), "min_sclk_mhz": min(ck), "max_sclk_mhz_of_the_part": 2400,
                "source": "rocm-smi --showclocks --showpower polled every %.2f s" % self.period}


def bf16_leg(args):
    """BASELINE config 3's per-GPU leg (bf16 activation storage, fp32 accumulate) measured by a second run of this script in its own
    process (own memory pool and graph) and attached to the fp32 headline line as a compact sub-object: the fp32 fields stay the
    headline, `dtype` of the sub-object is bf16.  A few seconds; any failure is reported in place of the numbers."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.abspath(__file__), "--act-dtype", "bf16", "--no-cpu-baseline", "--steps", str(min(args.steps, 200)),
           "--warmup", str(args.warmup), "--batch-per-gpu", str(args.batch_per_gpu), "--T", str(args.T), "--size", str(args.size)]
    if args.no_kernel_events:
        cmd.append("--no-kernel-events")
    if args.no_graph:
        cmd.append("--no-graph")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001 -- the headline line must not die with its appendix
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    out = {k: d.get(k) for k in ("dtype", "value", "unit", "ms_per_step", "steps", "step_hbm_roofline_frac", "power")}
    out["config"] = "the same workload with bf16 activation storage, fp32 statistics / weights / accumulation (BASELINE config 3, one GPU)"
    if "roofline" in d:
        out["roofline"] = {k: d["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "mean_launch_ms")}
    if "ltae_stage" in d:
        out["ltae_stage_roofline_frac"] = d["ltae_stage"].get("roofline_frac")
    return out


# ---- rank bookkeeping of the N > 1 launch (one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torch.distributed.run).
#      Plain functions so that a gloo world of 8 CPU processes can run them (tests/test_parallel_gloo.py): the first real 8-GPU run
#      must not fail on plumbing.

def rank_env(gpus_flag):
    """(world, rank, local_rank) from the launcher's environment; --gpus N > 1 must agree with WORLD_SIZE."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if gpus_flag > 1 and world != gpus_flag:
        raise SystemExit(f"--gpus {gpus_flag} needs torch.distributed.run with WORLD_SIZE={gpus_flag} (got {world})")
    if not 0 <= rank < world:
        raise SystemExit(f"RANK={rank} outside WORLD_SIZE={world}")
    return world, rank, local_rank


def check_roster(seen, world, backend):
    """One rank per GPU: over RCCL every rank must sit on a different device (the gloo development mode shares cuda:0)."""
    if len(seen) != world or any(s is None for s in seen):
        raise SystemExit(f"device roster incomplete: {seen}")
    if backend == "nccl" and len(set(seen)) != world:
        raise SystemExit(f"{world} ranks but only {len(set(seen))} distinct devices: {seen}")


def join_ranks(backend, rank, world, device, identity, device_name):
    """Create the process group and return the `collective` header of the result line: every rank reports the device it sits on
    (`identity`), rank bookkeeping is checked on every rank."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    # fail fast: a collective error or a lost rank must end the run, not hang it (the driver times the whole command)
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "0")
    tmo = datetime.timedelta(seconds=int(os.environ.get("UNCR_BENCH_COLL_TIMEOUT_S", "180")))
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device, timeout=tmo)
    else:
        dist.init_process_group(backend, timeout=tmo)
    if dist.get_world_size() != world or dist.get_rank() != rank:
        raise SystemExit(f"process group disagrees with the environment: rank {dist.get_rank()}/{dist.get_world_size()} vs {rank}/{world}")
    seen = [None] * world
    dist.all_gather_object(seen, identity)
    check_roster(seen, world, backend)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:      # noqa: BLE001
            ver = None
    return {"devices": seen, "rccl_version": ver, "device_name": device_name}


def max_over_ranks(dt, device):
    """The timed region's length as the job sees it: the slowest rank's."""
    import torch.distributed as dist
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def job_value(world, batch_per_gpu, steps, dt):
    """Whole-job samples/s: weak scaling, every rank processed batch_per_gpu samples per step."""
    return world * batch_per_gpu * steps / dt


def collective_object(dist_info, dp, coll_wait):
    """N > 1: what travels (one fp32 bucket per backward segment, all-reduce AVG), on which devices, and how long the compute stream
    stood still for it per step (HIP events around the waits in finish(): 0 = fully hidden behind the backward)."""
    return dict(dist_info, bucket_bytes=dp.bucket_bytes(), all_reduces_per_step=len(dp.buckets),
                wait_ms_per_step=round(sum(coll_wait) / max(len(coll_wait), 1), 4),
                wait_ms_max=round(max(coll_wait), 4) if coll_wait else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--T", type=int, default=3)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of HIP-graph replays")
    ap.add_argument("--torch-adam", action="store_true", help="torch's fused multi-tensor Adam instead of the one-launch HIP kernel")
    ap.add_argument("--act-dtype", choices=("fp32", "bf16"), default="fp32",
                    help="activation storage: fp32 (default: the headline line, BASELINE config 2) or bf16 (BASELINE config 3: bf16 "
                         "activations, fp32 accumulate; reported as dtype bf16, never as the fp32 headline)")
    ap.add_argument("--no-power", action="store_true", help="do not poll rocm-smi for power / clock during the timed steps")
    ap.add_argument("--no-bf16-leg", action="store_true",
                    help="skip the compact `bf16` sub-object (BASELINE config 3's per-GPU leg) the fp32 headline line carries at N = 1")
    ap.add_argument("--dev-options", default="",
                    help="development A/B runs only: engine.dev_options switches as k=v[,k=v...], e.g. bn_consumer=0 (the line then "
                         "carries them under config.dev_options; the defaults are the shipped path)")
    args = ap.parse_args()
    dev_opts = {}
    if args.dev_options:
        from uncrtaints_amd import engine as _engine
        for kv in args.dev_options.split(","):
            k, v = kv.split("=")
            dev_opts[k] = int(v) if k == "dw_variant" else bool(int(v))
        _engine.dev_options(**dev_opts).__enter__()

    world, rank, local_rank = rank_env(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # UNCR_BENCH_BACKEND=gloo: development switch to exercise the N > 1 code path on a single-GPU box (all ranks on
    # cuda:0, gradients reduced through the host); the real launch is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("UNCR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    dist_info = {}
    # UNCR_BENCH_FORCE_DP=1 (development / tests): take the N > 1 code path -- process group, bucketed all-reduces, segmented graphs --
    # with whatever world size the launcher set, including 1.  One rank over RCCL runs the real collective library and the
    # capture / replay pattern of the multi-GPU step on a single-GPU box (tests/test_gpu_ddp.py).
    dp_mode = world > 1 or os.environ.get("UNCR_BENCH_FORCE_DP") == "1"
    if dp_mode:
        props = torch.cuda.get_device_properties(device)
        me = f"{os.uname().nodename}:{getattr(props, 'uuid', None) or getattr(props, 'pci_bus_id', None) or local_rank}:{local_rank}"
        dist_info = join_ranks(backend, rank, world, device if backend == "nccl" else None, me, props.name)

    from uncrtaints_amd import hip_backend as hb
    from uncrtaints_amd.src import losses
    hb.lib()
    B, T, H = args.batch_per_gpu, args.T, args.size
    bf16 = args.act_dtype == "bf16"
    model = build_model(device, seed=1, act_dtype=args.act_dtype)
    crit = losses.MultiGaussianNLLLoss(reduction="mean", eps=1e-8, full=True, mode="diag")
    dp = None
    if dp_mode:
        from uncrtaints_amd.parallel import BucketedDataParallel
        # with a captured forward/backward the collectives stay outside the graph: no launches from autograd hooks
        dp = BucketedDataParallel(model, seed=1, overlap=args.no_graph)
        dp.time_waits = True          # HIP event pair around the waits in finish(): the part of the all-reduce NOT hidden
        segmented = (not args.no_graph) and len(dp.buckets) == 3
        model.keep_boundaries = segmented
    else:
        model.temporal_aggregator.set_seed(1)
    use_graph = not args.no_graph
    # Adam as ONE launch over the whole parameter set (uncrtaints_amd.optim.FusedAdam: a torch.optim.Adam whose step is a HIP
    # kernel; torch's own fused multi-tensor path takes three launches of 26 us for the 91 tensors); --torch-adam: torch's.
    # Captured with the step at N = 1, its own graph behind the gradient all-reduces at N > 1.
    if args.torch_adam:
        try:
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=use_graph, fused=True)
        except Exception:
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=use_graph)
        opt_kind = "torch.optim.Adam(fused, capturable)"
    else:
        from uncrtaints_amd.optim import FusedAdam
        opt = FusedAdam(model.parameters(), lr=1e-3)
        opt_kind = "uncrtaints_amd.optim.FusedAdam (torch.optim.Adam arithmetic, one HIP launch)"
    x, y, dates = synthetic(B, T, H, H, seed=1 + rank, device=device)
    step_counter = torch.zeros(1, dtype=torch.int64, device=device)
    model.temporal_aggregator.step_counter = step_counter     # dropout stream advances on the device

    def fwd_bwd():
        step_counter.add_(1)
        if dp is not None:
            dp.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        out = model(x, batch_positions=dates)
        loss, _ = crit(*_split(out, y))
        loss.backward()
        return loss

    def _split(out, y):     # mean / variance channels as BaseModel.get_loss_G hands them to the criterion
        mean, var = losses.split_prediction(out, 13, 26)
        return mean, y, var

    def eager_step():
        loss = fwd_bwd()
        if dp is not None:
            dp.finish()
        opt.step()
        return loss

    # N > 1: the backward pass cut at the two bucket boundaries (decoder + head | temporal encoder | encoder): each segment is
    # its own captured HIP graph, and the all-reduce of a segment's gradient bucket is launched (RCCL, asynchronous) as soon as
    # that graph has been enqueued -- it travels over xGMI while the next segment computes.
    def seg_forward_decoder():
        step_counter.add_(1)
        dp.zero_grad()
        out = model(x, batch_positions=dates)
        loss, _ = crit(*_split(out, y))
        torch.autograd.backward(loss, inputs=dp.bucket_params(0) + [model._boundary_agg])
        dp.pack_bucket(0)     # one multi-tensor copy into the flat bucket, part of the segment (and of its graph)
        return loss

    def seg_stage():
        g_t = model._boundary_agg
        torch.autograd.backward(g_t, grad_tensors=g_t.grad, inputs=dp.bucket_params(1) + [model._boundary_enc])
        dp.pack_bucket(1)

    def seg_encoder():
        e_t = model._boundary_enc
        torch.autograd.backward(e_t, grad_tensors=e_t.grad, inputs=dp.bucket_params(2))
        dp.pack_bucket(2)

    def segmented_step(run0, run1, run2, run_opt):
        loss = run0()
        dp.reduce_bucket(0, pack=False)
        run1()
        dp.reduce_bucket(1, pack=False)
        run2()
        dp.reduce_bucket(2, pack=False)
        dp.finish(packed_by_graph=True)           # the compute stream waits for the three collectives
        run_opt()
        return loss

    step = eager_step
    if dp is not None and use_graph:
        eager_seg = lambda: segmented_step(seg_forward_decoder, seg_stage, seg_encoder, opt.step)
        eager_step_plain, eager_step = eager_step, eager_seg      # the same launch order eagerly (warm-up, event-profiled re-run)
        step = eager_step
    graph_note = "eager launches"
    if use_graph:
        # Capture the whole step (fwd + MGNLL + bwd + Adam, ~450 kernel launches) into one HIP graph: the step is
        # then independent of host launch speed (on a busy host eager enqueue alone was measured at 18-20 ms/step
        # against 19.5 ms of GPU time).
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if dp is None:
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(graph):
                    static_loss = eager_step()
                torch.cuda.synchronize()

                def step():
                    graph.replay()
                    return static_loss
                graph_note = "HIP graph replay of the captured step"
            else:
                # N > 1: four graphs sharing one memory pool (forward + loss + decoder backward | temporal-encoder backward |
                # encoder backward | fused Adam); the three bucket all-reduces are launched between the replays
                graphs = [graph] + [torch.cuda.CUDAGraph() for _ in range(3)]
                pool = None
                static = {}
                for gi, fn in enumerate((seg_forward_decoder, seg_stage, seg_encoder, opt.step)):
                    # thread_local: RCCL's watchdog thread may touch its own events while this thread captures
                    with torch.cuda.graph(graphs[gi], pool=pool, capture_error_mode="thread_local"):
                        r = fn()
                    if gi == 0:
                        static["loss"] = r
                        pool = graphs[0].pool()
                torch.cuda.synchronize()
                dp.zero_grad()

                def step():
                    return segmented_step(lambda: (graphs[0].replay(), static["loss"])[1], graphs[1].replay, graphs[2].replay,
                                          graphs[3].replay)
                graph_note = ("HIP graph replays of forward+decoder-backward | temporal-encoder backward | encoder backward | fused "
                              "Adam, each gradient bucket all-reduced (RCCL, asynchronous) behind its segment")
        except Exception as exc:   # fall back loudly, never silently
            print(f"[bench] HIP-graph capture failed ({type(exc).__name__}: {exc}); timing eager launches",
                  file=sys.stderr)
            step, use_graph = eager_step, False

    # L-TAE stage as a stage ("A_ltae_step = (3T+2)*128*P*4 per sample"): pooled-gradient scatter + temporal
    # attention at 32x32 + up-sampling + temporal aggregation, forward and backward.  During the event-profiled eager steps every
    # launch made inside the two stage calls is timed with its own HIP event pair and the durations are summed (an event pair
    # around the whole call would count the host's launch gaps of the eager mode).  The 8x8 max-pool itself rides on the last
    # encoder block's residual kernel and is not part of these times.
    from uncrtaints_amd import engine as _E
    _orig_stage = (_E.ltae_stage_forward, _E.ltae_stage_backward)

    def _scoped(fn, tag):
        def wrapped(*a, **k):
            p_ = hb._PROF
            if p_ is None:
                return fn(*a, **k)
            p_.scope = tag
            try:
                return fn(*a, **k)
            finally:
                p_.scope = None
        return wrapped
    _E.ltae_stage_forward, _E.ltae_stage_backward = _scoped(_orig_stage[0], "fwd"), _scoped(_orig_stage[1], "bwd")

    def fence():
        torch.cuda.synchronize()
        if dp_mode:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = None
    want_events = rank == 0 and not args.no_kernel_events
    if want_events and not use_graph:
        prof = hb.EventProfiler(PROFILED)
        prof.tag = profile_tag
        hb.set_profiler(prof)
    # the sampler (a thread forking rocm-smi every 0.25 s) never runs next to the timed region: in the eager / segmented modes the
    # timed host loop enqueues launches and would compete with it.  It is started behind the timed steps, over ~3 s of UNTIMED
    # replays of the same step
    power = PowerSampler(local_rank) if rank == 0 and not args.no_power else None
    fence()
    if dp is not None:
        dp.wait_ms()                        # drop the warm-up records
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_dt = time.perf_counter() - t0      # host time to ENQUEUE the steps (no sync inside step())
    fence()
    dt = time.perf_counter() - t0
    power_summary = None
    # power / clock reading: the same step replayed UNTIMED for about three more seconds (not part of `value`).  Rank 0 decides,
    # EVERY rank replays: the steps contain the gradient all-reduces.
    n_extra = 0
    if power is not None:
        n_extra = int(3.0 / max(dt / args.steps, 1e-4)) + 1
    if dp_mode:
        ne = torch.tensor([n_extra], device=device, dtype=torch.int64)
        dist.broadcast(ne, src=0)
        n_extra = int(ne.item())
    if n_extra:
        if power is not None:
            power.start()
        tp0 = time.perf_counter()
        for _ in range(n_extra):
            loss = step()
        fence()
        tp1 = time.perf_counter()
    if power is not None:
        power.stop()
        power_summary = power.summary(tp0, tp1)
        power_summary["sampled_over"] = f"{n_extra} further untimed replays of the same step behind the timed region"
        power_summary["sampler_overlapped_timed_region"] = False
    hb.set_profiler(None)
    coll_wait = dp.wait_ms() if dp is not None else []
    eager_ms = None
    if dp_mode:   # the max over ranks of the timed region, taken before anything else touches the stream
        dt = max_over_ranks(dt, device)
    if use_graph and not args.no_kernel_events:
        # kernels inside a graph replay cannot be bracketed one by one: re-run the SAME steps eagerly with a HIP
        # event pair around every launch (same kernels, same shapes, same stream) for the roofline numbers.  Every
        # rank runs the steps (they contain the gradient all-reduce); only rank 0 records events.
        n_ev = min(args.steps, 5)
        eager_step(); fence()
        if want_events:
            prof = hb.EventProfiler(PROFILED)
            prof.tag = profile_tag
            # the fused pooled-gradient scatter + (sum de, sum de*h3) pass replaces the last encoder block's own statistics pass
            # (a full read of de and h3 that block needs with or without the L-TAE stage): attributed to that block, not the stage
            prof.scope_exclude.add("uncr_pool_scatter_stats")
            hb.set_profiler(prof)
        t1 = time.perf_counter()
        for _ in range(n_ev):
            eager_step()
        fence()
        eager_ms = (time.perf_counter() - t1) / n_ev * 1e3
        hb.set_profiler(None)
        prof_steps = n_ev
    else:
        prof_steps = args.steps
    final_loss = float(loss.item())
    # the step is fwd + MGNLL + bwd, "optimizer step reported separately".  The timed step above INCLUDES the Adam
    # update (the number a training run sees); its own cost is measured here, eagerly, with HIP events on the current stream.
    opt_ms = None
    if not dp_mode and not args.no_kernel_events:      # (at N > 1 the update is its own captured graph behind the all-reduces)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        opt.step(); fence()
        e0.record()
        for _ in range(10):
            opt.step()
        e1.record(); fence()
        opt_ms = e0.elapsed_time(e1) / 10

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = job_value(world, B, args.steps, dt)
        res = {
            "metric": "samples/sec fwd+bwd (BxTx15x256x256, T=3)", "value": round(value, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"uncrtaints --input_t {T} --n_head 16 --block_type mbconv --covmode diag, "
                                   f"B={B}/GPU, {H}x{H}, fwd+MGNLL+bwd+Adam, train mode (dropout on), "
                                   + ("bf16 activation storage / fp32 accumulate, statistics, weights and loss" if bf16 else "fp32"),
                       "global_batch": world * B, "T": T, "parallelism": f"dp{world}",
                       **({"dev_options": dev_opts} if dev_opts else {})},
            "ranks": world, "collective_backend": (backend if dp_mode else None),
            # N > 1: what travels (one fp32 bucket per backward segment, all-reduce AVG), on which devices, and how long the compute
            # stream stood still for it per step (HIP events around the waits in finish(): 0 = fully hidden behind the backward)
            "collective": collective_object(dist_info, dp, coll_wait) if dp is not None else None,
            "final_loss": final_loss, "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3),
            "launch_mode": graph_note,
            "optimizer": {"kind": opt_kind, "included_in_step": True,
                          "ms_per_step_eager": None if opt_ms is None else round(opt_ms, 4)},
            "step_hbm_roofline_frac": round(value / world * a_step_bytes(T, H * H, bf16) / 1e9 / HBM_PEAK_GBS, 4),
        }
        if prof is not None:
            summ = prof.summarize()
            rows = []
            for (name, key), (n, mean_ms) in summ.items():
                label, nbytes, flops, prod = kernel_model(name, key)
                rows.append(dict(kernel=label, launches=n, mean_ms=mean_ms, total_ms=n * mean_ms,
                                 prod=prod,
                                 bf16_pipe_util=(prod * flops / mean_ms / 1e9 / BF16_MFMA_PEAK_TF) if (prod and mean_ms > 0) else None,
                                 gbs=nbytes / mean_ms / 1e6 if mean_ms > 0 else 0.0,
                                 tflops=flops / mean_ms / 1e9 if mean_ms > 0 else 0.0, bytes=nbytes, flops=flops,
                                 wfrac=written_fraction(name, key)))
            rows.sort(key=lambda r: -r["total_ms"])
            tot = sum(r["total_ms"] for r in rows)
            top = rows[0]
            # the bound of a kernel = whichever roof it is closer to
            # matrix roof of a launch: the bf16 pipe for the kernels that run on it (6 / 2 / 1 products per MAC), else fp32 MFMA
            hbm_frac = top["gbs"] / HBM_PEAK_GBS
            mfma_frac = top["bf16_pipe_util"] if top["bf16_pipe_util"] is not None else top["tflops"] / FP32_MFMA_PEAK_TF
            if mfma_frac > hbm_frac:
                on_bf16 = top["bf16_pipe_util"] is not None
                res["roofline"] = {"bound": "mfma", "achieved": round(top["tflops"] * (top["prod"] if on_bf16 else 1), 2),
                                   "peak": BF16_MFMA_PEAK_TF if on_bf16 else FP32_MFMA_PEAK_TF,
                                   "unit": "TFLOP/s", "frac": round(mfma_frac, 4), "traffic": None,
                                   "kernel": top["kernel"]}
            else:
                res["roofline"] = {"bound": "hbm", "achieved": round(top["gbs"], 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(hbm_frac, 4), "traffic": None, "kernel": top["kernel"]}
            # What `frac` measures: algorithmic bytes over launch time.  The decoder's launches work on B frames (<= 268 MB per
            # operand): a consumer finds part of what its producer just wrote in the 256 MB Infinity Cache, so their rate is
            # cache-ASSISTED, not an HBM rate.  The encoder runs the same kernel on B*T frames (3x the bytes, nothing survives in the
            # cache): that launch's rate is the HBM truth, reported next to `frac` as `hbm_truth_frac`.
            by_label = {r["kernel"]: r for r in rows}

            def twin_of(label):
                m_ = re.search(r"N(\d+)", label)
                if not m_:
                    return None
                n_ = int(m_.group(1))
                for n2 in (n_ * T, n_ // T if n_ % T == 0 else 0):
                    if n2 and n2 != n_:
                        other = label[:m_.start()] + f"N{n2}" + label[m_.end():]
                        # (the encoder's dx launch takes its ReLU mask from x itself: same kernel family, one operand stream less)
                        for cand in (other, other + ",relu_x", other.replace(",relu_x", "")):
                            if cand in by_label and cand != label:
                                return by_label[cand]
                return None
            tw = twin_of(top["kernel"])
            if tw is not None and res["roofline"]["bound"] == "hbm":
                big = tw if tw["bytes"] > top["bytes"] else top
                res["roofline"]["hbm_truth_frac"] = round(big["gbs"] / HBM_PEAK_GBS, 4)
                res["roofline"]["hbm_truth_kernel"] = big["kernel"]
                res["roofline"]["frac_is"] = ("algorithmic bytes / launch time of the dominant (B-frame) launch: Infinity-Cache-assisted; "
                                              "hbm_truth_frac = the same kernel's B*T-frame launch, whose operands exceed the cache")
            # the same rate against what a PURE stream of this kernel's read : write mix reaches from a COLD start on this part
            # (tools/stream_roofs.py -> profiles/r04_stream_roofs.json).  A ratio above 1 is not "beyond a roof": it is the cache help
            # described above, and is reported as such
            sr = stream_roof_gbs(top["wfrac"])
            if sr:
                res["roofline"]["written_share_of_bytes"] = round(top["wfrac"], 3)
                res["roofline"]["cold_stream_probe_gbs_for_this_mix"] = round(sr, 1)
                res["roofline"].update(stream_ratio(top["gbs"], sr))
            # HBM bytes per launch from the PMC passes (tools/measure_traffic.sh -> profiles/<round>_traffic.json; rocprofv3
            # cannot run inside this process).  The file records the hash of the kernel sources it was measured on: the number
            # is attached only while the sources still hash to it (a stale figure is worse than null) and for a kernel that
            # was measured.
            res["roofline"]["algorithmic_bytes"] = int(top["bytes"])
            try:
                from uncrtaints_amd.build import source_sha
                tfile = traffic_file(bf16)
                with open(os.path.join(ROOT, "profiles", tfile)) as fh:
                    trj = json.load(fh)
                tr = trj.get(top["kernel"])
                if tr and trj.get("_source_sha") == source_sha():
                    res["roofline"]["traffic"] = tr["hbm_bytes"]
                    res["roofline"]["traffic_source"] = f"profiles/{tfile} (rocprofv3 --pmc passes, same kernel sources)"
                else:
                    res["roofline"]["traffic_source"] = (f"null: profiles/{tfile} was measured on other kernel sources "
                                                         "or does not hold this kernel")
            except OSError:
                pass
            res["roofline"]["mean_launch_ms"] = round(top["mean_ms"], 4)
            res["roofline"]["share_of_profiled_time"] = round(top["total_ms"] / max(tot, 1e-9), 4)
            res["kernel_breakdown"] = [
                dict(kernel=r["kernel"], launches_per_step=r["launches"] / prof_steps, mean_ms=round(r["mean_ms"], 4),
                     share=round(r["total_ms"] / max(tot, 1e-9), 4), gbs=round(r["gbs"], 1), tflops=round(r["tflops"], 2),
                     frac_of_8tbs=round(r["gbs"] / HBM_PEAK_GBS, 3),
                     # the launch of the same kernel on the other frame count (decoder: B frames, cache-assisted; encoder: B*T frames)
                     twin=(lambda t_: None if t_ is None else {"kernel": t_["kernel"], "gbs": round(t_["gbs"], 1),
                                                               "frac_of_8tbs": round(t_["gbs"] / HBM_PEAK_GBS, 3)})(twin_of(r["kernel"])),
                     **(stream_ratio(r["gbs"], stream_roof_gbs(r["wfrac"])) if stream_roof_gbs(r["wfrac"]) else {}),
                     # tflops = fp32-equivalent work; the exact-split GEMMs issue 6 bf16 MFMA products per fp32 MAC:
                     # bf16_pipe_util = 6 x tflops / 2500 TF = the share of the bf16 matrix pipe's dense peak really used
                     bf16_pipe_util=None if r["bf16_pipe_util"] is None else round(r["bf16_pipe_util"], 4))
                for r in rows[:12]]
            res["profiled_ms_per_step"] = round(tot / prof_steps, 3)
            # what the step would take if every profiled launch ran at its own roof (HBM 8 TB/s or fp32-MFMA peak,
            # whichever is slower for that launch): the distance to "speed of light" of the whole launch list
            sol = sum(r["launches"] * max(r["bytes"] / (HBM_PEAK_GBS * 1e6),
                                          r["prod"] * r["flops"] / (BF16_MFMA_PEAK_TF * 1e9) if r["prod"] else r["flops"] / (FP32_MFMA_PEAK_TF * 1e9))
                      for r in rows)
            res["profiled_roofline_ms_per_step"] = round(sol / prof_steps, 3)
            res["profiled_gbytes_per_step"] = round(sum(r["launches"] * r["bytes"] for r in rows) / prof_steps / 1e9, 2)
            sms = prof.scope_ms()
            if sms:
                tf, tb = sms.get("fwd", 0.0) / prof_steps, sms.get("bwd", 0.0) / prof_steps
                a_stage = (3 * T + 2) * 128 * H * H * (2.0 if bf16 else 4.0) * B
                gbs = a_stage / ((tf + tb) * 1e6) if tf + tb > 0 else 0.0
                det = prof.scope_detail()
                t_scatter = sum(v[1] for d in det.values() for k, v in d.items() if k == "-uncr_pool_scatter_stats") / prof_steps
                gbs_all = a_stage / ((tf + tb + t_scatter) * 1e6) if tf + tb > 0 else 0.0
                two_pass = any(k == "uncr_aggregate_bwd_de" for d in det.values() for k in d)
                planes = 128 * H * H * (2.0 if bf16 else 4.0) * B      # one [C = 128][P] plane set per sample
                moved = ((T + 1) + ((T + 1) + (2 * T + 1) if two_pass else (2 * T + 1) + 2 * T)) * planes
                res["ltae_stage"] = {"ms_forward": round(tf, 4), "ms_backward": round(tb + (0.0 if two_pass else t_scatter), 4),
                                     "algorithmic_bytes": int(a_stage),
                                     "gbs": round(gbs_all if not two_pass else gbs, 1),
                                     # the contract figure: SURVEY 8(d)'s A_ltae_step over the time of EVERY launch of the stage
                                     "roofline_frac": round((gbs if two_pass else gbs_all) / HBM_PEAK_GBS, 4),
                                     "roofline_frac_with_scatter_stats": round((gbs if two_pass else gbs_all) / HBM_PEAK_GBS, 4),
                                     # what the stage's launches really stream (forward T+1, attention-gradient pass T+1, de + scatter +
                                     # statistics pass 2T+1 planes -- the last includes the T planes of h3 the encoder block's statistics
                                     # need) over the same time: how close the launches are to the HBM peak
                                     "bytes_moved_by_the_stage_launches": int(moved),
                                     "stream_frac_of_8tbs": round(moved / ((tf + tb + (0.0 if two_pass else t_scatter)) * 1e6) / HBM_PEAK_GBS, 4)
                                     if tf + tb > 0 else 0.0,
                                     "definition": "stage = temporal attention at 32x32 + up-sampling + aggregation, forward and backward, "
                                                   "INCLUDING the scatter of the pooled gradient and the (sum de, sum de*h3) statistics of the "
                                                   "last encoder block, which ride on the stage's second backward pass (round 5: two-pass "
                                                   "backward, de written once); time = sum of the per-launch HIP-event times of every kernel "
                                                   "launched inside the two stage calls; roofline_frac = A_ltae_step = (3T+2)*128*P*bytes*B "
                                                   "(SURVEY 8(d)) over that time (rounds 1-4 reported this as "
                                                   "roofline_frac_with_scatter_stats; their roofline_frac left the scatter + statistics launch "
                                                   "out, which the fused pass no longer allows)",
                                     "launches": {tag: {k: [round(v[0] / prof_steps, 2), round(1e3 * v[1] / prof_steps, 1)]
                                                        for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])}
                                                  for tag, d in det.items()}}
            if eager_ms is not None:
                res["eager_event_profiled_ms_per_step"] = round(eager_ms, 3)
                res["roofline"]["source"] = ("eager re-run of the same steps with a HIP event pair around every launch "
                                             "(graph replays cannot be bracketed per kernel)")
        if power_summary is not None:
            res["power"] = power_summary
        if not dp_mode and not bf16 and not args.no_bf16_leg:
            res["bf16"] = bf16_leg(args)
        if not dp_mode and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(T, H, H)
        print(json.dumps(res))
    if dp_mode:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
