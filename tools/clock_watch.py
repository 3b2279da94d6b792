"""Power and shader clock of the step's kernels, one at a time (run on the GPU box):
    python tools/clock_watch.py gpurun_out/r04_power.json
Every kernel loops alone for ~3 s while a background thread samples `rocm-smi --showclocks --showpower` (socket power, sclk); the
first second of samples is dropped.  The GEMM family is measured on random data AND on all-zero activations: on a power-capped part
the same instruction stream runs faster on data that toggles fewer wires.  Power cap: `rocm-smi --showmaxpower`."""
import json, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncrtaints_amd import engine as E
from uncrtaints_amd import hip_backend as hb

N, P, dev = 4, 65536, "cuda"
torch.manual_seed(0)
samples, stop = [], False


def smi(*a):
    return subprocess.run(["rocm-smi", *a], capture_output=True, text=True, timeout=10).stdout


def sampler():
    while not stop:
        try:
            o = smi("--showclocks", "--showpower")
            sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", o)
            pw = re.findall(r"Power \(W\):\s*([0-9.]+)", o)
            if sclk and pw:
                samples.append((time.time(), int(sclk[0]), float(pw[0])))
        except Exception:
            pass
        time.sleep(0.25)


results = []


def loop(name, fn, by=0, secs=3.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(100):
            fn()
        n += 100
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    t1 = time.time()
    ms = e0.elapsed_time(e1) / n
    s = [x for x in samples if t0 + 1.0 <= x[0] <= t1]
    r = {"kernel": name, "us_per_launch": round(ms * 1e3, 1), "TBps_algorithmic": round(by / ms / 1e9, 2) if by else None,
         "power_w": round(sum(x[2] for x in s) / max(len(s), 1), 0), "sclk_mhz": round(sum(x[1] for x in s) / max(len(s), 1), 0),
         "samples": len(s)}
    results.append(r)
    print(r, flush=True)


def main(out_path):
    global stop
    cap = re.findall(r"Power \(W\):\s*([0-9.]+)", smi("--showmaxpower"))
    th = threading.Thread(target=sampler, daemon=True); th.start()
    time.sleep(1.5)
    idle = samples[-1] if samples else None
    t = lambda *s: torch.randn(*s, device=dev)
    amax = lambda x: x.abs().amax(dim=(1, 2)).view(N, 1).contiguous()
    for data in (() if "--bf16" in sys.argv else ("random", "zero")):
        z = (lambda x: x) if data == "random" else torch.zeros_like
        h2 = z(t(N, 256, P)); x = z(t(N, 128, P)); x2 = z(t(N, 128, P))
        W2 = E.pack_wt(t(128, 256) * 0.05, transpose=True); W1 = E.pack_wt(t(256, 128) * 0.05, transpose=True)
        k2 = tuple(torch.rand(N * 256, device=dev) for _ in range(3)); k1 = tuple(t(N * 128) for _ in range(3))
        ub2 = (k2[0].view(N, 256) * 4.5 + k2[1].view(N, 256)).reshape(-1).contiguous()
        ub1 = (k1[0].abs().view(N, 128) * 4.5 + k1[1].abs().view(N, 128)).reshape(-1).contiguous()
        o128, o256 = torch.empty(N, 128, P, device=dev), torch.empty(N, 256, P, device=dev)
        a45 = torch.full((N, 1), 4.5, device=dev)
        ek = tuple(torch.rand(N * 256, device=dev) for _ in range(4))
        loop(f"pw2 fwd [{data}]", lambda: E.pw_gemm(h2, W2, N, 256, 128, P, pro=2, k=k2, epi=1, in_amax=ub2, out=o128), 4.0 * N * P * 384)
        loop(f"pw1 fwd [{data}]", lambda: E.pw_gemm(x, W1, N, 128, 256, P, pro=1, k=k1, epi=1, in_amax=ub1, out=o256), 4.0 * N * P * 384)
        loop(f"dz + pass-B [{data}]", lambda: E.pw_gemm(x, W1, N, 128, 256, P, pro=3, k=k1, x2=x2, epi=3, aux=h2, ek=ek, in_amax=a45, in2_amax=a45,
                                                        out=o256), 4.0 * N * P * 768)
        d, d2 = z(t(N, 256, P)), z(t(N, 256, P))
        dk = tuple(t(N * 256) for _ in range(3))
        W1k = E.pack_wt(t(256, 128) * 0.05, transpose=False)
        c = tuple(t(N * 128) for _ in range(3))
        part = torch.empty(N * 128, hb.query("uncr_pw_stat_slots", N, 128, P), 2, device=dev)
        loop(f"dx [{data}]", lambda: hb.call("uncr_pw_gemm_dx", d, d2, W1k, o128, dk[0], dk[1], dk[2], None, x, x2, x, c[0], c[1], c[2], None, None,
                                             None, None, part, N, 256, 128, P, 0, None, a45, 1, a45, 1, P, E._stream()), 4.0 * N * P * 1024)
        xk = (t(N * 128), t(N * 128), None)
        loop(f"wgrad [256x128] [{data}]", lambda: E.pw_wgrad(d, x, N, 256, 128, P, pro_d=3, dk=dk, d2=d2, pro_x=1, xk=xk), 4.0 * N * P * 640)
        dk1 = tuple(v[:N * 128] for v in dk)
        loop(f"wgrad [128x256] [{data}]", lambda: E.pw_wgrad(x, d, N, 128, 256, P, pro_d=3, dk=dk1, d2=x2, pro_x=2, xk=(k2[0], k2[1], None),
                                                            d_amax=a45, d2_amax=a45, x_ub=ub2), 4.0 * N * P * 512)
        del h2, x, x2, d, d2, o128, o256
    if "--bf16" in sys.argv:       # the same launches on bf16 activation storage (BASELINE config 3)
        bf = torch.bfloat16
        tb = lambda *s: torch.randn(*s, device=dev).to(bf)
        h2, x, x2 = tb(N, 256, P), tb(N, 128, P), tb(N, 128, P)
        W2 = E.pack_wt(t(128, 256) * 0.05, transpose=True); W1 = E.pack_wt(t(256, 128) * 0.05, transpose=True)
        k2 = tuple(torch.rand(N * 256, device=dev) for _ in range(3)); k1 = tuple(t(N * 128) for _ in range(3))
        ek = tuple(torch.rand(N * 256, device=dev) for _ in range(4))
        o128, o256 = torch.empty(N, 128, P, device=dev, dtype=bf), torch.empty(N, 256, P, device=dev, dtype=bf)
        loop("bf16 pw2 fwd", lambda: E.pw_gemm(h2, W2, N, 256, 128, P, pro=2, k=k2, epi=1, out=o128), 2.0 * N * P * 384)
        loop("bf16 pw1 fwd", lambda: E.pw_gemm(x, W1, N, 128, 256, P, pro=1, k=k1, epi=1, out=o256), 2.0 * N * P * 384)
        loop("bf16 dz + pass-B", lambda: E.pw_gemm(x, W1, N, 128, 256, P, pro=3, k=k1, x2=x2, epi=3, aux=h2, ek=ek, out=o256), 2.0 * N * P * 768)
        d, d2 = tb(N, 256, P), tb(N, 256, P)
        dk = tuple(t(N * 256) for _ in range(3))
        W1k = E.pack_wt(t(256, 128) * 0.05, transpose=False)
        c = tuple(t(N * 128) for _ in range(3))
        part = torch.empty(N * 128, hb.query("uncr_pw_stat_slots", N, 128, P), 2, device=dev)
        loop("bf16 dx", lambda: hb.call("uncr_pw_gemm_dx", d, d2, W1k, o128, dk[0], dk[1], dk[2], None, x, x2, x, c[0], c[1], c[2], None, None,
                                        None, None, part, N, 256, 128, P, 1, None, None, 0, None, 0, P, E._stream()), 2.0 * N * P * 1024)
        xk = (t(N * 128), t(N * 128), None)
        loop("bf16 wgrad [256x128]", lambda: E.pw_wgrad(d, x, N, 256, 128, P, pro_d=3, dk=dk, d2=d2, pro_x=1, xk=xk), 2.0 * N * P * 640)
        dk1 = tuple(v[:N * 128] for v in dk)
        loop("bf16 wgrad [128x256]", lambda: E.pw_wgrad(x, d, N, 128, 256, P, pro_d=3, dk=dk1, d2=x2, pro_x=2, xk=(k2[0], k2[1], None)), 2.0 * N * P * 512)
        C, H, W = 256, 256, 256
        h1, hh2, du2 = tb(N, C, H, W), tb(N, C, H, W), tb(N, C, H, W)
        o4d = torch.empty(N, C, H, W, device=dev, dtype=bf)
        cA, cB, q1, q2, q3 = (t(N * C) for _ in range(5))
        w9 = t(C, 9)
        sf, sb = hb.query("uncr_dw_slots_fwd", H), hb.query("uncr_dw_slots_bwd", H)
        partf, partb, dwp = torch.empty(N * C, sf, 2, device=dev), torch.empty(N * C, sb, 2, device=dev), torch.empty(N * C, sb, 9, device=dev)
        loop("bf16 dw_fwd", lambda: hb.call("uncr_dw_fwd", h1, cA, cB, w9, o4d, partf, N, C, H, W, 1, 0, E._stream()), 4.0 * N * C * P)
        loop("bf16 dw_bwd", lambda: hb.call("uncr_dw_bwd", du2, hh2, h1, q1, q2, q3, None, cA, cB, w9, o4d, partb, dwp, None, 0, N, C, H, W, 1, 0, None,
                                            E._stream()), 8.0 * N * C * P)
        stop = True
        json.dump({"method": __doc__.strip(), "power_cap_w": float(cap[0]) if cap else None, "kernels": results}, open(out_path, "w"), indent=1)
        return
    C, H, W = 256, 256, 256
    h1, hh2, du2, out = t(N, C, H, W), t(N, C, H, W), t(N, C, H, W), torch.empty(N, C, H, W, device=dev)
    cA, cB, q1, q2, q3 = (t(N * C) for _ in range(5))
    w9 = t(C, 9)
    sf, sb = hb.query("uncr_dw_slots_fwd", H), hb.query("uncr_dw_slots_bwd", H)
    partf, partb, dwp = torch.empty(N * C, sf, 2, device=dev), torch.empty(N * C, sb, 2, device=dev), torch.empty(N * C, sb, 9, device=dev)
    am = torch.empty(N * C * sb, device=dev)
    loop("dw_fwd", lambda: hb.call("uncr_dw_fwd", h1, cA, cB, w9, out, partf, N, C, H, W, 0, 0, E._stream()), 8.0 * N * C * P)
    loop("dw_bwd", lambda: hb.call("uncr_dw_bwd", du2, hh2, h1, q1, q2, q3, None, cA, cB, w9, out, partb, dwp, None, 0, N, C, H, W, 0, 0, am,
                                   E._stream()), 16.0 * N * C * P)
    d = hb.dev_lib()
    n = 1 << 28
    bufs = [torch.empty(n, device=dev).normal_() for _ in range(5)]
    s = torch.cuda.current_stream().cuda_stream
    for mode, nm, nb in ((0, "stream probe read-only nt", 1), (2, "stream probe copy nt", 2), (3, "stream probe 2r:1w nt", 3)):
        loop(nm, lambda: d.fn["uncr_debug_stream_probe"](*[b.data_ptr() for b in bufs], n, mode, 1, 2048, s), nb * n * 4.0)
    o4 = torch.zeros(4, device=dev)
    loop("bf16 MFMA peak probe", lambda: d.fn["uncr_debug_mfma_probe_bf16"](o4.data_ptr(), 1024, 2000, s))
    stop = True
    res = {"method": __doc__.strip(), "power_cap_w": float(cap[0]) if cap else None,
           "idle": {"sclk_mhz": idle[1], "power_w": idle[2]} if idle else None, "kernels": results}
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/power.json")
