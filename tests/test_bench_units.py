"""CPU checks of bench.py's bookkeeping (no GPU): the per-kernel byte model, the written share and the stream-roof interpolation behind
`frac_of_stream_roof`, and the rocm-smi parser behind the `power` object."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_stream_roof_interpolation_and_written_share():
    b = _bench()
    ro, wo = b.stream_roof_gbs(0.0), b.stream_roof_gbs(1.0)
    assert 6000 < ro < 7500 and 5000 < wo < 6500                     # read-only ~7.0 TB/s, write-only ~5.9 (profiles/r04_stream_roofs.json)
    mid = [b.stream_roof_gbs(w) for w in (0.125, 0.25, 1 / 3, 0.5, 2 / 3)]
    assert all(5000 < v < ro for v in mid)
    assert abs(b.stream_roof_gbs(0.125) - 0.5 * (ro + b.stream_roof_gbs(0.25))) < 1.0      # piecewise linear
    # dx GEMM: 2*256 + 3*128 rows read, 128 written; dz: 2*128 + 256 read, 256 written; depthwise forward: 1 : 1
    assert abs(b.written_fraction("uncr_pw_gemm_dx", (4, 256, 128, 65536, 0, 1, 1)) - 128 / 1024) < 1e-9
    assert abs(b.written_fraction("uncr_pw_gemm", (0, 4, 128, 256, 65536, 3, 3, 0, 0, 1, 1)) - 256 / 768) < 1e-9
    assert b.written_fraction("uncr_dw_fwd", (4, 256, 256, 256, 0)) == 0.5
    assert b.written_fraction("uncr_pw_wgrad", (4, 256, 128, 65536, 64, 3, 1, 0)) == 0.0
    label, nbytes, flops, prod = b.kernel_model("uncr_pw_gemm_dx", (4, 256, 128, 65536, 0, 1, 1))
    assert nbytes == 4.0 * 4 * 65536 * (2 * 256 + 4 * 128) and prod == 3 and "pw_gemm_dx" in label
    # every element-wise op code of include/uncr_hip.h has a byte model (the bf16 leg launches op 17, the four-chunk SE pooling pass)
    import re
    hdr = open(os.path.join(ROOT, "include", "uncr_hip.h")).read()
    ops = sorted(int(v) for v in re.findall(r"#define UNCR_EW_\w+ (\d+)", hdr))
    assert ops and set(ops) <= set(b.EW_TENSORS)
    for op in ops:
        label, nbytes, _, _ = b.kernel_model("uncr_ew", (op, 1024, 65536, 256, 0, 1))
        assert nbytes > 0 and 0.0 <= b.written_fraction("uncr_ew", (op, 1024, 65536, 256, 0, 1)) <= 0.5
    assert b.kernel_model("uncr_ew", (17, 1024, 65536, 256, 0, 1))[1] == 2.0 * 1024 * 65536
    # eval-mode MBConv tail (epi 10): pw2 reads its 256 input rows and the 128 skip rows, writes 128
    _, nb10, _, prod10 = b.kernel_model("uncr_pw_gemm", (0, 4, 256, 128, 65536, 2, 10, 0, 0, 256, 0))
    assert nb10 == 4.0 * 4 * 65536 * (256 + 128 + 128) and prod10 == 3
    assert abs(b.written_fraction("uncr_pw_gemm", (0, 4, 256, 128, 65536, 2, 10, 0, 0, 256, 0)) - 0.25) < 1e-9


def test_stream_ratio_never_reports_a_fraction_above_one():
    b = _bench()
    assert b.stream_ratio(5000.0, 6000.0) == {"frac_of_stream_roof": round(5000.0 / 6000.0, 4)}
    r = b.stream_ratio(6600.0, 6000.0)
    assert r["frac_of_stream_roof"] is None and r["infinity_cache_assisted"] is True and r["ratio_to_cold_stream_probe"] == 1.1


def test_power_sampler_parses_rocm_smi(monkeypatch):
    b = _bench()
    txt_max = "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\nGPU[1]\t\t: Max Graphics Package Power (W): 1400.0\n"
    txt = ("GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: sclk clock level: 5: (2026Mhz)\n"
           "GPU[1]\t\t: sclk clock level: 5: (1999Mhz)\n"
           "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1381.0\nGPU[1]\t\t: Current Socket Graphics Package Power (W): 900.0\n")
    monkeypatch.setattr(b.PowerSampler, "_smi", staticmethod(lambda *a: txt_max if "--showmaxpower" in a else txt))
    ps = b.PowerSampler(1, period_s=0.01)
    assert ps.cap == 1400.0
    ps.start()
    import time
    t0 = time.perf_counter()
    time.sleep(0.8)
    ps.stop()
    s = ps.summary(t0 - 0.6, time.perf_counter())
    assert s["samples"] > 3 and s["avg_power_w"] == 900.0 and s["avg_sclk_mhz"] == 1999 and s["power_cap_w"] == 1400.0
    empty = b.PowerSampler(0, period_s=0.01).summary(0.0, 1.0)
    assert empty["samples"] == 0 and "note" in empty


def test_traffic_file_follows_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py attaches PMC traffic only from a profiles/<tag>_traffic[_bf16].json measured on the present kernel sources: the newest
    file whose `_source_sha` matches wins; without a match the newest file is named (and the line then reports traffic null)."""
    import json
    b = _bench()
    from uncrtaints_amd import build
    sha = build.source_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    for tag, s_ in (("r01", "old"), ("r02", sha), ("r03", "other")):
        (prof / f"{tag}_traffic.json").write_text(json.dumps({"_source_sha": s_}))
        (prof / f"{tag}_traffic_bf16.json").write_text(json.dumps({"_source_sha": s_ if tag != "r02" else "stale"}))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    assert b.traffic_file() == "r02_traffic.json"
    assert b.traffic_file(True) == "r03_traffic_bf16.json"          # no bf16 file matches: the newest one is named
    (prof / "r04_traffic_bf16.json").write_text(json.dumps({"_source_sha": sha}))
    assert b.traffic_file(True) == "r04_traffic_bf16.json"
