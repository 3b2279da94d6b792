"""Turn the two rocprofv3 PMC passes of `tools/bench_kernels.py traffic` (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_traffic.json.
usage: python tools/parse_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Both counters are reported in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)."""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncrtaints_amd.build import source_sha

BF16 = "--bf16" in sys.argv      # kernel names and element size of the bf16 activation storage (tools/bench_kernels.py traffic_bf16)
if BF16:
    sys.argv.remove("--bf16")
EB = 2 if BF16 else 4
if BF16:
    KEYS = [("dw_bwd_row_kernel<unsigned short", "dw_bwd[N4,C256,256x256]", EB * 4 * 256 * 65536 * 4),
            ("dw_fwd_row_kernel<unsigned short>", "dw_fwd[N4,C256,256x256]", EB * 4 * 256 * 65536 * 2),
            ("pw_gemm_split_kernel<2, 3, 3, 1, unsigned short, false>", "pw_gemm[128->256,pro3,epi3,N4,P65536]", EB * 4 * 65536 * (2 * 128 + 2 * 256)),
            ("pw_gemm_split_kernel<2, 1, 1, 1, unsigned short, false>", "pw_gemm[128->256,pro1,epi1,N4,P65536]", EB * 4 * 65536 * (128 + 256)),
            ("pw_gemm_split_kernel<1, 2, 1, 2, unsigned short, false>", "pw_gemm[256->128,pro2,epi1,N4,P65536]", EB * 4 * 65536 * (256 + 128)),
            ("pw_wgrad_a16_kernel<4, 2", "pw_wgrad[256x128,N4,P65536]", EB * 4 * 65536 * (2 * 256 + 128)),
            ("pw_wgrad_a16_kernel<2, 4", "pw_wgrad[128x256,N4,P65536]", EB * 4 * 65536 * (2 * 128 + 256)),
            ("pw_gemm_split_kernel<1, 3, 5, 2, unsigned short, false>", "pw_gemm_dx[256->128,N4,P65536]", EB * 4 * 65536 * (2 * 256 + 4 * 128))]
else:
    KEYS = [("dw_bwd_row_kernel<float, true>", "dw_bwd[N4,C256,256x256]", 4 * 4 * 256 * 65536 * 4),
            ("dw_fwd_row_kernel<float>", "dw_fwd[N4,C256,256x256]", 4 * 4 * 256 * 65536 * 2),
            ("pw_gemm_split_kernel<2, 3, 3, 1, float, true>", "pw_gemm[128->256,pro3,epi3,N4,P65536]", 4 * 4 * 65536 * (2 * 128 + 2 * 256)),
            ("pw_gemm_split_kernel<2, 1, 1, 1, float, true>", "pw_gemm[128->256,pro1,epi1,N4,P65536]", 4 * 4 * 65536 * (128 + 256)),
            ("pw_gemm_split_kernel<1, 2, 1, 2, float, true>", "pw_gemm[256->128,pro2,epi1,N4,P65536]", 4 * 4 * 65536 * (256 + 128)),
            ("pw_wgrad_split_kernel<4, 2, 3, 1, false>", "pw_wgrad[256x128,N4,P65536]", 4 * 4 * 65536 * (2 * 256 + 128)),
            ("pw_wgrad_split_kernel<2, 4, 3, 2, true>", "pw_wgrad[128x256,N4,P65536],fp16x2", 4 * 4 * 65536 * (2 * 128 + 256)),
            ("pw_gemm_split_kernel<1, 3, 5, 2, float, true>", "pw_gemm_dx[256->128,N4,P65536]", 4 * 4 * 65536 * (2 * 256 + 4 * 128))]


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        if (r.get("Counter_Name") or r.get("Counter Name")) != counter:
            continue
        acc[name].append(float(r.get("Counter_Value") or r.get("Counter Value")))
    return acc


def main():
    f, w, out = sys.argv[1:4]
    fe, wr = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    res = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes: two separate runs (--kernel-trace --pmc FETCH_SIZE, "
                       "--kernel-trace --pmc WRITE_SIZE) of `python tools/bench_kernels.py traffic` (tools/measure_traffic.sh), which "
                       "launches the kernels in isolation at the default bench shapes (N=4 frames, P=65536 px), 3 launches each, mean. "
                       "Both counters are reported in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at "
                       "64 B), WRITE_SIZE is used as reported. hbm_bytes = (2*fetch_kb_raw + write_kb_raw)*1024."}
    res["_storage"] = "bf16" if BF16 else "fp32"
    res["_source_sha"] = source_sha()      # bench.py attaches these numbers only while the kernel sources still hash to this
    for sub, key, alg in KEYS:
        fk = [v for n, vs in fe.items() if sub in n for v in vs]
        wk = [v for n, vs in wr.items() if sub in n for v in vs]
        if not fk or not wk:
            print("missing", sub, file=sys.stderr)
            continue
        fm, wm = sum(fk) / len(fk), sum(wk) / len(wk)
        res[key] = {"fetch_kb_raw": round(fm, 1), "write_kb_raw": round(wm, 1), "hbm_bytes": int((2 * fm + wm) * 1024),
                    "algorithmic_bytes": alg, "launches_averaged": len(fk)}
        print(key, res[key], "ratio", round(res[key]["hbm_bytes"] / alg, 3))
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
