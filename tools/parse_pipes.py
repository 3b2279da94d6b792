"""Summarise the SQ counter passes of tools/measure_pipes.sh per kernel.
usage: python tools/parse_pipes.py <out.json> <pass1 counter_collection.csv> [<pass2 ...>]
Ratios: the SQ_WAIT_* / SQ_ACTIVE_INST_* counters and SQ_WAVE_CYCLES count quad-cycles summed over waves, so their quotients say how
a resident wave spends its time; SQ_VALU_MFMA_BUSY_CYCLES counts cycles of a busy MFMA pipe summed over the 1024 SIMDs and
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel_cycles = GRBM_GUI_ACTIVE / 8, mfma_pipe_busy = MFMA_BUSY / (1024 * kernel_cycles)
and valu_issue_busy = (SQ_INSTS_VALU - SQ_INSTS_MFMA) * 4 / (1024 * kernel_cycles)."""
import csv, json, sys, collections

NAMES = [("dw_bwd_row_kernel<float", "dw_bwd"), ("dw_fwd_row_kernel<float", "dw_fwd"),
         ("pw_gemm_split_kernel<2, 3, 3, 1, float, ", "pw_gemm[128->256,pro3,epi3] (dz)"),
         ("pw_gemm_split_kernel<2, 1, 1, 1, float, ", "pw_gemm[128->256,pro1,epi1] (pw1 fwd)"),
         ("pw_gemm_split_kernel<1, 2, 1, 2, float, ", "pw_gemm[256->128,pro2,epi1] (pw2 fwd)"),
         ("pw_gemm_split_kernel<1, 3, 5, 2, float, ", "pw_gemm_dx[256->128]"),
         ("pw_wgrad_split_kernel<4, 2,", "pw_wgrad[256x128]"), ("pw_wgrad_split_kernel<2, 4,", "pw_wgrad[128x256]")]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
            c = r.get("Counter_Name") or r.get("Counter Name")
            v = float(r.get("Counter_Value") or r.get("Counter Value"))
            for sub, key in NAMES:
                if sub in name:
                    acc[key][c].append(v)
    res = {"_comment": __doc__}
    for key, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = {"counters_mean_per_launch": {c: round(v, 1) for c, v in sorted(m.items())}}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c, lab in (("SQ_ACTIVE_INST_ANY", "wave_issuing"), ("SQ_WAIT_ANY", "wave_parked_waitcnt_or_barrier"),
                           ("SQ_WAIT_INST_ANY", "wave_issue_stalled"), ("SQ_ACTIVE_INST_VALU", "wave_issuing_valu"),
                           ("SQ_WAIT_INST_LDS", "wave_issue_stalled_on_lds")):
                if c in m:
                    d[lab] = round(m[c] / wc, 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            d["kernel_cycles"] = round(cyc)
            d["mfma_pipe_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
            if "SQ_INSTS_VALU" in m:
                d["valu_issue_busy"] = round((m["SQ_INSTS_VALU"] - m.get("SQ_INSTS_MFMA", 0.0)) * 4.0 / (1024.0 * cyc), 4)
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_share"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        res[key] = d
        print(key, {k: v for k, v in d.items() if k != "counters_mean_per_launch"})
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
