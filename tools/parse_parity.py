"""Parse the `[parity]` lines of `pytest tests -m gpu -s -v` into a tracked summary (profiles/rNN_parity.json):
    python tools/parse_parity.py gpurun_out/r04_pytest_gpu_sv.log profiles/r04_parity.json
Per test: number of parity lines, the largest fp32-reference distance; for every close_grad / close_vs_truth line whose distance to
the fp32 reference is >= 1e-4 (i.e. that passed only on the fp64 clause): e_ref, e_got (HIP vs fp64 truth), e_cpu (CPU fp32 vs fp64
truth) and the ratio e_got / e_cpu.  Totals: worst e_got, worst ratio, how many lines needed the second clause."""
import json
import re
import sys

TEST = re.compile(r"^(tests/\S+::\S+)")
P3 = re.compile(r"\[parity\] (.+?): vs fp32 ref ([0-9.eE+-]+); vs fp64 truth: hip ([0-9.eE+-]+), cpu-fp32 ([0-9.eE+-]+)")
P1 = re.compile(r"\[parity\] (.+?): rel_err=([0-9.eE+-]+)")
PG = re.compile(r"\[parity\] grad (\w+) (.+?): ([0-9.eE+-]+)")


def main(log, out):
    tests, cur = {}, None
    for line in open(log, errors="replace"):
        m = TEST.match(line)
        if m:
            cur = m.group(1)
            line = line[m.end():]
        if cur is None or "[parity]" not in line:
            continue
        t = tests.setdefault(cur, {"lines": 0, "max_e_ref": 0.0, "second_clause": []})
        m = P3.search(line)
        if m:
            name, e_ref, e_got, e_cpu = m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4))
            t["lines"] += 1
            t["max_e_ref"] = max(t["max_e_ref"], e_ref)
            t["max_e_got"] = max(t.get("max_e_got", 0.0), e_got)
            if e_ref >= 1e-4:
                t["second_clause"].append({"name": name, "e_ref": e_ref, "e_got": e_got, "e_cpu": e_cpu,
                                           "ratio": round(e_got / e_cpu, 2) if e_cpu > 0 else None})
            continue
        m = PG.search(line) or P1.search(line)
        if m:
            t["lines"] += 1
            t["max_e_ref"] = max(t["max_e_ref"], float(m.groups()[-1]))
    second = [dict(test=k, **s) for k, v in tests.items() for s in v["second_clause"]]
    three = [v for v in tests.values() if "max_e_got" in v]
    summ = {
        "source": log, "tests_with_parity_lines": len(tests), "parity_lines": sum(v["lines"] for v in tests.values()),
        "lines_passing_only_on_the_fp64_clause": len(second),
        "worst_e_got_of_those": max((s["e_got"] for s in second), default=0.0),
        # of those, the lines further than 1e-4 from the fp64 evaluation too: they pass on NOISE x the CPU evaluations' own distance
        # (the printed cpu-fp32 figure is the largest distance among the fp32 evaluations that were needed to admit the line)
        "lines_beyond_1e-4_of_fp64": sum(1 for s in second if s["e_got"] > 1e-4),
        "worst_ratio_e_got_over_e_cpu_of_those_beyond_1e-4": max((s["ratio"] or 0.0 for s in second if s["e_got"] > 1e-4), default=0.0),
        "worst_ratio_e_got_over_e_cpu_of_those": max((s["ratio"] or 0.0 for s in second), default=0.0),
        "worst_e_got_overall": max((v["max_e_got"] for v in three), default=0.0),
        "second_clause_lines": sorted(second, key=lambda s: -(s["ratio"] or 0.0)),
        "per_test": {k: {"lines": v["lines"], "max_e_ref": v["max_e_ref"], **({"max_e_got": v["max_e_got"]} if "max_e_got" in v else {}),
                         "second_clause": len(v["second_clause"])} for k, v in sorted(tests.items())},
    }
    json.dump(summ, open(out, "w"), indent=1)
    print({k: v for k, v in summ.items() if k not in ("second_clause_lines", "per_test")})
    for s in summ["second_clause_lines"][:15]:
        print(s)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
