"""Summarise rocprofv3 CSV output (gpurun_out/prof_<tag>/) into profiles/<tag>_<workload>.md and update
profiles/pmc_traffic.json (HBM bytes per sweep-kernel launch, read by bench.py).

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate --pmc passes, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, section HBM) -- the kernel reads n_kw rows as 16 B/lane global_load_dwordx4.
WRITE_SIZE is uncalibrated on gfx950 (same section) and is taken as is.

usage: python profiles/summarize.py <tag> <workload> <docs_per_gpu>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, workload, docs):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    lines = ["# rocprofv3 summary %s / %s (docs per GPU %s)" % (tag, workload, docs), ""]
    stats = os.path.join(src, "stats_%s_kernel_stats.csv" % workload)
    lines += ["## kernel stats (rocprofv3 --kernel-trace --stats)", "", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        if float(r["Percentage"]) < 0.05:
            continue
        lines.append("| `%s` | %s | %.4f | %s |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
    lines += ["", "## PMC counters of llda_sweep_kernel (mean per launch; separate passes)", "", "| counter | mean |", "|---|---|"]
    means = {}
    for k in ("fetch", "write", "sq", "sq2"):
        f = os.path.join(src, "%s_%s_counter_collection.csv" % (k, workload))
        if not os.path.exists(f):
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "llda_sweep" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            means[c] = sum(v) / len(v)
            lines.append("| %s | %.6g |" % (c, means[c]))
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        traffic = (2.0 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024.0
        lines += ["", "HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.4g bytes" % traffic]
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        t["%s:%s" % (workload, docs)] = traffic
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
    out = os.path.join(ROOT, "profiles", "%s_%s.md" % (tag, workload))
    open(out, "w").write("\n".join(lines) + "\n")
    # keep the raw stats csv too
    dst = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, workload))
    open(dst, "w").write(open(stats).read())
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
