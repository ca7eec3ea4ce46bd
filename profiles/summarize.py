"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into profiles/:

    profiles/<tag>_bench_default.json       the bench line of the default run (roofline from in-run PMC passes)
    profiles/<tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the same command (kernels >= 0.05 %)
    profiles/<tag>_pmc_*.csv                the sweep-kernel rows of the raw counter CSVs of the in-run passes
    profiles/<tag>_summary.md               kernel table + counters per launch + the derived roofline numbers
    profiles/pmc_traffic.json               HBM bytes per site per workload (bench.py's fallback when it cannot profile)

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate --pmc passes, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, section HBM) -- the kernel reads n_kw rows as 16 B/lane global_load_dwordx4.  WRITE_SIZE is
uncalibrated on gfx950 (same section) and is taken as is.

usage: python profiles/summarize.py <tag>
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    prof = os.path.join(ROOT, "profiles")
    line = json.load(open(os.path.join(src, "bench_default.json")))
    json.dump(line, open(os.path.join(prof, "%s_bench_default.json" % tag), "w"), indent=1)
    out = ["# rocprofv3 summary %s" % tag, "", "Command: `python bench.py` (default: the 1M-document corpus on one GPU).", ""]
    stats = os.path.join(src, "stats_kernel_stats.csv")
    if os.path.exists(stats):
        out += ["## kernel stats (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu --no-pmc --no-extras`)", "",
                "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
        keep = []
        rows = list(csv.DictReader(open(stats)))
        for r in rows:
            if float(r["Percentage"]) >= 0.05:
                keep.append(r)
                out.append("| `%s` | %s | %.4f | %s |" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
        with open(os.path.join(prof, "%s_kernel_stats.csv" % tag), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
    # counters: the sweep-kernel rows of the in-run passes
    out += ["", "## PMC counters of llda_sweep_kernel, per launch (in-run passes of bench.py, 3 launches per workload)", ""]
    for tagp in ("fetch", "write", "sq"):
        f = os.path.join(src, "pmc", "pmc_%s_counter_collection.csv" % tagp)
        if not os.path.exists(f):
            continue
        rows = [r for r in csv.DictReader(open(f)) if "llda_sweep_kernel" in r["Kernel_Name"]]
        with open(os.path.join(prof, "%s_pmc_%s.csv" % (tag, tagp)), "w", newline="") as fh:
            cols = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                    "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
            w = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore")
            w.writeheader()
            for r in rows:
                r["Kernel_Name"] = r["Kernel_Name"][:80]
                w.writerow(r)
    traffic = {}

    def block(name, key, sites):
        r = key.get("roofline")
        if not r:
            return
        out.append("### %s" % name)
        out.append("")
        out.append("| quantity | value |")
        out.append("|---|---|")
        out.append("| kernel ms (HIP events, timed region) | %.4f |" % r["kernel_ms"])
        if r.get("traffic"):
            out.append("| HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) x 1024 | %.4g (%.1f B/site) |" % (r["traffic"], r["traffic"] / sites))
            out.append("| HBM GB/s achieved / 8000 | %.0f / frac %.3f |" % (r["achieved"], r["frac"]))
            traffic[name] = r["traffic"] / sites
        out.append("| algorithmic bytes per launch (SURVEY 8d) | %.4g (%.0f GB/s over the kernel time) |" % (r["algorithmic_bytes_per_launch"], r["algorithmic_GBps"]))
        v = r.get("valu_issue")
        if v:
            out.append("| VALU instructions per site | %.1f |" % v["valu_insts_per_site"])
            out.append("| VALU issue frac (x 4 cycles / 1024 SIMDs x 2.4 GHz) | %.3f |" % v["frac"])
            if "valu_busy_frac" in v:
                out.append("| VALU busy share of the profiled launch's cycles | %.3f (clock %.2f GHz) |" % (v["valu_busy_frac"], v.get("effective_clock_GHz", float("nan"))))
        out.append("| binding roof | %s |" % r.get("binding_roof", "-"))
        out.append("")
    block("synth2", line, line["config"]["sites_per_sweep"])
    for k, nm in (("synth1", "synth1"), ("hbm_bound", "synth2_hostile")):
        e = line.get("extra", {}).get(k)
        if e:
            block(nm, e, e["sites_per_sweep"])
    if traffic:
        tpath = os.path.join(prof, "pmc_traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        t.update(traffic)
        t["_unit"] = "HBM bytes per site of llda_sweep_kernel: (2*FETCH_SIZE+WRITE_SIZE)*1024 / sites, rocprofv3 --pmc passes (profiles/%s_summary.md)" % tag
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
    for extra in ("bench_synth2_sparse.json", "bench_cascade.json", "bench_cascade_one_by_one.json"):
        if os.path.exists(os.path.join(src, extra)):
            shutil.copy(os.path.join(src, extra), os.path.join(prof, "%s_%s" % (tag, extra)))
    open(os.path.join(prof, "%s_summary.md" % tag), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
