"""Summarise one profiling run (gpurun_out/<dir>/, written by tools/profile.sh on the GPU box) into profiles/:

    profiles/<tag>_bench_default.json       the bench line of the default run (rooflines from in-run PMC passes, all extras)
    profiles/<tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the timed line alone (kernels >= 0.02 %)
    profiles/<tag>_kernel_stats_extras.csv  ... and of the same command with the extras
    profiles/<tag>_pmc_<pass>.csv           the counter passes bench.py ran on itself: one row per sweep-kernel dispatch and
                                            counter (summed over the counter's instances), passes fetch | write | sq | l2 | ta
    profiles/<tag>_summary.md               kernel table + per workload: counters per launch and the derived roofline numbers
    profiles/pmc_traffic.json               fabric bytes per site per workload (bench.py's labelled fallback at N > 1)

Fabric traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate --pmc passes; FETCH_SIZE tallies every L2 line fill (a 128-byte request) as 64 bytes (MI355X_MICROARCH.md, section
HBM, for wide reads; tools/gather_ubench.hip for 4-byte gathers).  The requests include Infinity-Cache hits.

usage: python profiles/summarize.py <tag> [<dir under gpurun_out, default prof_<tag>>]
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTRAS = (("synth1", "synth1"), ("hbm_bound", "synth2_hostile"), ("sparse_labels", "synth2_sparse"),
          ("sparse_labels_colocated", "synth2_sparse_hier"),
          ("wide_sparse_k2048", "synth_wide_sparse"), ("wide_k2048", "synth_wide"), ("abstracts", "abstracts"))


def main(tag, sub=None):
    src = os.path.join(ROOT, "gpurun_out", sub or ("prof_" + tag))
    prof = os.path.join(ROOT, "profiles")
    text = open(os.path.join(src, "bench_default.json")).read().strip().split("\n")[-1]
    line = json.loads(text)
    json.dump(line, open(os.path.join(prof, "%s_bench_default.json" % tag), "w"), indent=1)
    out = ["# rocprofv3 summary %s" % tag, "", "Command: `python bench.py` (default: the 1M-document corpus on one GPU, all extras).", ""]
    for fname, dest, title in (("stats_kernel_stats.csv", "%s_kernel_stats.csv",
                                "## kernel stats of the timed line (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 "
                                "--no-cpu --no-pmc --no-extras`): the sweep kernel's average is what `roofline.kernel_ms` must agree with"),
                               ("stats_extras_kernel_stats.csv", "%s_kernel_stats_extras.csv",
                                "## kernel stats with the extras (`... --no-cpu --no-pmc`): every kernel of the library; an instantiation "
                                "shared by two workloads shows their mixed average")):
        stats = os.path.join(src, fname)
        if not os.path.exists(stats):
            continue
        out += [title, "", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
        keep = []
        rows = list(csv.DictReader(open(stats)))
        for r in rows:
            if float(r["Percentage"]) >= 0.02:
                keep.append(r)
                name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                if name.startswith("llda_") or float(r["Percentage"]) >= 0.5:
                    out.append("| `%s` | %s | %.4f | %s |" % (name[:110], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
        out.append("")
        with open(os.path.join(prof, dest % tag), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
    counters = {}
    for tagp in ("fetch", "write", "sq", "l2", "ta"):
        f = os.path.join(src, "pmc", "pmc_%s.csv" % tagp)
        if not os.path.exists(f):
            continue
        shutil.copy(f, os.path.join(prof, "%s_pmc_%s.csv" % (tag, tagp)))
        for r in csv.DictReader(open(f)):
            counters.setdefault(r["Workload"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value_summed_over_instances"]))
    out += ["", "## sweep kernels: PMC counters per launch (in-run passes of bench.py, 3 launches per workload) and what follows from them", ""]
    traffic = {}

    def block(name, key, sites):
        r = key.get("roofline")
        if not r:
            return
        out.append("### %s -- `%s`" % (name, r.get("kernel", "?")))
        out.append("")
        out.append("| quantity | value |")
        out.append("|---|---|")
        out.append("| sites per launch | %d |" % sites)
        out.append("| kernel ms (HIP events, timed region) | %.4f |" % r["kernel_ms"])
        for c, v in sorted(counters.get(name, {}).items()):
            out.append("| %s (mean per launch) | %.6g |" % (c, sum(v) / len(v)))
        if r.get("traffic"):
            out.append("| fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) x 1024 | %.4g (%.1f B/site, %.2f x algorithmic) |" %
                       (r["traffic"], r["traffic"] / sites, r["traffic_over_algorithmic"]))
            out.append("| fabric GB/s / 8000 (`achieved`, `frac`) | %.0f / %.3f |" % (r["achieved"], r["frac"]))
            h = r.get("hbm_estimate")
            if h:
                out.append("| HBM estimate GB/s (lower .. upper) | %.0f .. %.0f |" % (h["lower_GBps"], h["upper_GBps"]))
            traffic[name] = r["traffic"] / sites
        out.append("| algorithmic bytes per launch (SURVEY 8d) | %.4g (%.0f GB/s over the kernel time) |" % (r["algorithmic_bytes_per_launch"], r["algorithmic_GBps"]))
        if r.get("l2"):
            out.append("| L2 hit rate; fabric read requests per site | %.3f; %.2f |" % (r["l2"]["hit_rate"], r["l2"]["fabric_read_requests_per_site"]))
        if "ta_busy_frac" in r:
            out.append("| TA busy; TCP pending-stall share | %.3f; %.3f |" % (r["ta_busy_frac"], r.get("tcp_pending_stall_frac", float("nan"))))
        v = r.get("valu_issue")
        if v:
            out.append("| VALU instructions per site | %.1f |" % v["valu_insts_per_site"])
            out.append("| VALU issue frac (x 4 cycles / 1024 SIMDs x 2.4 GHz) | %.3f |" % v["frac"])
            if "valu_busy_frac" in v:
                out.append("| VALU busy share of the profiled launch's cycles | %.3f (clock %.2f GHz) |" % (v["valu_busy_frac"], v.get("effective_clock_GHz", float("nan"))))
            if "wave_cycles_waiting_frac" in v:
                out.append("| wave cycles waiting on an instruction; waves per SIMD | %.3f; %.2f |" % (v["wave_cycles_waiting_frac"], v.get("waves_per_simd_avg", float("nan"))))
        im = r.get("issue_model")
        if im:
            out.append("| SALU instructions per site; issue model (4 x VALU + 2.5 x SALU cycles per SIMD / shader cycles) | %.1f; %.3f |" %
                       (im["salu_insts_per_site"], im["frac"]))
        out.append("| binding roof (headroom) | %s (%.2f) |" % (r.get("binding_roof", "-"), r.get("headroom", float("nan"))))
        out.append("")
    block("synth2", line, line["config"]["sites_per_sweep"])
    for k, nm in EXTRAS:
        e = line.get("extra", {}).get(k)
        if e:
            block(nm, e, e["sites_per_sweep"])
    out += ["## the harness steps (SURVEY 8f rows), timed in the same run", ""]
    for k in ("cascade", "pipeline_abstracts", "cascade_test"):
        e = line.get("extra", {}).get(k)
        if e:
            cb = e.get("cpu_baseline", {})
            out.append("* `%s`: %.4f s%s" % (k, e["value"], (" -- CPU port %.1f s (%s)" % (cb["value"], cb.get("sample", ""))) if cb else ""))
    if traffic:
        tpath = os.path.join(prof, "pmc_traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        t.update(traffic)
        t["_unit"] = ("fabric bytes per site of the sweep kernel: (2*FETCH_SIZE+WRITE_SIZE)*1024 / sites, rocprofv3 --pmc passes on ONE GPU "
                      "(profiles/%s_summary.md); bench.py uses it, labelled 'stored', where it cannot profile (N > 1)" % tag)
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
    for extra in os.listdir(src):
        if extra.startswith(("bench_", "gather_", "abl_")) and extra != "bench_default.json" and not extra.endswith(".err"):
            shutil.copy(os.path.join(src, extra), os.path.join(prof, "%s_%s" % (tag, extra)))
    # the int32 ablation of the timed line (tools/profile.sh: LLDA_BENCH_ROWS16=off)
    i32 = os.path.join(src, "int32")
    if os.path.exists(os.path.join(i32, "bench.json")):
        shutil.copy(os.path.join(i32, "bench.json"), os.path.join(prof, "%s_int32_bench.json" % tag))
        for tagp in ("fetch", "write", "sq", "l2", "ta"):
            f = os.path.join(i32, "pmc", "pmc_%s.csv" % tagp)
            if os.path.exists(f):
                shutil.copy(f, os.path.join(prof, "%s_int32_pmc_%s.csv" % (tag, tagp)))
        l32 = json.loads(open(os.path.join(i32, "bench.json")).read().strip().split("\n")[-1])
        r = l32.get("roofline", {})
        out += ["", "## the timed line with int32 rows (`LLDA_BENCH_ROWS16=off python bench.py --no-cpu --no-extras`)", "",
                "* %.0f M sites/s, %.2f ms per sweep, kernel `%s` %.2f ms; fabric %.2f TB/s = %.2f of 8 TB/s, binding roof: %s" %
                (l32["value"], l32["ms_per_step"], r.get("kernel", ""), r.get("kernel_ms", 0.0), r.get("achieved", 0.0) / 1e3,
                 r.get("frac", 0.0), r.get("binding_roof", ""))]
    open(os.path.join(prof, "%s_summary.md" % tag), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
