"""bench.py's line-building helpers on the CPU: the driver's end-of-round run must not trip over a missing counter pass, an
unknown workload or a kernel name rocprofv3 prints differently."""
import importlib.util
import json
import os

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_short_kernel_name(bench):
    f = bench.short_kernel_name
    assert f("void (anonymous namespace)::llda_sweep_kernel<32, 16, false, true, true, false>((anonymous namespace)::KParams)") == \
        "llda_sweep_kernel<32, 16, false, true, true, false>"
    assert f("void (anonymous namespace)::llda_sweep_sparse_kernel<8, (anonymous namespace)::KParams>((anonymous namespace)::KParams)") == \
        "llda_sweep_sparse_kernel<8, KParams>"
    assert f("llda_sweep_wide_kernel<true>(WParams)") == "llda_sweep_wide_kernel<true>"
    assert f("something_else") == "something_else"
    assert bench.is_sweep_kernel("x::llda_sweep_batch_kernel<8>") and not bench.is_sweep_kernel("llda_commit_log_kernel")


def test_algorithmic_bytes_is_the_survey_formula(bench):
    # per site 4*A + 32, per document 12*A + 16
    assert bench.algorithmic_bytes(3 * 10 ** 8, 10 ** 6, 512) == 3 * 10 ** 8 * 2080 + 10 ** 6 * 6160 == 630160000000
    assert bench.algorithmic_bytes(10, 2, 8) == 10 * 64 + 2 * 112


def test_roofline_block_with_all_some_and_no_counters(bench):
    full = {"FETCH_SIZE": 1.8e8, "WRITE_SIZE": 2.9e7, "SQ_INSTS_VALU": 2.6e10, "SQ_ACTIVE_INST_VALU": 2.7e10, "SQ_WAVE_CYCLES": 1.0e11,
            "SQ_BUSY_CYCLES": 4.3e9, "SQ_WAIT_INST_ANY": 2.0e10, "GRBM_GUI_ACTIVE": 1.07e9, "TCC_HIT_sum": 3.5e9, "TCC_MISS_sum": 3.1e9,
            "TCC_EA0_RDREQ_sum": 2.8e9, "TCC_EA0_WRREQ_sum": 6.0e8, "TA_BUSY_avr": 9.6e7, "TCP_PENDING_STALL_CYCLES_sum": 2.0e10,
            "sq_pass_kernel_ns": 5.7e7, "kernel_name": "llda_sweep_kernel<32, 16, false, true, true, false>"}
    r = bench.roofline_json(55.7, 3 * 10 ** 8, 10 ** 6, 512.0, full, "test", stored_key="synth2", shared_bytes=205 * 10 ** 6)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["traffic_kind"] == "measured"
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic"] == (2 * 1.8e8 + 2.9e7) * 1024 and abs(r["traffic_over_algorithmic"] - r["traffic"] / 630160000000) < 1e-12
    assert r["binding_roof"].startswith("fabric") and 0.5 < r["l2"]["hit_rate"] < 0.6 and 0.7 < r["valu_issue"]["frac"] < 0.85
    assert r["hbm_estimate"]["lower_GBps"] < r["hbm_estimate"]["upper_GBps"] <= 6300.0
    assert r["kernel"].startswith("llda_sweep_kernel<32")
    json.dumps(r)
    # counts beyond the Infinity Cache: the roof is called hbm
    assert bench.roofline_json(12.4, 37500000, 125000, 512.0, dict(full, FETCH_SIZE=3.86e7, WRITE_SIZE=3.8e6), "t",
                               shared_bytes=10 ** 9)["roof_fractions"].keys() >= {"hbm"}
    # a failed pass: only some counters
    part = {"FETCH_SIZE": 1.8e8, "WRITE_SIZE": 2.9e7}
    r = bench.roofline_json(55.7, 3 * 10 ** 8, 10 ** 6, 512.0, part, "test", shared_bytes=205 * 10 ** 6)
    assert r["traffic"] is not None and "valu_issue" not in r and "l2" not in r
    json.dumps(r)
    # no counters at all (N > 1): the stored per-site figure, labelled as such -- or nothing
    r = bench.roofline_json(7.0, 37500000, 125000, 512.0, None, "not collected", stored_key="synth2", shared_bytes=205 * 10 ** 6)
    assert r["traffic_kind"] in ("stored", None)
    if r["traffic_kind"] == "stored":
        assert "STORED" in r["traffic_source"] and r["traffic"] > 0
    r = bench.roofline_json(7.0, 1000, 10, 8.0, None, "not collected", stored_key="no_such_workload")
    assert r["traffic"] is None and r["frac"] is None and r["achieved"] is None
    json.dumps(r)


def test_checksum_verdict_against_the_stored_table(bench):
    table = json.load(open(bench.CHECKSUM_FILE))
    t = table["synth2:1000000"]
    assert len(t["n_k"]) == len(t["n_kw"]) >= 45            # covers the driver's --steps 20 --warmup 5 and the default 40 + 3
    ok, note = bench.checksum_verdict("synth2", 1000000, 25, {"n_k": t["n_k"][24], "n_kw": t["n_kw"][24]})
    assert ok is True and "25 sweeps" in note
    ok, _ = bench.checksum_verdict("synth2", 1000000, 25, {"n_k": t["n_k"][24] + 1, "n_kw": t["n_kw"][24]})
    assert ok is False
    assert bench.checksum_verdict("synth2", 1000000, 10 ** 6, {"n_k": 0, "n_kw": 0})[0] is None     # beyond the table
    assert bench.checksum_verdict("synth2", 200000, 3, {"n_k": 0, "n_kw": 0})[0] is None            # another corpus
    assert bench.checksum_verdict("abstracts", 4171, 3, {"n_k": 0, "n_kw": 0})[0] is None


def test_workload_table_is_consistent(bench):
    for name in bench.PMC_WORKLOADS:
        assert name in bench.WORKLOADS
    for name, (docs, n, v, k, zs, block, desc) in bench.WORKLOADS.items():
        if name == "abstracts":
            continue
        assert docs % block == 0 and n <= v and k >= 1 and desc


def _canned_detail(bench, n_gpus=1):
    """a full record shaped like main() builds it, with prose as long as round 3's in every block"""
    full = {"FETCH_SIZE": 1.03e8, "WRITE_SIZE": 3.06e7, "SQ_INSTS_VALU": 2.6e10, "SQ_ACTIVE_INST_VALU": 2.7e10, "SQ_WAVE_CYCLES": 1.0e11,
            "SQ_BUSY_CYCLES": 4.3e9, "SQ_WAIT_INST_ANY": 2.0e10, "GRBM_GUI_ACTIVE": 8.7e8, "TCC_HIT_sum": 3.5e9, "TCC_MISS_sum": 3.1e9,
            "TCC_EA0_RDREQ_sum": 2.8e9, "TCC_EA0_WRREQ_sum": 6.0e8, "TA_BUSY_avr": 9.6e7, "TCP_PENDING_STALL_CYCLES_sum": 2.0e10,
            "SQ_INSTS_SALU": 7.0e9, "sq_pass_kernel_ns": 4.7e7,
            "kernel_name": "llda_sweep_kernel<32, 16, false, true, true, false, true, true>"}
    roof = bench.roofline_json(46.93, 3 * 10 ** 8, 10 ** 6, 512.0, full, "in-run rocprofv3 passes " * 20, stored_key="synth2",
                               shared_bytes=205 * 10 ** 6)
    essay = "a paragraph of explanation that belongs in the detail file, not in the line the driver parses. " * 9
    cpu = {"cpu_model": "AMD EPYC 9575F 64-Core Processor", "physical_cores": 128, "value": 0.0819498709, "unit": "Mtokens/s", "cores": 1,
           "kind": "port", "sample": "first 3000 docs (900000 sites) of the same workload, " + essay, "c_port_1thread_Mtokens_s": 0.33124,
           "c_port_allcores_Mtokens_s": 1.4989, "host_cores": 256, "port_vs_reference": essay,
           "c_port_reference_layout": {"best": {"threads": 64, "docs": 100000, "sites": 30000000, "seconds": 2.9, "value": 10.3},
                                       "legs": [{"threads": t, "docs": 100000 * t // 64, "sites": 1, "seconds": 3.0, "value": v}
                                                for t, v in ((64, 10.3), (32, 6.1), (16, 3.2))], "physical_cores": 128, "logical_cores": 256,
                                       "layout": essay,
                                       "word_major": {"best": {"threads": 64, "docs": 100000, "sites": 30000000, "seconds": 1.1, "value": 27.5},
                                                      "legs": [{"threads": t, "docs": 100000 * t // 64, "sites": 1, "seconds": 1.0, "value": v}
                                                               for t, v in ((64, 27.5), (32, 15.0), (16, 8.0))], "layout": essay}}}
    detail = {"metric": "million tokens resampled/sec (Gibbs sweep)", "value": 6245.746077706536, "unit": "Mtokens/s", "n_gpus": n_gpus,
              "steps": 20, "warmup": 5, "ms_per_step": 48.03269237454515, "higher_is_better": True, "scaling": "strong",
              "vs_baseline": None, "dtype": "f64-exact (fp32 tier 0 + fp64 tiers: the drawn topic is the fp64 pipeline's)", "data": "synthetic",
              "config": {"workload": bench.WORKLOADS["synth2"][6], "docs_total": 1000000, "docs_per_gpu": 1000000 // n_gpus,
                         "sites_per_doc": 300, "K": 512, "V": 100000, "alpha": 0.1, "beta": 0.01, "label_mask": "dense",
                         "kernel": "dense", "n_kw_rows": essay, "n_kw_rows_short": "16-bit image + int32 hot rows", "build_info": 0,
                         "abi": 17, "library": "lda_thesis_amd/libllda_gibbs.so", "sites_per_sweep": 300000000,
                         "timed_seconds": 0.96, "exchange": essay, "semantics": essay, "draw": essay,
                         "state_checksum_n_k": 39403620123456, "state_checksum_n_kw": 8123456789012345678, "sweeps_behind_checksum": 25},
              "draw_tiers": {"sites": 7500000000, "fp32_tier_unsure": 12345678, "exact_tier": 1234},
              "roofline": roof, "cpu_baseline": cpu, "speedup_vs_cpu_port": 76213.4}
    if n_gpus > 1:
        detail["checksum_matches_n1"], detail["checksum_note"] = True, essay
        detail["exchange_ms"] = {"exposed_ms_per_sweep": 1.234567, "collectives_per_sweep": 1, "bytes_per_collective": 102500000,
                                 "overlap_ranges": 1, "allreduce_alone_ms": 0.98765}
        detail["overlap_probe"] = {"overlap_ranges": 2, "steps": 5, "ms_per_step": 7.5, "sweeps": 7, "checksum_matches_n1": True,
                                   "exchange_ms": {"exposed_ms_per_sweep": 0.5, "collectives_per_sweep": 2}}
        detail.pop("cpu_baseline")
        detail["roofline"] = bench.roofline_json(6.1, 37500000, 125000, 512.0, None, "not collected (N > 1)", stored_key="synth2",
                                                 shared_bytes=205 * 10 ** 6)
    else:
        extra = {}
        for key in ("synth1", "hbm_bound", "sparse_labels", "sparse_labels_colocated", "abstracts", "wide_k2048", "wide_sparse_k2048"):
            extra[key] = {"workload": essay, "value": 11028.723864360003, "unit": "Mtokens/s", "steps": 100, "warmup": 5,
                          "ms_per_step": 3.4002120699733496, "timed_seconds": 0.34, "docs": 125000, "sites_per_sweep": 37500000, "K": 512,
                          "V": 100000, "kernel": "sparse", "n_kw_rows": essay, "kernel_ms": 3.31, "note": essay,
                          "roofline": bench.roofline_json(3.31, 37500000, 125000, 8.0, full, essay, shared_bytes=205 * 10 ** 6)}
        extra["abstracts"]["cpu_baseline"] = dict(cpu, c_port_reference_layout=None)
        extra["abstracts"].update(unit="Msites/s", Mtokens_s=11028.7 * 1.24, tokens_per_site=1.24)
        extra["abstracts"]["speedup_vs_cpu_port"] = 19876.5
        extra["cascade"] = {"workload": essay, "value": 0.085, "median_s": 0.09, "max_s": 0.18, "unit": "s", "higher_is_better": False,
                            "cold_first_call_s": 1.9, "warm_calls_s": [0.098, 0.085, 0.181, 0.09, 0.09, 0.09], "reference_cpu_note": essay}
        for key in ("pipeline_abstracts", "cascade_test"):
            extra[key] = {"workload": essay, "value": 0.0706, "unit": "s", "higher_is_better": False,
                          "stages_s": {"train_s": 0.05, "test_s": 0.02, "metrics_s": 0.001}, "speedup_vs_cpu_port": 6358.2,
                          "cpu_baseline": {"kind": "port", "cores": 1, "unit": "s", "value": 449.0,
                                           "sample": "the walk of the first 3 held-out documents (1.2 s), scaled by sites x node visits"}}
        detail["extra"] = extra
    return detail


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_the_printed_line_is_small_parseable_and_numbers_only(bench, n_gpus):
    """BENCH_r03.json had parsed: null because the one line had grown to 40 KB: the line main() prints is compact_line() of the
    full record -- under 8 KB, json round trip, the contract's keys, roofline and cpu_baseline inside, no prose."""
    detail = _canned_detail(bench, n_gpus)
    assert len(json.dumps(detail)) > (20000 if n_gpus == 1 else 8000)                   # the full record IS an essay; it goes to --detail-out
    line = bench.compact_line(detail, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 8000 and "\n" not in text
    back = json.loads(text)
    assert back == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in back, k
    assert back["n_gpus"] == n_gpus and back["value"] == pytest.approx(6245.746, rel=1e-5) and back["vs_baseline"] is None
    assert back["config"]["workload"].startswith("synthetic 1M docs") and back["config"]["build_info"] == 0
    assert back["config"]["detail"] == "gpurun_out/bench_detail.json"
    r = back["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    longest = max((len(v) for v in _strings(back)), default=0)
    assert longest <= 200, longest                               # no paragraph survives
    if n_gpus == 1:
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["traffic"] > 0
        assert r["algorithmic_bytes"] == 630160000000 and r["traffic_over_algorithmic"] == pytest.approx(r["traffic"] / 630160000000, rel=1e-4)
        assert 0 < r["valu_busy_frac"] <= 1.0 and 0 < r["l2_hit_rate"] < 1 and r["binding_roof"] in ("valu_issue", "fabric", "hbm")
        c = back["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] == 1 and c["cpu_model"].startswith("AMD EPYC") and c["value"] > 0
        assert c["c_port_reference_layout"] == {"value": 10.3, "threads": 64, "docs": 100000, "seconds": 2.9,
                                                "legs": [[64, 10.3], [32, 6.1], [16, 3.2]]}
        assert c["c_port_word_major"] == {"value": 27.5, "threads": 64, "docs": 100000, "seconds": 1.1,
                                          "legs": [[64, 27.5], [32, 15.0], [16, 8.0]]}
        # what the numbers ARE: the SURVEY 8(d) rate over the peak is flagged as no fraction, the HBM fraction is bracketed, the
        # cache-hostile extra is the kernel's HBM credential, the dtype names both tiers, a real corpus reports sites AND tokens
        assert r["algorithmic_frac"] == pytest.approx(r["algorithmic_GBps"] / 8000.0, rel=1e-4) and "not a fraction" in r["algorithmic_frac_note"]
        assert len(r["hbm_frac_bounds"]) == 2 and 0 < r["hbm_frac_bounds"][0] < r["hbm_frac_bounds"][1] <= r["frac"] + 1e-9
        assert r["hbm_credential"]["workload"] == "hbm_bound" and r["hbm_credential"]["frac"] > 0
        assert back["dtype"].startswith("f64-exact") and "fp32 tier 0" in back["dtype"]
        assert back["extra"]["abstracts"]["unit"] == "Msites/s" and back["extra"]["abstracts"]["tokens_per_site"] == 1.24
        assert back["extra"]["abstracts"]["Mtokens_s"] == pytest.approx(11028.7 * 1.24, rel=1e-4)
        assert set(back["extra"]) == set(detail["extra"])
        for key, e in back["extra"].items():
            assert "value" in e and "unit" in e, key
        assert back["extra"]["sparse_labels"]["frac"] > 0 and back["extra"]["sparse_labels"]["binding_roof"]
        assert back["extra"]["cascade"]["median_s"] == 0.09 and back["extra"]["cascade"]["max_s"] == 0.18
        assert back["extra"]["cascade_test"]["cpu"]["scaled_from_sample"] is True
        assert back["extra"]["abstracts"]["speedup_vs_cpu_port"] == pytest.approx(19876.5, rel=1e-4)
    else:
        assert back["checksum_matches_n1"] is True and "cpu_baseline" in back and back["cpu_baseline"] is None
        assert back["exchange_ms"]["exposed_ms_per_sweep"] == pytest.approx(1.2346, rel=1e-4)
        assert back["exchange_ms"]["bytes_per_collective"] == 102500000
        assert back["overlap_probe"] == {"overlap_ranges": 2, "steps": 5, "ms_per_step": 7.5, "checksum_matches_n1": True,
                                         "exposed_ms_per_sweep": 0.5}


def _strings(x):
    if isinstance(x, str):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _strings(v)
    elif isinstance(x, list):
        for v in x:
            yield from _strings(v)


def test_the_line_sheds_its_extras_rather_than_outgrow_the_limit(bench):
    detail = _canned_detail(bench, 1)
    for i in range(400):
        detail["extra"]["more_%d" % i] = dict(detail["extra"]["synth1"])
    line = bench.compact_line(detail, None)
    assert len(json.dumps(line)) < bench.LINE_LIMIT and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_host_cpu_reads_proc_cpuinfo(bench):
    model, logical, phys = bench.host_cpu()
    assert isinstance(model, str) and model and 1 <= phys <= logical
