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
