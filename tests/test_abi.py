"""The C-ABI library loads without a GPU and exports every symbol include/llda_gibbs.h declares.
Host-only entry points are exercised; device entry points are only checked for argument validation
(they return before touching HIP)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "llda_gibbs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(llda_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("llda_sweep", "llda_apply_delta", "llda_count_init", "llda_loglik", "llda_layout_init",
              "llda_abi_version", "llda_strerror", "llda_last_hip_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from lda_thesis_amd import _native
    L = _native.lib()
    for s in declared_symbols():
        assert hasattr(L, s), "libllda_gibbs.so does not export %s" % s
    assert set(_native.EXPORTS) <= set(declared_symbols())
    assert L.llda_abi_version() == _native.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from lda_thesis_amd import _native
    monkeypatch.setattr(_native, "_LIB", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError):
        _native.lib()


def test_sampler_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lda_thesis_amd import _native
    from lda_thesis_amd.sampler import GibbsSampler
    with pytest.raises(_native.NativeError):
        GibbsSampler(np.array([0, 1]), np.array([0]), np.array([1]), np.array([0]), 2, 3, 0.1, 0.01)


@pytest.mark.parametrize("K", [1, 5, 8, 12, 20, 64, 100, 128, 129, 130, 200, 255, 256, 392, 512, 513, 777, 968, 1024,
                               969, 1023, 1031, 1500, 2047, 2100, 3000, 4096, 5000, 7688])
def test_layout_init_matches_python_layout(K):
    from lda_thesis_amd import _native
    from lda_thesis_amd.layout import GroupLayout
    a, b = _native.layout_init(K), GroupLayout(K)
    for k in ("G", "T", "KP", "tail", "tail_row", "n_rounds"):
        assert a[k] == getattr(b, k), k
    assert a["n_leaves"] == b.m and bool(a["wide"]) == b.wide and a["tiers"] == b.NT
    assert a["comb"] == list(b.comb)
    np.testing.assert_array_equal(a["topic_pos"], b.topic_pos)
    np.testing.assert_array_equal(a["pos_topic"], b.pos_topic)
    np.testing.assert_array_equal(a["pos_lane"], b.pos_lane)
    np.testing.assert_array_equal(a["pos_slot"], b.pos_slot)
    for r in range(a["n_rounds"]):
        np.testing.assert_array_equal(a["rounds"][r], b.rounds[r])


@pytest.mark.parametrize("K", [0, -3, 7689, 100000])
def test_layout_init_rejects_bad_k(K):
    from lda_thesis_amd import _native
    out = _native.LldaLayout()
    assert _native.lib().llda_layout_init(K, ctypes.byref(out)) == -1        # LLDA_E_BAD_K


def test_which_k_take_the_16_bit_rows():
    """host-only predicates: llda_rows16_ok (static flags: 16 slots per lane, 32 or 64 lanes, no padded slot) and llda_quad_ok (per-sweep
    flags, KP / 32 lanes x 32 slots per document: 8, 16 or 32 lanes, each lane group a leaf)"""
    from lda_thesis_amd import _native
    assert [K for K in (64, 100, 128, 200, 256, 392, 512, 777, 1024, 1031, 2048) if _native.rows16_ok(K)] == [512, 1024]
    # (K < KP is fine as long as every lane group holds a leaf of numpy's pairwise sum: 250 = 120 + 64 + 66 and 500 = five leaves are out)
    assert [K for K in (64, 96, 97, 100, 128, 129, 192, 200, 248, 250, 256, 257, 392, 400, 480, 500, 512, 777, 1024, 1031, 2048)
            if _native.quad_ok(K)] == [97, 100, 128, 200, 248, 256, 392, 400, 480, 512]
    assert not _native.quad_ok(0) and not _native.quad_ok(10 ** 6)


def test_device_entry_points_validate_arguments():
    from lda_thesis_amd import _native
    L = _native.lib()
    assert L.llda_sweep(None, None) == -2
    a = _native.LldaSweepArgs()
    a.D, a.V, a.K = 3, 10, 8
    assert L.llda_sweep(ctypes.byref(a), None) == -2                          # NULL pointers
    a.K = 0
    assert L.llda_sweep(ctypes.byref(a), None) == -1                          # bad K
    a.D, a.K = 0, 8
    assert L.llda_sweep(ctypes.byref(a), None) == 0                           # empty shard: nothing to do
    assert L.llda_apply_delta(None, None, 4, None) == -2
    assert L.llda_count_init(None, None, None, None, 1, 8, None, None, None, None) == -2
    assert L.llda_count_init(None, None, None, None, 0, 8, None, None, None, None) == 0
    assert L.llda_loglik(None, None, None, None, None, None, 1, 1, 8, 0.1, 0.1, None, None) == -2
    assert L.llda_foldin(None, None) == -2
    assert b"argument" in L.llda_strerror(-2)


def test_batched_sweep_validates_arguments():
    """llda_sweep_batch (ABI 9): NULL pointers, a lane count that is not 8/16/32/64 and priors the sparse arithmetic
    does not cover are refused before anything touches HIP; an empty launch is a no-op."""
    from lda_thesis_amd import _native
    L = _native.lib()
    assert L.llda_sweep_batch(None, None) == -2
    a = _native.LldaBatchArgs()
    a.n_inst, a.V, a.lanes = 0, 10, 8
    assert L.llda_sweep_batch(ctypes.byref(a), None) == 0                     # nothing to sample
    a.n_inst = 4
    assert L.llda_sweep_batch(ctypes.byref(a), None) == -2                    # NULL pointers
    a.V = 0
    assert L.llda_sweep_batch(ctypes.byref(a), None) == -2


def test_ensemble_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lda_thesis_amd import _native
    from lda_thesis_amd.ensemble import Ensemble
    plan = dict(K=2, docs=np.array([0]), allowed=np.array([[0, 1]]), n_allowed=np.array([2]))
    with pytest.raises(_native.NativeError):
        Ensemble([plan], [np.array([0])], np.array([0, 1]), np.array([0], dtype=np.int32), np.array([1], dtype=np.int32),
                 3, 0.1, 0.01, 1)


def test_integration_md_stub_struct_definitions_match_the_library():
    """the ctypes stub printed in INTEGRATION.md declares llda_layout / llda_sweep_args itself and checks them against
    llda_struct_size when it is imported: executing its definitions needs no GPU (the sweep through it runs in
    tests/test_gpu_dropin.py)."""
    import os
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# llda_gpu.py.*?```", text, flags=re.S).group(0)
    code = code[len("```python\n"):-3].replace('ctypes.CDLL("lda_thesis_amd/libllda_gibbs.so")',
                                               'ctypes.CDLL(%r)' % os.path.join(ROOT, "lda_thesis_amd", "libllda_gibbs.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)          # runs the stub's own assert on the struct sizes
    assert ns["lib"].llda_abi_version() == _native_abi()
    names = [n for n, _ in ns["SweepArgs"]._fields_]
    assert "scratch" in names and "resume" not in names        # ABI 14


def _native_abi():
    from lda_thesis_amd import _native
    return _native.ABI_VERSION


def test_build_info_reports_every_ablation_switch(tmp_path):
    """llda_build_info(): 0 for the production library (the one in the tree: bench.py prints no line otherwise); every compile-time
    switch of tools/ (-DABL_*, -DLLDA_MARGIN0 / _WAVES overrides) sets its LLDA_BUILD_* bit -- checked on csrc/build_info.hpp, which
    is all of llda_build_info(), with the host compiler."""
    import subprocess
    from lda_thesis_amd import _native
    assert _native.build_info() == (0, [])
    src = tmp_path / "bi.cpp"
    src.write_text('#include "build_info.hpp"\n#include "llda_gibbs.h"\n#include <stdio.h>\n'
                   'int main() { printf("%d\\n", (int)(LLDA_BUILD_INFO_BITS)); return 0; }\n')
    inc = ["-I", os.path.join(ROOT, "lda_thesis_amd", "csrc"), "-I", os.path.join(ROOT, "include")]

    def bits(*defs):
        exe = str(tmp_path / "bi")
        subprocess.check_call(["g++", "-o", exe, str(src)] + inc + list(defs))
        return int(subprocess.check_output([exe]).decode())
    assert bits() == 0
    want = {"-DLLDA_MARGIN0=0x1p-16f": 0x001, "-DLLDA_WAVES=2": 0x002, "-DLLDA_MARGIN0_WIDE=0.1f": 0x004, "-DABL_NOLOAD": 0x008,
            "-DABL_NOCOMMIT": 0x010, "-DABL_WIDE_NOROW": 0x020, "-DABL_WIDE_NOADDLOAD": 0x040, "-DABL_NOFMA": 0x080,
            "-DABL_EXTRA_LDS_BYTES=20000": 0x100, "-DQUAD_PROFILE": 0x200, "-DLLDA_BUDGET_MARKS": 0x400, "-DLLDA_QUAD_PRIO=202": 0x800}
    for d, b in want.items():
        assert bits(d) == b, d
    assert bits(*want) == 0xfff
    assert [n for i, n in enumerate(_native.BUILD_SWITCHES)] == [d[2:].split("=")[0] for d in want]
