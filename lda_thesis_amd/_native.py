"""ctypes binding of the HIP library (include/llda_gibbs.h).

The product path has NO CPU fallback: if ``libllda_gibbs.so`` is missing this module raises, and
every device entry point raises on a non-zero return code.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLDA_GIBBS_LIB") or os.path.join(_HERE, "libllda_gibbs.so")   # (override: ablation builds, tools/)

MAX_K = 7688            # LLDA_MAX_K: every K up to here has <= 64 pairwise leaves
MAX_KP = 8192
MAX_LEAVES = 8          # narrow layouts
MAX_WIDE_LEAVES = 64
MAX_ROUNDS = 4
ABI_VERSION = 21

_c_i32, _c_i64, _c_u32, _c_u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
_c_p, _c_d = ctypes.c_void_p, ctypes.c_double


class LldaLayout(ctypes.Structure):
    """struct llda_layout (include/llda_gibbs.h)."""
    _fields_ = [("K", _c_i32), ("n_leaves", _c_i32), ("G", _c_i32), ("T", _c_i32), ("KP", _c_i32),
                ("tail", _c_i32), ("tail_row", _c_i32), ("n_rounds", _c_i32), ("wide", _c_i32), ("tiers", _c_i32),
                ("leaf_start", _c_i32 * MAX_WIDE_LEAVES), ("leaf_len", _c_i32 * MAX_WIDE_LEAVES),
                ("rounds", (_c_i32 * MAX_LEAVES) * MAX_ROUNDS),
                ("comb_dst", _c_i32 * MAX_WIDE_LEAVES), ("comb_src", _c_i32 * MAX_WIDE_LEAVES),
                ("topic_pos", _c_i32 * MAX_KP), ("pos_topic", _c_i32 * MAX_KP),
                ("pos_lane", _c_i32 * MAX_KP), ("pos_slot", _c_i32 * MAX_KP)]


class LldaFoldinArgs(ctypes.Structure):
    """struct llda_foldin_args (include/llda_gibbs.h)."""
    _fields_ = [("doc_off", _c_p), ("word", _c_p), ("init_idx", _c_p), ("freq", _c_p), ("ph", _c_p),
                ("init_rows", _c_p), ("slot_valid", _c_p), ("z", _c_p), ("n_dk", _c_p), ("th", _c_p),
                ("status", _c_p), ("D", _c_i64), ("doc_base", _c_i64), ("K", _c_i32), ("iters", _c_i32),
                ("thinning", _c_i32), ("beta_fallback", _c_i32), ("avg_mode", _c_i32), ("exact_only", _c_i32),
                ("alpha", _c_d), ("beta", _c_d), ("c_init", _c_d), ("c_loop", _c_d), ("seed", _c_u64),
                ("stream_id", _c_u32), ("reserved2", _c_u32), ("doc_ids", _c_p), ("n_sites", _c_i64),
                ("ph_base", _c_p), ("doc_stream", _c_p)]


class LldaSweepArgs(ctypes.Structure):
    """struct llda_sweep_args (include/llda_gibbs.h)."""
    _fields_ = [("doc_off", _c_p), ("doc_order", _c_p), ("word", _c_p), ("freq", _c_p), ("z", _c_p),
                ("lab_mask", _c_p), ("n_dk", _c_p), ("n_kw", _c_p), ("n_kw_delta", _c_p),
                ("n_k", _c_p), ("n_k_delta", _c_p), ("status", _c_p),
                ("D", _c_i64), ("V", _c_i64), ("K", _c_i32), ("docs_per_group", _c_i32),
                ("dense_mask", _c_i32), ("debug_margin", _c_i32), ("alpha", _c_d), ("beta", _c_d), ("seed", _c_u64), ("sweep", _c_u32),
                ("stream_id", _c_u32), ("doc_base", _c_i64),
                ("live_off", _c_p), ("live_pos", _c_p), ("scratch", _c_p), ("scratch_bytes", _c_i64),
                ("live_max", _c_i32), ("max_doc_tokens", _c_i32), ("csc_pos", _c_p), ("commit_log", _c_p),
                ("n_sites", _c_i64), ("site_rec", _c_p), ("n_kw16", _c_p), ("site_row", _c_p),
                ("n_kw_img", _c_p), ("img_bits", _c_i32), ("reserved_img", _c_i32), ("row16", _c_p), ("img_col", _c_p)]


class LldaBatchArgs(ctypes.Structure):
    """struct llda_batch_args (include/llda_gibbs.h)."""
    _fields_ = [("inst_off", _c_p), ("order", _c_p), ("n_inst", _c_i64), ("word", _c_p), ("freq", _c_p), ("z", _c_p),
                ("inst_prob", _c_p), ("inst_doc", _c_p), ("live_off", _c_p), ("live_pos", _c_p), ("ndk_off", _c_p),
                ("n_dk", _c_p), ("kw_off", _c_p), ("nk_off", _c_p), ("kp", _c_p), ("prob_stream", _c_p), ("k", _c_p), ("counts", _c_p),
                ("delta", _c_p),
                ("status", _c_p), ("V", _c_i64), ("lanes", _c_i32), ("debug_margin", _c_i32), ("alpha", _c_d),
                ("beta", _c_d), ("seed", _c_u64), ("sweep", _c_u32), ("reserved", _c_u32)]


EXPORTS = ("llda_abi_version", "llda_build_info", "llda_strerror", "llda_last_hip_error", "llda_struct_size", "llda_layout_init",
           "llda_sweep_scratch_bytes", "llda_rows16_ok", "llda_quad_ok", "llda_pack_rows16", "llda_pack_rows16_all", "llda_pack_image", "llda_pack_image_cols",

           "llda_sweep", "llda_sweep_batch", "llda_commit_log", "llda_apply_rows", "llda_apply_delta", "llda_count_init", "llda_loglik", "llda_foldin",
           "llda_readout_phi", "llda_readout_theta", "llda_selftest_div")

_LIB = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load the HIP library; raise if it is not built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise NativeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (or make -C lda_thesis_amd/csrc). There is no CPU fallback."
                          % LIB_PATH)
    # torch ships its own libamdhip64 / libhsa-runtime64: import it first so this library binds to
    # the SAME HIP runtime instance as the tensors and streams it is handed (loading ours first pulls
    # in /opt/rocm's copy and kernel launches then fail with hipErrorNoDevice).
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.llda_abi_version.restype = ctypes.c_int
    L.llda_abi_version.argtypes = []
    L.llda_build_info.restype = ctypes.c_int
    L.llda_build_info.argtypes = []
    L.llda_strerror.restype = ctypes.c_char_p
    L.llda_strerror.argtypes = [ctypes.c_int]
    L.llda_last_hip_error.restype = ctypes.c_int
    L.llda_last_hip_error.argtypes = []
    L.llda_layout_init.restype = ctypes.c_int
    L.llda_layout_init.argtypes = [_c_i32, ctypes.POINTER(LldaLayout)]
    L.llda_sweep.restype = ctypes.c_int
    L.llda_sweep.argtypes = [ctypes.POINTER(LldaSweepArgs), _c_p]
    L.llda_sweep_scratch_bytes.restype = _c_i64
    L.llda_sweep_scratch_bytes.argtypes = [_c_i32, _c_i64]
    L.llda_rows16_ok.restype = ctypes.c_int
    L.llda_rows16_ok.argtypes = [_c_i32]
    L.llda_quad_ok.restype = ctypes.c_int
    L.llda_quad_ok.argtypes = [_c_i32]
    L.llda_pack_rows16.restype = ctypes.c_int
    L.llda_pack_rows16.argtypes = [_c_p, _c_p, _c_i64, _c_i32, _c_p, _c_p, _c_p]
    L.llda_pack_rows16_all.restype = ctypes.c_int
    L.llda_pack_rows16_all.argtypes = [_c_p, _c_i64, _c_i32, _c_p, _c_p, _c_p]
    L.llda_pack_image_cols.restype = ctypes.c_int
    L.llda_pack_image_cols.argtypes = [_c_p, _c_i64, _c_i32, _c_i32, _c_p, _c_p, _c_p]
    L.llda_pack_image.restype = ctypes.c_int
    L.llda_pack_image.argtypes = [_c_p, _c_i64, _c_i32, _c_p, _c_p]
    L.llda_sweep_batch.restype = ctypes.c_int
    L.llda_sweep_batch.argtypes = [ctypes.POINTER(LldaBatchArgs), _c_p]
    L.llda_commit_log.restype = ctypes.c_int
    L.llda_commit_log.argtypes = [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_i32, _c_p, _c_p, _c_p, _c_p, _c_p]
    L.llda_apply_rows.restype = ctypes.c_int
    L.llda_apply_rows.argtypes = [_c_p, _c_p, _c_i64, _c_i32, _c_p, _c_p]
    L.llda_apply_delta.restype = ctypes.c_int
    L.llda_apply_delta.argtypes = [_c_p, _c_p, _c_i64, _c_p]
    L.llda_count_init.restype = ctypes.c_int
    L.llda_count_init.argtypes = [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i32, _c_p, _c_p, _c_p, _c_p]
    L.llda_loglik.restype = ctypes.c_int
    L.llda_loglik.argtypes = [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i32, _c_d, _c_d,
                              _c_p, _c_p]
    L.llda_foldin.restype = ctypes.c_int
    L.llda_foldin.argtypes = [ctypes.POINTER(LldaFoldinArgs), _c_p]
    L.llda_readout_phi.restype = ctypes.c_int
    L.llda_readout_phi.argtypes = [_c_p, _c_p, _c_p, _c_i64, _c_i32, _c_d, _c_i32, _c_d, _c_d, _c_p, _c_p, _c_p]
    L.llda_readout_theta.restype = ctypes.c_int
    L.llda_readout_theta.argtypes = [_c_p, _c_p, _c_i64, _c_i32, _c_d, _c_i32, _c_d, _c_d, _c_p, _c_p]
    L.llda_selftest_div.restype = ctypes.c_int
    L.llda_selftest_div.argtypes = [_c_u64, _c_i64, _c_p, _c_p]
    if L.llda_abi_version() != ABI_VERSION:
        raise NativeError("libllda_gibbs.so ABI %d != binding ABI %d" % (L.llda_abi_version(), ABI_VERSION))
    L.llda_struct_size.restype = ctypes.c_int
    L.llda_struct_size.argtypes = [ctypes.c_int]
    for which, struct in enumerate((LldaLayout, LldaSweepArgs, LldaBatchArgs, LldaFoldinArgs)):
        if L.llda_struct_size(which) != ctypes.sizeof(struct):
            raise NativeError("%s: binding has %d bytes, the library %d" % (struct.__name__, ctypes.sizeof(struct),
                                                                           L.llda_struct_size(which)))
    _LIB = L
    return L


BUILD_SWITCHES = ("LLDA_MARGIN0", "LLDA_WAVES", "LLDA_MARGIN0_WIDE", "ABL_NOLOAD", "ABL_NOCOMMIT", "ABL_WIDE_NOROW",
                  "ABL_WIDE_NOADDLOAD", "ABL_NOFMA", "ABL_EXTRA_LDS_BYTES", "QUAD_PROFILE", "LLDA_BUDGET_MARKS", "LLDA_QUAD_PRIO")      # bit i of llda_build_info()


def build_info():
    """llda_build_info: (bits, names of the compile-time switches the loaded library was built with); (0, []) = production."""
    bits = int(lib().llda_build_info())
    return bits, [n for i, n in enumerate(BUILD_SWITCHES) if bits >> i & 1]


def require_device():
    """Raise unless a HIP device is visible (the product has no CPU path)."""
    import torch
    if not torch.cuda.is_available():
        raise NativeError("no HIP device visible: the sampler has no CPU fallback")


def check(rc, what):
    if rc != 0:
        L = lib()
        raise NativeError("%s failed: %s (code %d, hipError %d)"
                          % (what, L.llda_strerror(rc).decode(), rc, L.llda_last_hip_error()))


def layout_init(K):
    """llda_layout_init -> dict of numpy arrays / ints (host only, needs no device)."""
    out = LldaLayout()
    check(lib().llda_layout_init(int(K), ctypes.byref(out)), "llda_layout_init(K=%d)" % K)
    return dict(K=out.K, n_leaves=out.n_leaves, G=out.G, T=out.T, KP=out.KP, tail=out.tail,
                tail_row=out.tail_row, n_rounds=out.n_rounds, wide=out.wide, tiers=out.tiers,
                comb=[(out.comb_dst[i], out.comb_src[i]) for i in range(out.n_leaves - 1)],
                leaf_start=np.array(out.leaf_start[:out.n_leaves]),
                leaf_len=np.array(out.leaf_len[:out.n_leaves]),
                rounds=np.array([list(r) for r in out.rounds])[:out.n_rounds],
                topic_pos=np.array(out.topic_pos[:out.K], dtype=np.int32),
                pos_topic=np.array(out.pos_topic[:out.KP], dtype=np.int32),
                pos_lane=np.array(out.pos_lane[:out.KP], dtype=np.int32),
                pos_slot=np.array(out.pos_slot[:out.KP], dtype=np.int32))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _launch(ref, fn, what, *args):
    """call the enqueue-only entry point ``fn(*args, stream)`` on the device that holds the tensor ``ref``: that
    device is made current for the call and the stream is torch's current stream OF THAT DEVICE (not of whichever
    device happens to be current), so the kernels are ordered with the torch ops on the tensors they touch."""
    import torch
    dev = ref.device
    with torch.cuda.device(dev):
        check(fn(*args, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), what)


def sweep(*, doc_off, doc_order, word, freq, z, lab_mask, n_dk, n_kw, n_kw_delta, n_k, n_k_delta,
          status, D, V, K, alpha, beta, seed, sweep, stream_id=0, doc_base=0, docs_per_group=0,
          dense_mask=False, debug_margin=0, live_off=None, live_pos=None, live_max=0, csc_pos=None, commit_log=None,
          n_sites=None, site_rec=None, max_doc_tokens=0, scratch=None, n_kw16=None, site_row=None, n_kw_img=None, row16=None,
          img_col=None):
    """llda_sweep on the current torch stream.  All array arguments are torch CUDA tensors.  n_sites = the sites
    the D documents span (default: all of ``word``); scratch = a uint8 tensor of sweep_scratch_bytes(K, D) bytes (wide
    layouts) or None; n_kw16 = the 16-bit image written by pack_rows16 (then csc_pos carries the row flags in bit 31 and site_row the row starts) or None;
    n_kw_img = the saturating uint8 / int16 image written by pack_image (sparse label sets) or None; row16 = the per-word flags
    written by pack_rows16_all (with n_kw16, without site_row: the four-documents-per-wavefront kernel of K = 512) or None."""
    img_bits = 0 if n_kw_img is None else 8 * n_kw_img.element_size()
    a = LldaSweepArgs(_ptr(doc_off), _ptr(doc_order), _ptr(word), _ptr(freq), _ptr(z), _ptr(lab_mask),
                      _ptr(n_dk), _ptr(n_kw), _ptr(n_kw_delta), _ptr(n_k), _ptr(n_k_delta), _ptr(status),
                      int(D), int(V), int(K), int(docs_per_group), 1 if dense_mask else 0, int(debug_margin),
                      float(alpha), float(beta),
                      int(seed) & 0xFFFFFFFFFFFFFFFF, int(sweep) & 0xFFFFFFFF,
                      int(stream_id) & 0xFFFFFFFF, int(doc_base), _ptr(live_off), _ptr(live_pos), _ptr(scratch),
                      0 if scratch is None else int(scratch.numel() * scratch.element_size()), int(live_max), int(max_doc_tokens),
                      _ptr(csc_pos), _ptr(commit_log), int(word.numel() if n_sites is None else n_sites), _ptr(site_rec),
                      _ptr(n_kw16), _ptr(site_row), _ptr(n_kw_img), img_bits, 0, _ptr(row16), _ptr(img_col))
    _launch(z, lib().llda_sweep, "llda_sweep", ctypes.byref(a))


def sweep_scratch_bytes(K, D):
    """llda_sweep_scratch_bytes: bytes of work space that let llda_sweep take its fastest kernel (0: none needed)."""
    return int(lib().llda_sweep_scratch_bytes(int(K), int(D)))


def rows16_ok(K):
    """llda_rows16_ok: can llda_sweep read 16-bit rows for K topics?"""
    return bool(lib().llda_rows16_ok(int(K)))


def quad_ok(K):
    """llda_quad_ok: does llda_sweep take the per-sweep row flags (llda_sweep_args.row16: K / 32 lanes x 32 slots per document)?"""
    return bool(lib().llda_quad_ok(int(K)))


def pack_rows16(n_kw, row16, K, n_kw16, status):
    """llda_pack_rows16 on the current torch stream: the 16-bit image of the rows of n_kw flagged in row16 (uint8 [V])."""
    _launch(n_kw, lib().llda_pack_rows16, "llda_pack_rows16", _ptr(n_kw), _ptr(row16), int(row16.numel()), int(K),
            _ptr(n_kw16), _ptr(status))


def pack_rows16_all(n_kw, K, n_kw16, row16):
    """llda_pack_rows16_all on the current torch stream: the 16-bit image of EVERY row of n_kw and, in row16 (uint8 [V]), whether
    all counts of the row fit 16 bits."""
    _launch(n_kw, lib().llda_pack_rows16_all, "llda_pack_rows16_all", _ptr(n_kw), int(row16.numel()), int(K), _ptr(n_kw16), _ptr(row16))


def pack_image_cols(n_kw, K, col_src, img):
    """llda_pack_image_cols: the saturating image with its columns in the order col_src (int32 [KP] on the device)."""
    _launch(n_kw, lib().llda_pack_image_cols, "llda_pack_image_cols", _ptr(n_kw), int(n_kw.shape[0]), int(K), 8 * img.element_size(),
            _ptr(col_src), _ptr(img))


def pack_image(n_kw, img):
    """llda_pack_image on the current torch stream: img (uint8 or int16, as many elements as n_kw) = n_kw saturated at 255 / 65535."""
    _launch(n_kw, lib().llda_pack_image, "llda_pack_image", _ptr(n_kw), int(n_kw.numel()), 8 * img.element_size(), _ptr(img))


def sweep_batch(*, inst_off, order, word, freq, z, inst_prob, inst_doc, live_off, live_pos, ndk_off, n_dk, kw_off,
                nk_off, kp, prob_stream, k, counts, delta, status, V, lanes, alpha, beta, seed, sweep, debug_margin=0):
    """llda_sweep_batch on the current torch stream: one sweep of the instances in ``order`` (each with at most
    ``lanes`` allowed topics) of an ensemble of small problems."""
    a = LldaBatchArgs(_ptr(inst_off), _ptr(order), int(order.numel()), _ptr(word), _ptr(freq), _ptr(z),
                      _ptr(inst_prob), _ptr(inst_doc), _ptr(live_off), _ptr(live_pos), _ptr(ndk_off), _ptr(n_dk),
                      _ptr(kw_off), _ptr(nk_off), _ptr(kp), _ptr(prob_stream), _ptr(k), _ptr(counts), _ptr(delta), _ptr(status), int(V),
                      int(lanes), int(debug_margin), float(alpha), float(beta), int(seed) & 0xFFFFFFFFFFFFFFFF,
                      int(sweep) & 0xFFFFFFFF, 0)
    _launch(z, lib().llda_sweep_batch, "llda_sweep_batch", ctypes.byref(a))


def commit_log(item_begin, item_len, item_word, commit_log, freq_csc, K, target, n_k=None, n_k_delta=None,
               row_off=None):
    """llda_commit_log: fold the word-major commit log of a sweep into ``target`` (n_kw, its delta buffer, or --
    with ``row_off`` -- the exchange rows, int16 pairs where the offset is negative)."""
    _launch(target, lib().llda_commit_log, "llda_commit_log", _ptr(item_begin), _ptr(item_len), _ptr(item_word),
            int(item_len.numel()), _ptr(commit_log), _ptr(freq_csc), int(K), _ptr(row_off), _ptr(target), _ptr(n_k),
            _ptr(n_k_delta))


def apply_rows(row_off, rows, K, counts):
    """llda_apply_rows: counts[r] += row r of the exchange rows (pairs decoded), rows cleared."""
    _launch(rows, lib().llda_apply_rows, "llda_apply_rows", _ptr(row_off), _ptr(rows), int(row_off.numel()), int(K),
            _ptr(counts))


def apply_delta(counts, delta):
    _launch(counts, lib().llda_apply_delta, "llda_apply_delta", _ptr(counts), _ptr(delta), counts.numel())


def count_init(doc_off, word, freq, z, D, K, n_dk, n_kw, n_k):
    _launch(n_kw, lib().llda_count_init, "llda_count_init", _ptr(doc_off), _ptr(word), _ptr(freq), _ptr(z), int(D), int(K),
            _ptr(n_dk), _ptr(n_kw), _ptr(n_k))


def loglik(doc_off, word, lab_mask, n_dk, n_kw, n_k, D, V, K, alpha, beta, out_doc):
    _launch(out_doc, lib().llda_loglik, "llda_loglik", _ptr(doc_off), _ptr(word), _ptr(lab_mask), _ptr(n_dk), _ptr(n_kw),
            _ptr(n_k), int(D), int(V), int(K), float(alpha), float(beta), _ptr(out_doc))


READOUT_NEGATIVE, READOUT_NAN, READOUT_NO_LOAD = 1, 2, 4


def readout_phi(n_kw, n_k, den, V, K, beta, out, flags=None, keep=None, share=None):
    """llda_readout_phi: out (K, V) float64 = phi of the counts, or keep*out + share*phi when the two
    coefficients are given."""
    mode = 0 if keep is None else 1
    _launch(out, lib().llda_readout_phi, "llda_readout_phi", _ptr(n_kw), _ptr(n_k), _ptr(den), int(V), int(K), float(beta),
            mode, float(keep or 0.0), float(share or 0.0), _ptr(out), _ptr(flags))


def readout_theta(n_dk, lab_mask, D, K, alpha, out, keep=None, share=None):
    """llda_readout_theta: out (D, K) float64 = theta of the counts, or its running mean (as readout_phi)."""
    mode = 0 if keep is None else 1
    _launch(out, lib().llda_readout_theta, "llda_readout_theta", _ptr(n_dk), _ptr(lab_mask), int(D), int(K), float(alpha),
            mode, float(keep or 0.0), float(share or 0.0), _ptr(out))


def selftest_div(n, seed=1):
    """llda_selftest_div: number of (a, b) pairs (out of >= n) where the kernel's reciprocal-based
    division differs from the hardware IEEE division.  Must be 0."""
    import torch
    bad = torch.zeros((1,), dtype=torch.int64, device="cuda")
    _launch(bad, lib().llda_selftest_div, "llda_selftest_div", int(seed), int(n), _ptr(bad))
    return int(bad.item())


def foldin(*, doc_off, word, init_idx, freq, ph, init_rows, slot_valid, z, n_dk, th, status, D, K, iters, thinning,
           alpha, beta, c_init, c_loop, seed, stream_id, doc_base=0, beta_fallback=False, avg_mode=0, doc_ids=None,
           ph_base=None, doc_stream=None, exact_only=False):
    a = LldaFoldinArgs(_ptr(doc_off), _ptr(word), _ptr(init_idx), _ptr(freq), _ptr(ph), _ptr(init_rows),
                       _ptr(slot_valid), _ptr(z), _ptr(n_dk), _ptr(th), _ptr(status), int(D), int(doc_base), int(K),
                       int(iters), int(thinning), 1 if beta_fallback else 0, int(avg_mode), 1 if exact_only else 0, float(alpha),
                       float(beta), float(c_init), float(c_loop), int(seed) & 0xFFFFFFFFFFFFFFFF,
                       int(stream_id) & 0xFFFFFFFF, 0, _ptr(doc_ids), int(word.numel()), _ptr(ph_base), _ptr(doc_stream))
    _launch(z, lib().llda_foldin, "llda_foldin", ctypes.byref(a))
