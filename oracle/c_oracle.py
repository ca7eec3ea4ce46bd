"""ctypes binding of oracle/libllda_oracle.so (the C restatement)  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libllda_oracle.so")
    src = os.path.join(_HERE, "llda_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libllda_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libllda_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        P = ctypes.c_void_p
        L.llda_oracle_sweep.restype = ctypes.c_int
        L.llda_oracle_sweep.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                        P, P, P, P, P, P, P, P,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_uint64,
                                        ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int64, ctypes.c_int]
        L.llda_oracle_sweep_docs.restype = ctypes.c_int
        L.llda_oracle_sweep_docs.argtypes = [ctypes.c_int64, P, ctypes.c_int, ctypes.c_int64, P, P, P, P, P, P, P, P,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint32,
                                             ctypes.c_uint32, ctypes.c_int]
        L.llda_oracle_sweep_wm.restype = ctypes.c_int
        L.llda_oracle_sweep_wm.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, P, P, P, P, P, P, P,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_int64, ctypes.c_int]
        L.llda_oracle_pairwise_sum.restype = ctypes.c_double
        L.llda_oracle_pairwise_sum.argtypes = [P, ctypes.c_int64]
        L.llda_oracle_uniform.restype = ctypes.c_double
        L.llda_oracle_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                          ctypes.c_uint32, ctypes.c_uint32]
        L.llda_oracle_draw.restype = ctypes.c_int
        L.llda_oracle_draw.argtypes = [ctypes.c_int, P, ctypes.c_double]
        L.llda_oracle_philox.restype = None
        L.llda_oracle_philox.argtypes = [P, P, P]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pairwise_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().llda_oracle_pairwise_sum(_p(a), a.shape[0])


def uniform(seed, sweep, stream, doc, site):
    return lib().llda_oracle_uniform(seed, sweep, stream, doc, site)


def draw(prob, u):
    prob = np.ascontiguousarray(prob, dtype=np.float64)
    return lib().llda_oracle_draw(prob.shape[0], _p(prob), u)


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().llda_oracle_philox(_p(c), _p(k), _p(out))
    return out


class CState(object):
    """Reference-layout state on the host for the C oracle: n_d_k (D,K) int64, n_k_v (K,V) int64,
    n_zk (K,) int64, z int32[S], CSR corpus, labs uint8 (D,K)."""

    def __init__(self, doc_off, word, freq, z, labs, n_d_k, n_k_v, n_zk, V, alpha, beta):
        self.doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        self.word = np.ascontiguousarray(word, dtype=np.int32)
        self.freq = np.ascontiguousarray(freq, dtype=np.int32)
        self.z = np.array(z, dtype=np.int32)
        self.labs = np.ascontiguousarray(np.asarray(labs) != 0, dtype=np.uint8)
        self.n_d_k = np.array(n_d_k, dtype=np.int64, order="C")
        self.n_k_v = np.array(n_k_v, dtype=np.int64, order="C")
        self.n_zk = np.array(n_zk, dtype=np.int64)
        self.D, self.K = self.labs.shape
        self.V = int(V)
        self.alpha, self.beta = float(alpha), float(beta)

    def sweep(self, mode, seed, sweep, stream=0, doc_base=0, threads=1):
        rc = lib().llda_oracle_sweep(mode, self.D, self.K, self.V, _p(self.doc_off), _p(self.word),
                                     _p(self.freq), _p(self.z), _p(self.labs), _p(self.n_d_k),
                                     _p(self.n_k_v), _p(self.n_zk), self.alpha, self.beta,
                                     seed, sweep, stream, doc_base, threads)
        if rc != 0:
            raise RuntimeError("llda_oracle_sweep failed: %d" % rc)


class WMState(object):
    """Word-major int32 state for llda_oracle_sweep_wm (dense label mask): n_d_k (D,K) int32, n_kw (V,K) int32, n_k (K,) int32 -- the
    layout of the GPU kernels; the assignments after a sweep equal CState.sweep(1, ...)'s."""

    def __init__(self, doc_off, word, freq, z, n_d_k, n_k_v, n_zk, V, alpha, beta):
        self.doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        self.word = np.ascontiguousarray(word, dtype=np.int32)
        self.freq = np.ascontiguousarray(freq, dtype=np.int32)
        self.z = np.array(z, dtype=np.int32)
        self.n_d_k = np.array(n_d_k, dtype=np.int32, order="C")
        self.n_kw = np.ascontiguousarray(np.asarray(n_k_v).T, dtype=np.int32)
        self.n_k = np.array(n_zk, dtype=np.int32)
        self.D, self.K = self.n_d_k.shape
        self.V = int(V)
        self.alpha, self.beta = float(alpha), float(beta)

    def sweep(self, seed, sweep, stream=0, doc_base=0, threads=1):
        rc = lib().llda_oracle_sweep_wm(self.D, self.K, self.V, _p(self.doc_off), _p(self.word), _p(self.freq), _p(self.z),
                                        _p(self.n_d_k), _p(self.n_kw), _p(self.n_k), self.alpha, self.beta, seed, sweep, stream,
                                        doc_base, threads)
        if rc != 0:
            raise RuntimeError("llda_oracle_sweep_wm failed: %d" % rc)


def sweep_docs(doc_ids, doc_off, word, freq, z, labs, n_d_k, n_k_v, n_zk, V, alpha, beta, seed, sweep, stream=0,
               threads=1):
    """llda_oracle_sweep_docs: snapshot sweep (O3) of a SELECTION of documents of a larger corpus against the
    sweep-start counts n_k_v (K, V) int64 / n_zk (K) int64, which are left untouched.  doc_ids = global document
    ids (RNG key); doc_off / word / freq / z = CSR of the selected documents only; labs = (n_sel, K) 0/1 or None
    (dense); n_d_k = (n_sel, K) int64 rows.  Returns (z_new int32, n_d_k_new int64)."""
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int64)
    doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
    word = np.ascontiguousarray(word, dtype=np.int32)
    freq = np.ascontiguousarray(freq, dtype=np.int32)
    z = np.array(z, dtype=np.int32)
    n_d_k = np.array(n_d_k, dtype=np.int64, order="C")
    assert n_k_v.dtype == np.int64 and n_k_v.flags.c_contiguous and n_zk.dtype == np.int64
    K = n_k_v.shape[0]
    lab_p = None
    if labs is not None:
        labs = np.ascontiguousarray(np.asarray(labs) != 0, dtype=np.uint8)
        lab_p = _p(labs)
    rc = lib().llda_oracle_sweep_docs(doc_ids.shape[0], _p(doc_ids), K, int(V), _p(doc_off), _p(word), _p(freq), _p(z),
                                      lab_p, _p(n_d_k), _p(n_k_v), _p(n_zk), float(alpha), float(beta), int(seed),
                                      int(sweep), int(stream), int(threads))
    if rc != 0:
        raise RuntimeError("llda_oracle_sweep_docs failed: %d" % rc)
    return z, n_d_k
