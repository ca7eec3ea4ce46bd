"""Shared helpers for the parity tests (oracle side)."""
import numpy as np

import llda_oracle as orc


def oracle_state(g, prefix="init_", phantom_counts=True):
    """numpy-oracle State from a golden fixture, counts taken verbatim from the fixture (so the
    SubLDA phantom columns are whatever the reference produced)."""
    doc_off = g["doc_off"]
    D = int(g["D"])
    docs = [g["word"][doc_off[d]:doc_off[d + 1]].tolist() for d in range(D)]
    freqs = [g["freq"][doc_off[d]:doc_off[d + 1]].tolist() for d in range(D)]
    z = g[prefix + "z"]
    z_dn = [z[doc_off[d]:doc_off[d + 1]] for d in range(D)]
    st = orc.State(docs, freqs, g["labs"].astype(np.float64), int(g["V"]), float(g["alpha"]),
                   float(g["beta"]), z_dn)
    st.n_k_v = g[prefix + "n_k_v"].astype(np.int64)
    st.n_d_k = g[prefix + "n_d_k"].astype(np.int64)
    st.n_zk = g[prefix + "n_zk"].astype(np.int64)
    return st


def c_state(co, g, prefix="init_"):
    return co.CState(g["doc_off"], g["word"], g["freq"], g[prefix + "z"], g["labs"],
                     g[prefix + "n_d_k"], g[prefix + "n_k_v"], g[prefix + "n_zk"],
                     int(g["V"]), float(g["alpha"]), float(g["beta"]))


def assert_state_equal(g, key, n_k_v, n_d_k, n_zk, z, what=""):
    np.testing.assert_array_equal(np.asarray(z).astype(np.int64), g[key + "_z"].astype(np.int64), err_msg=what + " z")
    np.testing.assert_array_equal(np.asarray(n_zk).astype(np.int64), g[key + "_n_zk"].astype(np.int64), err_msg=what + " n_zk")
    np.testing.assert_array_equal(np.asarray(n_d_k).astype(np.int64), g[key + "_n_d_k"].astype(np.int64), err_msg=what + " n_d_k")
    np.testing.assert_array_equal(np.asarray(n_k_v).astype(np.int64), g[key + "_n_k_v"].astype(np.int64), err_msg=what + " n_k_v")
