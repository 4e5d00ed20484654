"""ctypes binding of the TEST-ONLY CPU oracle (oracle/_build/libpgm_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libpgm_oracle.so")
PGMSA_ORACLE = os.path.join(ROOT, "oracle", "_build", "pgmsa_oracle")


def _lib():
    if not os.path.exists(_SO) or not os.path.exists(PGMSA_ORACLE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return C.CDLL(_SO)


_l = _lib()
from prographmsa_amd import pgm_align_out, pgm_graph, pgm_model, pgm_scores  # noqa: E402  (struct layouts only)

_l.pgmo_align_graphs.restype = C.c_int
_l.pgmo_align_graphs.argtypes = [C.POINTER(pgm_graph), C.POINTER(pgm_graph), C.POINTER(pgm_model), C.POINTER(pgm_scores),
                                 C.POINTER(pgm_align_out), C.POINTER(C.POINTER(C.c_float))]
_l.pgmo_nw_pairs_batch.restype = C.c_int
_l.pgmo_csprofile_create.restype = C.c_int


def align_graphs(job, want_matrices=False):
    """Oracle alignGraphs for one prographmsa_amd.jobs.Job. Returns result dict (+ M,X,Y,W,S as [y,x] arrays)."""
    g1, g2 = job.g1.c(), job.g2.c()
    m = pgm_model()
    m.M = job.M.ctypes.data_as(C.POINTER(C.c_double))
    m.pi = job.pi.ctypes.data_as(C.POINTER(C.c_double))
    sc = pgm_scores()
    C.memmove(C.byref(sc), job.scores.ctypes.data, 40)
    n1, n2 = job.g1.n, job.g2.n
    map1, map2 = np.zeros(n1 + n2, np.uint32), np.zeros(n1 + n2, np.uint32)
    out = pgm_align_out()
    out.map1 = map1.ctypes.data_as(C.POINTER(C.c_uint32))
    out.map2 = map2.ctypes.data_as(C.POINTER(C.c_uint32))
    dbg = None
    mats = None
    if want_matrices:
        mats = [np.zeros(n1 * n2, np.float32) for _ in range(5)]
        dbg = (C.POINTER(C.c_float) * 5)(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in mats])
    rc = _l.pgmo_align_graphs(C.byref(g1), C.byref(g2), C.byref(m), C.byref(sc), C.byref(out), dbg)
    res = dict(score=out.score, n_tr_indels=out.n_tr_indels, status=rc, map1=map1[:out.len].copy(), map2=map2[:out.len].copy())
    if want_matrices:
        res["mats"] = [a.reshape(n2, n1).T for a in mats]
    return res


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def nw_pairs(dim, score, go, ge, syms, offs, pi, pj):
    score = np.ascontiguousarray(score, np.int32)
    syms = np.ascontiguousarray(syms, np.int8)
    offs = np.ascontiguousarray(offs, np.uint32)
    pi = np.ascontiguousarray(pi, np.uint32)
    pj = np.ascontiguousarray(pj, np.uint32)
    npairs = len(pi)
    counts = np.zeros(npairs * dim * dim, np.int32)
    gaps = np.zeros(npairs, np.uint32)
    rc = _l.pgmo_nw_pairs_batch(C.c_uint32(dim), _p(score, C.c_int32), C.c_int32(go), C.c_int32(ge), C.c_uint32(len(offs) - 1),
                                _p(syms, C.c_int8), _p(offs, C.c_uint32), C.c_uint32(npairs), _p(pi, C.c_uint32), _p(pj, C.c_uint32),
                                _p(counts, C.c_int32), _p(gaps, C.c_uint32))
    assert rc == 0
    return counts.reshape(npairs, dim * dim), gaps


def csprofile_create(K, ncols, lprofiles, centre, priors, seq, tau, pi, p_uniform):
    seq = np.ascontiguousarray(seq, np.int8)
    out = np.zeros(20 * (len(seq) + 2), np.float64)
    arrs = [np.ascontiguousarray(a, np.float64) for a in (lprofiles, centre, priors, pi, p_uniform)]
    rc = _l.pgmo_csprofile_create(C.c_uint32(K), C.c_uint32(ncols), _p(arrs[0], C.c_double), _p(arrs[1], C.c_double),
                                  _p(arrs[2], C.c_double), _p(seq, C.c_int8), C.c_uint32(len(seq)), C.c_double(tau),
                                  _p(arrs[3], C.c_double), _p(arrs[4], C.c_double), _p(out, C.c_double))
    assert rc == 0
    return out


# ---- f1 / f3: node profiles of a merge, pair counts of an alignment, ML distances -----------------------------------------
from prographmsa_amd import pgm_merge_job, pgm_mldist_model  # noqa: E402

_l.pgmo_merge_profiles.restype = C.c_int
_l.pgmo_merge_profiles.argtypes = [C.POINTER(pgm_merge_job)]
_l.pgmo_prealigned_counts.restype = C.c_int
_l.pgmo_prealigned_counts.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int8), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
_l.pgmo_mldist.restype = C.c_int
_l.pgmo_mldist.argtypes = [C.POINTER(pgm_mldist_model), C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.POINTER(C.c_double)]


def merge_profiles(job):
    """Oracle node profiles of one pgm_merge_job (writes job.profiles); returns the status."""
    return _l.pgmo_merge_profiles(C.byref(job))


def prealigned_counts(dim, rows, pi, pj):
    """rows: (nrows, ncols) int8.  Returns (counts [npairs * dim * dim] int32, gaps [npairs] uint32)."""
    rows = np.ascontiguousarray(rows, np.int8)
    pi = np.ascontiguousarray(pi, np.uint32); pj = np.ascontiguousarray(pj, np.uint32)
    counts = np.zeros(len(pi) * dim * dim, np.int32); gaps = np.zeros(len(pi), np.uint32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = _l.pgmo_prealigned_counts(dim, rows.shape[0], rows.shape[1], P(rows, C.c_int8), len(pi), P(pi, C.c_uint32), P(pj, C.c_uint32),
                                   P(counts, C.c_int32), P(gaps, C.c_uint32))
    assert rc == 0
    return counts, gaps


def mldist(model, counts, gaps, seqlen):
    """model: pgm_mldist_model.  Returns (dist, var) float64 arrays."""
    counts = np.ascontiguousarray(counts, np.int32); gaps = np.ascontiguousarray(gaps, np.uint32); seqlen = np.ascontiguousarray(seqlen, np.float64)
    n = len(gaps)
    dist = np.zeros(n); var = np.zeros(n)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = _l.pgmo_mldist(C.byref(model), n, P(counts, C.c_int32), P(gaps, C.c_uint32), P(seqlen, C.c_double), P(dist, C.c_double), P(var, C.c_double))
    assert rc == 0
    return dist, var


_l.pgmo_kmer_cosine.restype = C.c_int
_l.pgmo_kmer_cosine.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]


def kmer_cosine(counts):
    """counts: (nseq, ncols) int32 -> flat nseq * nseq float64 (column-major)."""
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(counts.shape[0] * counts.shape[0])
    rc = _l.pgmo_kmer_cosine(counts.shape[0], counts.shape[1], counts.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out
