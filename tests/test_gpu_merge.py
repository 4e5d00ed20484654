"""GPU parity of the merged-graph node profiles (numeric part of mergeGraphs, GraphAlign.h:569-620):
pgm_merge_profiles_batch against the oracle's pgmo_merge_profiles (oracle/pgm_oracle.c: Eigen 3.0-3.2 column-major gemv
association, sequential sum of squares, multiplication by the reciprocal) — bit for bit.
The same entry point runs inside every FASTA fixture of tests/test_gpu_e2e.py (the product driver uses it for every level)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GAP = 0xFFFFFFFF


@pytest.mark.parametrize("D", [20, 61])
def test_merge_profiles_bit_exact(ctx, D):
    import oracle_lib
    import prographmsa_amd as pg
    rng = np.random.default_rng(40 + D)
    jobs, keep, want = [], [], []
    for n1, n2 in ((2, 2), (9, 7), (130, 141), (517, 300)):
        def graph(n):
            g = np.zeros((D, n))
            for i in range(1, n - 1):
                if rng.random() < 0.5:
                    g[rng.integers(D), i] = 1.0
                else:
                    v = rng.gamma(0.5, 1.0, D) + 1e-3
                    g[:, i] = v / np.sqrt((v * v).sum())
            return np.asfortranarray(g)
        g1, g2 = graph(n1), graph(n2)
        def pmat():
            P = rng.gamma(0.3, 1.0, (D, D)) + 0.02
            P += np.eye(D) * D * 0.3
            return np.asfortranarray(P / P.sum(1, keepdims=True))
        P1, P2 = pmat(), pmat()
        # a random monotone walk through both graphs: matched pairs, one-sided nodes, skipped nodes (START / END included)
        k1, k2, fl = [], [], []
        i1 = i2 = 0
        while i1 < n1 or i2 < n2:
            r = rng.random()
            if i1 < n1 and i2 < n2 and r < 0.6:
                k1.append(i1); k2.append(i2); fl.append(0); i1 += 1; i2 += 1
            elif i1 < n1 and (r < 0.8 or i2 >= n2):
                k1.append(i1); k2.append(GAP); fl.append(0); i1 += 1
            else:
                k1.append(GAP); k2.append(i2); fl.append(int(rng.random() < 0.5)); i2 += 1
        k1 = np.array(k1, np.uint32); k2 = np.array(k2, np.uint32); fl = np.array(fl, np.uint8)
        out = np.full(D * len(k1), np.nan)
        j = pg.pgm_merge_job()
        j.dim, j.n1, j.n2, j.nnodes = D, n1, n2, len(k1)
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        j.sites1, j.sites2, j.P1, j.P2 = P(g1, C.c_double), P(g2, C.c_double), P(P1, C.c_double), P(P2, C.c_double)
        j.k1, j.k2, j.g2_with_P1, j.profiles = P(k1, C.c_uint32), P(k2, C.c_uint32), P(fl, C.c_uint8), P(out, C.c_double)
        jobs.append(j)
        keep.append((g1, g2, P1, P2, k1, k2, fl, out))
        ref = np.full(D * len(k1), np.nan)
        jo = pg.pgm_merge_job()
        C.memmove(C.byref(jo), C.byref(j), C.sizeof(jo))
        jo.profiles = P(ref, C.c_double)
        assert oracle_lib.merge_profiles(jo) == 0
        want.append(ref)
    arr = (pg.pgm_merge_job * len(jobs))(*jobs)
    pg.check(pg.lib.pgm_merge_profiles_batch(ctx.handle, len(jobs), arr))
    for (g1, g2, P1, P2, k1, k2, fl, out), w in zip(keep, want):
        assert np.array_equal(out.view(np.uint64), w.view(np.uint64)), np.argwhere(out.view(np.uint64) != w.view(np.uint64))[:4]


def test_merge_profiles_rejects_bad_mappings(ctx):
    import prographmsa_amd as pg
    D = 20
    g = np.asfortranarray(np.zeros((D, 4)))
    Pm = np.asfortranarray(np.eye(D))
    k1 = np.array([0, 9], np.uint32); k2 = np.array([0, 1], np.uint32); fl = np.zeros(2, np.uint8)
    out = np.zeros(D * 2)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    j = pg.pgm_merge_job()
    j.dim, j.n1, j.n2, j.nnodes = D, 4, 4, 2
    j.sites1, j.sites2, j.P1, j.P2 = P(g, C.c_double), P(g, C.c_double), P(Pm, C.c_double), P(Pm, C.c_double)
    j.k1, j.k2, j.g2_with_P1, j.profiles = P(k1, C.c_uint32), P(k2, C.c_uint32), P(fl, C.c_uint8), P(out, C.c_double)
    assert pg.lib.pgm_merge_profiles_batch(ctx.handle, 1, C.byref(j)) == pg.PGM_ERR_INVALID
