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


def test_resident_profiles_through_merge_and_alignment(ctx):
    """pgm_merge_profiles_batch_ex(PGM_MERGE_RESIDENT) leaves the merged profiles in HBM; a second merge reads them from there and
    pgm_align_graphs_batch_res aligns graphs whose sites are those matrices gathered through a node map (a CleanedGraph): all
    bit-identical to the path through the host."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    D = 20
    rng = np.random.default_rng(77)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))

    def merge_job(g1, g2, out):
        n1, n2 = g1.shape[1], g2.shape[1]
        k1, k2, fl = [], [], []
        i1 = i2 = 0
        while i1 < n1 or i2 < n2:
            r = rng.random()
            if i1 < n1 and i2 < n2 and r < 0.7:
                k1.append(i1); k2.append(i2); fl.append(0); i1 += 1; i2 += 1
            elif i1 < n1 and (r < 0.85 or i2 >= n2):
                k1.append(i1); k2.append(GAP); fl.append(0); i1 += 1
            else:
                k1.append(GAP); k2.append(i2); fl.append(0); i2 += 1
        k1 = np.array(k1, np.uint32); k2 = np.array(k2, np.uint32); fl = np.array(fl, np.uint8)
        Pm = np.asfortranarray(np.eye(D) * 0.7 + 0.3 / D)
        j = pg.pgm_merge_job()
        j.dim, j.n1, j.n2, j.nnodes = D, n1, n2, len(k1)
        j.P1, j.P2 = P(Pm, C.c_double), P(Pm, C.c_double)
        j.k1, j.k2, j.g2_with_P1 = P(k1, C.c_uint32), P(k2, C.c_uint32), P(fl, C.c_uint8)
        return j, (k1, k2, fl, Pm)

    # two alignment jobs give four graphs; merge them pairwise, once through the host and once resident
    js = [J.random_job(510 + i, n1, n2, skip_frac=0.2) for i, (n1, n2) in enumerate([(140, 150), (160, 130)])]
    host_out, keep, mj = [], [], []
    for j in js:
        g1 = np.asfortranarray(j.g1.sites.reshape(j.g1.n, D).T.copy()) if j.g1.sites.ndim == 1 else np.asfortranarray(j.g1.sites)
        g2 = np.asfortranarray(j.g2.sites.reshape(j.g2.n, D).T.copy()) if j.g2.sites.ndim == 1 else np.asfortranarray(j.g2.sites)
        m, k = merge_job(g1, g2, None)
        m.sites1, m.sites2 = P(g1, C.c_double), P(g2, C.c_double)
        out = np.zeros(D * m.nnodes)
        m.profiles = P(out, C.c_double)
        mj.append(m); keep.append((g1, g2, k)); host_out.append(out)
    arr = (pg.pgm_merge_job * 2)(*mj)
    pg.check(pg.lib.pgm_merge_profiles_batch(ctx.handle, 2, arr))
    dev = (C.POINTER(C.c_double) * 2)()
    for m in mj:
        m.profiles = None
    arr = (pg.pgm_merge_job * 2)(*mj)
    pg.check(pg.lib.pgm_merge_profiles_batch_ex(ctx.handle, 2, arr, pg.PGM_MERGE_RESIDENT, dev))
    assert dev[0] and dev[1]
    # (1) a merge whose inputs are the two resident matrices against the same merge from the host copies
    h1 = np.asfortranarray(host_out[0].reshape(-1, D).T.copy()); h2 = np.asfortranarray(host_out[1].reshape(-1, D).T.copy())
    m2, k2_ = merge_job(h1, h2, None)
    want = np.zeros(D * m2.nnodes); got = np.zeros(D * m2.nnodes)
    m2.sites1, m2.sites2, m2.profiles = P(h1, C.c_double), P(h2, C.c_double), P(want, C.c_double)
    pg.check(pg.lib.pgm_merge_profiles_batch(ctx.handle, 1, C.byref(m2)))
    m2.sites1, m2.sites2, m2.profiles = dev[0], dev[1], P(got, C.c_double)
    pg.check(pg.lib.pgm_merge_profiles_batch(ctx.handle, 1, C.byref(m2)))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # (2) an alignment of "cleaned" versions of the two merged graphs: nodes picked by a map, chain edges; host sites against the
    # resident matrices gathered through the same map
    def cleaned(h, seed):
        r = np.random.default_rng(seed)
        n = h.shape[1]
        keepn = np.sort(np.concatenate([[0], r.choice(np.arange(1, n - 1), size=(n - 2) * 3 // 4, replace=False), [n - 1]])).astype(np.uint32)
        return keepn, np.asfortranarray(h[:, keepn])
    map1, s1 = cleaned(h1, 1); map2, s2 = cleaned(h2, 2)
    job = J.random_job(991, len(map1), len(map2), skip_frac=0.15)
    job.g1.sites = np.ascontiguousarray(s1.T).reshape(-1) if job.g1.sites.ndim == 1 else s1
    job.g2.sites = np.ascontiguousarray(s2.T).reshape(-1) if job.g2.sites.ndim == 1 else s2
    ref = J.align_graphs_batch(ctx, [job])[0]
    cj = J.CJobs([job])
    r1 = (pg.pgm_site_ref * 1)(); r2 = (pg.pgm_site_ref * 1)()
    r1[0].dev_sites, r1[0].node_map, r1[0].ncols = dev[0], P(map1, C.c_uint32), h1.shape[1]
    r2[0].dev_sites, r2[0].node_map, r2[0].ncols = dev[1], P(map2, C.c_uint32), h2.shape[1]
    # a node map that points beyond the resident matrix is refused on the host (the device gathers unchecked)
    r2[0].ncols = int(map2.max())
    assert pg.lib.pgm_align_graphs_batch_res(ctx.handle, 1, cj.g1, cj.g2, cj.m, cj.sc, r1, r2, cj.out) == pg.PGM_ERR_INVALID
    r2[0].ncols = h2.shape[1]
    pg.check(pg.lib.pgm_align_graphs_batch_res(ctx.handle, 1, cj.g1, cj.g2, cj.m, cj.sc, r1, r2, cj.out))
    got = cj.results()[0]
    assert np.float32(got["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32)
    assert np.array_equal(got["map1"], ref["map1"]) and np.array_equal(got["map2"], ref["map2"])


def test_resident_onehot_leaves_against_the_oracle(ctx):
    """pgm_resident_onehot (the leaf graphs of a pass built in HBM: reference src/SequenceGraph.h:101-109) checked at unit level
    against the ORACLE: two sequence graphs whose profile columns exist only on the device — one-hot for a residue with a value,
    uniform 1 / dim for one without (negative symbol), zero START / END columns — aligned through pgm_site_ref give the oracle's
    score and mappings for the same graphs with the columns written out on the host."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    import oracle_lib
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for D in (20, 61):
        rng = np.random.default_rng(900 + D)
        lens = [73, 131]
        syms = [rng.integers(0, D, L).astype(np.int8) for L in lens]
        for s in syms:
            s[rng.random(len(s)) < 0.08] = -1          # residues without a value: uniform columns
        syms[0][5] = -2                                 # any negative symbol is "no value"
        flat = np.concatenate(syms).astype(np.int8)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        dev = (C.POINTER(C.c_double) * 2)()
        pg.check(pg.lib.pgm_resident_reset(ctx.handle))
        pg.check(pg.lib.pgm_resident_onehot(ctx.handle, D, 2, P(flat, C.c_int8), P(offs, C.c_uint32), dev))
        # the same columns on the host, as the reference's SequenceGraph lays them out (node-major: n x dim)
        def columns(s):
            m = np.zeros((len(s) + 2, D))
            for i, v in enumerate(s):
                if v >= 0: m[i + 1, v] = 1.0
                else: m[i + 1, :] = 1.0 / D
            return m
        job = J.random_job(4242 + D, lens[0] + 2, lens[1] + 2, dim=D, skip_frac=0.0, drop_chain_frac=0.0)
        job.g1.sites = columns(syms[0]).reshape(-1)
        job.g2.sites = columns(syms[1]).reshape(-1)
        ref = oracle_lib.align_graphs(job)
        cj = J.CJobs([job])
        r1 = (pg.pgm_site_ref * 1)(); r2 = (pg.pgm_site_ref * 1)()
        r1[0].dev_sites, r1[0].node_map, r1[0].ncols = dev[0], None, lens[0] + 2
        r2[0].dev_sites, r2[0].node_map, r2[0].ncols = dev[1], None, lens[1] + 2
        pg.check(pg.lib.pgm_align_graphs_batch_res(ctx.handle, 1, cj.g1, cj.g2, cj.m, cj.sc, r1, r2, cj.out))
        got = cj.results()[0]
        assert np.float32(got["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32)
        assert np.array_equal(got["map1"], ref["map1"]) and np.array_equal(got["map2"], ref["map2"])


def test_resident_import_between_contexts(ctx):
    """pgm_resident_import: a resident matrix of one context copied into another's resident memory (the pass sharded by subtree,
    host/progressive.cpp) — two contexts on this one device: the copy gives the same alignment as the original, an address that is
    not a resident one of the source context is refused, and the source stays valid."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    import oracle_lib
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    other = pg.Context(0)
    D = 20
    rng = np.random.default_rng(77)
    lens = [90, 110]
    syms = [rng.integers(0, D, L).astype(np.int8) for L in lens]
    flat = np.concatenate(syms).astype(np.int8)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    dev = (C.POINTER(C.c_double) * 2)()
    pg.check(pg.lib.pgm_resident_reset(ctx.handle)); pg.check(pg.lib.pgm_resident_reset(other.handle))
    pg.check(pg.lib.pgm_resident_onehot(ctx.handle, D, 2, P(flat, C.c_int8), P(offs, C.c_uint32), dev))
    there = (C.POINTER(C.c_double) * 2)()
    for k in range(2):
        got = C.POINTER(C.c_double)()
        pg.check(pg.lib.pgm_resident_import(other.handle, ctx.handle, dev[k], D * (lens[k] + 2), C.byref(got)))
        there[k] = got
    bogus = C.POINTER(C.c_double)()
    host = np.zeros(64)
    assert pg.lib.pgm_resident_import(other.handle, ctx.handle, P(host, C.c_double), 64, C.byref(bogus)) == pg.PGM_ERR_INVALID
    def columns(s):
        m = np.zeros((len(s) + 2, D))
        for i, v in enumerate(s): m[i + 1, v] = 1.0
        return m
    job = J.random_job(4343, lens[0] + 2, lens[1] + 2, dim=D, skip_frac=0.0, drop_chain_frac=0.0)
    job.g1.sites = columns(syms[0]).reshape(-1); job.g2.sites = columns(syms[1]).reshape(-1)
    ref = oracle_lib.align_graphs(job)
    for handle, where in ((other.handle, there), (ctx.handle, dev)):
        cj = J.CJobs([job])
        r1 = (pg.pgm_site_ref * 1)(); r2 = (pg.pgm_site_ref * 1)()
        r1[0].dev_sites, r1[0].node_map, r1[0].ncols = where[0], None, lens[0] + 2
        r2[0].dev_sites, r2[0].node_map, r2[0].ncols = where[1], None, lens[1] + 2
        pg.check(pg.lib.pgm_align_graphs_batch_res(handle, 1, cj.g1, cj.g2, cj.m, cj.sc, r1, r2, cj.out))
        got = cj.results()[0]
        assert np.float32(got["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32)
        assert np.array_equal(got["map1"], ref["map1"]) and np.array_equal(got["map2"], ref["map2"])
    other.close()
