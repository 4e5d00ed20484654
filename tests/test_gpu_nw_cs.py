"""GPU parity for the two satellite stages through the C ABI: all-pairs NW counts (bit-exact int32) and
context-specific profiles (fp64, tolerance 1e-12 relative: only the device exp() differs from glibc)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nw_gpu(ctx, dim, score, go, ge, syms, offs, pi, pj):
    import prographmsa_amd as pg
    score = np.ascontiguousarray(score, np.int32); syms = np.ascontiguousarray(syms, np.int8)
    offs = np.ascontiguousarray(offs, np.uint32); pi = np.ascontiguousarray(pi, np.uint32); pj = np.ascontiguousarray(pj, np.uint32)
    npairs = len(pi)
    counts = np.full(max(1, npairs * dim * dim), -1, np.int32)
    gaps = np.zeros(max(1, npairs), np.uint32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    pg.check(pg.lib.pgm_nw_pairs_batch(ctx.handle, dim, P(score, C.c_int32), go, ge, len(offs) - 1, P(syms, C.c_int8), P(offs, C.c_uint32),
                                       npairs, P(pi, C.c_uint32), P(pj, C.c_uint32), P(counts, C.c_int32), P(gaps, C.c_uint32)))
    return counts[: npairs * dim * dim].reshape(npairs, dim * dim), gaps[:npairs]


def _score(dim, rng):
    s = rng.integers(-4, 3, (dim + 1, dim + 1)).astype(np.int32)
    s = np.minimum(s, s.T)
    s[np.arange(dim + 1), np.arange(dim + 1)] = rng.integers(4, 12, dim + 1)
    return s.reshape(-1)


@pytest.mark.parametrize("dim", [20, 61])
def test_nw_pairs_bit_exact(ctx, dim):
    import oracle_lib
    rng = np.random.default_rng(5 + dim)
    # the kernel sweeps bands of 512 rows (8 rows per lane) and stores 2 steps per direction word: lengths around those edges
    lens = [0, 1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 200, 333, 511, 512, 513, 700, 1030]
    base = rng.integers(0, dim, 1100)
    seqs = []
    for L in lens:   # related sequences so that the traceback has matches, gaps and mismatches
        s = base[:L].copy()
        mut = rng.random(L) < 0.2
        s[mut] = rng.integers(0, dim + 1, mut.sum())   # dim == the "invalid -> 20" style extra symbol when dim == 20
        if L > 10:
            cut = int(rng.integers(1, L - 5))
            s = np.concatenate([s[:cut], s[cut + int(rng.integers(1, 5)):]])
        seqs.append(np.minimum(s, dim).astype(np.int8))
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    syms = np.concatenate(seqs) if len(seqs) else np.zeros(0, np.int8)
    pi, pj = zip(*[(i, j) for i in range(len(seqs)) for j in range(len(seqs)) if i != j])
    score = _score(dim, rng)
    cg, gg = _nw_gpu(ctx, dim, score, -10, -2, syms, offs, pi, pj)
    co, go_ = oracle_lib.nw_pairs(dim, score, -10, -2, syms, offs, pi, pj)
    assert np.array_equal(cg, co)
    assert np.array_equal(gg, go_)


def test_nw_two_tiles_in_flight_reduced_and_pinned(ctx):
    """pgm_nw_pairs_submit / _wait: two tiles in flight on one context (pageable and pinned result buffers), a third submit is
    refused, PGM_NW_REDUCED returns (trace, sum) of the count matrices; all against the oracle's full matrices."""
    import oracle_lib
    import prographmsa_amd as pg
    rng = np.random.default_rng(77)
    dim = 20
    base = rng.integers(0, dim, 700)
    seqs = []
    for L in [40, 300, 511, 513, 640, 700, 90, 257, 129, 64]:
        s_ = base[:L].copy()
        mut = rng.random(L) < 0.25
        s_[mut] = rng.integers(0, dim + 1, mut.sum())
        seqs.append(np.minimum(s_, dim).astype(np.int8))
    offs = np.concatenate([[0], np.cumsum([len(s_) for s_ in seqs])]).astype(np.uint32)
    syms = np.concatenate(seqs)
    pairs = [(i, j) for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
    score = np.ascontiguousarray(_score(dim, rng), np.int32)
    co, go_ = oracle_lib.nw_pairs(dim, score, -10, -2, syms, offs, [a for a, _ in pairs], [b for _, b in pairs])
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    half = len(pairs) // 2
    tiles = [pairs[:half], pairs[half:]]
    for flags, per in ((0, dim * dim), (pg.PGM_NW_REDUCED, 2)):
        for pinned in (False, True):
            bufs, tickets, keep = [], [], []
            for tl in tiles:
                pi = np.array([a for a, _ in tl], np.uint32); pj = np.array([b for _, b in tl], np.uint32)
                if pinned:
                    cp, gp = pg.lib.pgm_host_alloc(len(tl) * per * 4), pg.lib.pgm_host_alloc(len(tl) * 4)
                    assert cp and gp
                    c = np.ctypeslib.as_array(C.cast(cp, C.POINTER(C.c_int32)), (len(tl) * per,)); g = np.ctypeslib.as_array(C.cast(gp, C.POINTER(C.c_uint32)), (len(tl),))
                    keep.append((cp, gp))
                else:
                    c, g = np.full(len(tl) * per, -7, np.int32), np.zeros(len(tl), np.uint32)
                t = C.c_int(-1)
                pg.check(pg.lib.pgm_nw_pairs_submit(ctx.handle, dim, P(score, C.c_int32), -10, -2, len(seqs), P(syms, C.c_int8), P(offs, C.c_uint32), len(tl),
                                                    P(pi, C.c_uint32), P(pj, C.c_uint32), flags, P(c, C.c_int32), P(g, C.c_uint32), C.byref(t)))
                bufs.append((c, g, pi, pj)); tickets.append(t.value)
            assert sorted(tickets) == [0, 1]
            t = C.c_int(-1)
            c3, g3 = np.zeros(per, np.int32), np.zeros(1, np.uint32)
            rc = pg.lib.pgm_nw_pairs_submit(ctx.handle, dim, P(score, C.c_int32), -10, -2, len(seqs), P(syms, C.c_int8), P(offs, C.c_uint32), 1,
                                            P(bufs[0][2], C.c_uint32), P(bufs[0][3], C.c_uint32), flags, P(c3, C.c_int32), P(g3, C.c_uint32), C.byref(t))
            assert rc == pg.PGM_ERR_INVALID                       # a third tile without a wait
            for k in (0, 1):
                pg.check(pg.lib.pgm_nw_pairs_wait(ctx.handle, tickets[k]))
            assert pg.lib.pgm_nw_pairs_wait(ctx.handle, 0) == pg.PGM_ERR_INVALID   # nothing in flight any more
            got_c = np.concatenate([b[0] for b in bufs]).reshape(len(pairs), per)
            got_g = np.concatenate([b[1] for b in bufs])
            assert np.array_equal(got_g, go_)
            if flags:
                cm = co.reshape(len(pairs), dim, dim)
                assert np.array_equal(got_c[:, 0], np.trace(cm, axis1=1, axis2=2)) and np.array_equal(got_c[:, 1], cm.sum((1, 2)))
            else:
                assert np.array_equal(got_c, co)
            del bufs, got_c, got_g
            for cp, gp in keep:
                pg.lib.pgm_host_free(cp); pg.lib.pgm_host_free(gp)


def test_nw_empty(ctx):
    c, g = _nw_gpu(ctx, 20, _score(20, np.random.default_rng(0)), -10, -2, np.zeros(4, np.int8), [0, 4], [], [])
    assert c.size == 0 and g.size == 0


def test_csprofile_matches_oracle(ctx):
    import oracle_lib
    import prographmsa_amd as pg
    rng = np.random.default_rng(3)
    K, ncols = 37, 13
    p = rng.gamma(0.3, 1.0, (K, ncols, 20)) + 1e-4
    p /= p.sum(2, keepdims=True)
    w = 1.3 * 0.9 ** np.abs(np.arange(ncols) - ncols // 2)
    lp = np.zeros((K, ncols, 21))
    lp[:, :, :20] = np.log(p) * w[None, :, None]
    centre = p[:, ncols // 2, :].copy()
    priors = np.log(rng.dirichlet(np.ones(K)))
    lens = [0, 1, 5, 12, 13, 14, 100, 257]
    seqs = [rng.integers(0, 21, L).astype(np.int8) for L in lens]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    syms = np.concatenate(seqs).astype(np.int8)
    out_offs = np.concatenate([[0], np.cumsum([20 * (L + 2) for L in lens])]).astype(np.uint64)
    tau = rng.uniform(0.05, 0.9, len(lens))
    pi = rng.dirichlet(np.ones(20) * 5)
    pu = rng.dirichlet(np.ones(20), len(lens))
    out = np.full(int(out_offs[-1]), np.nan)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    lpf, cf, prf, puf = [np.ascontiguousarray(a, np.float64).reshape(-1) for a in (lp, centre, priors, pu)]
    pg.check(pg.lib.pgm_csprofile_load(ctx.handle, K, ncols, P(lpf, C.c_double), P(cf, C.c_double), P(prf, C.c_double)))
    pg.check(pg.lib.pgm_csprofile_create_batch(ctx.handle, len(lens), P(syms, C.c_int8), P(offs, C.c_uint32), P(tau, C.c_double),
                                               P(pi, C.c_double), P(puf, C.c_double), P(out, C.c_double), P(out_offs, C.c_uint64)))
    for s, L in enumerate(lens):
        ref = oracle_lib.csprofile_create(K, ncols, lpf, cf, prf, seqs[s], tau[s], pi, pu[s])
        got = out[int(out_offs[s]): int(out_offs[s + 1])]
        assert np.all(np.isfinite(got))
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=0)


def test_csprofile_profiles_left_on_the_device(ctx):
    """pgm_csprofile_create_batch_res (the leaf graphs of a resident pass with --cs_profile): the matrices stay in the context's
    resident arena.  Two such leaves aligned through pgm_site_ref give the score bits and mappings of the same two graphs with the
    host variant's columns uploaded the ordinary way — the kernel and its output are the same, only the destination differs."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    rng = np.random.default_rng(5)
    K, ncols = 23, 13
    p = rng.gamma(0.3, 1.0, (K, ncols, 20)) + 1e-4
    p /= p.sum(2, keepdims=True)
    w = 1.3 * 0.9 ** np.abs(np.arange(ncols) - ncols // 2)
    lp = np.zeros((K, ncols, 21))
    lp[:, :, :20] = np.log(p) * w[None, :, None]
    centre = p[:, ncols // 2, :].copy()
    priors = np.log(rng.dirichlet(np.ones(K)))
    lens = [83, 140]
    seqs = [rng.integers(0, 21, L).astype(np.int8) for L in lens]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    syms = np.concatenate(seqs).astype(np.int8)
    out_offs = np.concatenate([[0], np.cumsum([20 * (L + 2) for L in lens])]).astype(np.uint64)
    tau = rng.uniform(0.05, 0.9, len(lens))
    pi = rng.dirichlet(np.ones(20) * 5)
    pu = rng.dirichlet(np.ones(20), len(lens))
    out = np.full(int(out_offs[-1]), np.nan)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    lpf, cf, prf, puf = [np.ascontiguousarray(a, np.float64).reshape(-1) for a in (lp, centre, priors, pu)]
    pg.check(pg.lib.pgm_csprofile_load(ctx.handle, K, ncols, P(lpf, C.c_double), P(cf, C.c_double), P(prf, C.c_double)))
    pg.check(pg.lib.pgm_csprofile_create_batch(ctx.handle, 2, P(syms, C.c_int8), P(offs, C.c_uint32), P(tau, C.c_double),
                                               P(pi, C.c_double), P(puf, C.c_double), P(out, C.c_double), P(out_offs, C.c_uint64)))
    dev = (C.POINTER(C.c_double) * 2)()
    pg.check(pg.lib.pgm_resident_reset(ctx.handle))
    pg.check(pg.lib.pgm_csprofile_create_batch_res(ctx.handle, 2, P(syms, C.c_int8), P(offs, C.c_uint32), P(tau, C.c_double),
                                                   P(pi, C.c_double), P(puf, C.c_double), dev))
    job = J.random_job(4343, lens[0] + 2, lens[1] + 2, dim=20, skip_frac=0.0, drop_chain_frac=0.0)
    job.g1.sites = out[int(out_offs[0]): int(out_offs[1])].copy()
    job.g2.sites = out[int(out_offs[1]): int(out_offs[2])].copy()
    b = J.Batch(ctx, [job])
    b.run()
    want = b.fetch()[0]
    b.close()
    cj = J.CJobs([job])
    r1 = (pg.pgm_site_ref * 1)(); r2 = (pg.pgm_site_ref * 1)()
    r1[0].dev_sites, r1[0].node_map, r1[0].ncols = dev[0], None, lens[0] + 2
    r2[0].dev_sites, r2[0].node_map, r2[0].ncols = dev[1], None, lens[1] + 2
    pg.check(pg.lib.pgm_align_graphs_batch_res(ctx.handle, 1, cj.g1, cj.g2, cj.m, cj.sc, r1, r2, cj.out))
    got = cj.results()[0]
    assert want["status"] == 0 and np.float32(got["score"]).view(np.uint32) == np.float32(want["score"]).view(np.uint32)
    assert np.array_equal(got["map1"], want["map1"]) and np.array_equal(got["map2"], want["map2"])


def test_csprofile_config5_scale_sample(ctx):
    """BASELINE config 5 scale: the synthetic K = 4000 library and 1024 leaves x 600 residues of bench.py's `csprofile` record
    (same generator, same seeds); every 64th leaf is compared with the oracle (the library streams through LDS in 250 chunks)."""
    import oracle_lib
    import prographmsa_amd as pg
    rng = np.random.default_rng(5)
    K, ncols, nleaf, L = 4000, 13, 1024, 600
    p = rng.gamma(0.3, 1.0, (K, ncols, 20)) + 1e-4
    p /= p.sum(2, keepdims=True)
    w = 1.3 * 0.9 ** np.abs(np.arange(ncols) - ncols // 2)
    lp = np.zeros((K, ncols, 21))
    lp[:, :, :20] = np.log(p) * w[None, :, None]
    lpf = np.ascontiguousarray(lp, np.float64).reshape(-1)
    cf = np.ascontiguousarray(p[:, ncols // 2, :], np.float64).reshape(-1)
    prf = np.log(rng.dirichlet(np.ones(K)))
    syms = rng.integers(0, 20, nleaf * L).astype(np.int8)
    offs = (np.arange(nleaf + 1) * L).astype(np.uint32)
    out_offs = (np.arange(nleaf + 1) * 20 * (L + 2)).astype(np.uint64)
    tau = np.full(nleaf, 0.3)
    pi = np.full(20, 0.05)
    pu = np.full(nleaf * 20, 0.05)
    out = np.full(int(out_offs[-1]), np.nan)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    pg.check(pg.lib.pgm_csprofile_load(ctx.handle, K, ncols, P(lpf, C.c_double), P(cf, C.c_double), P(prf, C.c_double)))
    pg.check(pg.lib.pgm_csprofile_create_batch(ctx.handle, nleaf, P(syms, C.c_int8), P(offs, C.c_uint32), P(tau, C.c_double),
                                               P(pi, C.c_double), P(pu, C.c_double), P(out, C.c_double), P(out_offs, C.c_uint64)))
    assert np.all(np.isfinite(out))
    for s in range(0, nleaf, 64):
        ref = oracle_lib.csprofile_create(K, ncols, lpf, cf, prf, syms[s * L:(s + 1) * L], 0.3, pi, pu[s * 20:(s + 1) * 20])
        np.testing.assert_allclose(out[int(out_offs[s]): int(out_offs[s + 1])], ref, rtol=1e-12, atol=0)
