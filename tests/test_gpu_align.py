"""GPU parity: HIP alignGraphs through the C ABI vs the CPU oracle (bit-exact DP matrices and mappings)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cmp_job(batch, idx, job, res):
    import oracle_lib
    ref = oracle_lib.align_graphs(job, want_matrices=True)
    mats = batch.read_matrices(idx)
    n1, n2 = job.g1.n, job.g2.n
    names = "MXYWS"
    for k in range(5):
        a = mats[k][: n1 - 1, : n2 - 1]
        b = ref["mats"][k][: n1 - 1, : n2 - 1]
        if k == 4:   # S: border row/column 0 hold 0/0 garbage on both sides, never read
            a, b = a[1:, 1:], b[1:, 1:]
        same = (a.view(np.uint32) == b.view(np.uint32))
        assert same.all(), "matrix %s differs at %s (job %d, %dx%d): %r vs %r" % (
            names[k], np.argwhere(~same)[:3], idx, n1, n2, a[~same][:3], b[~same][:3])
    assert res["status"] == ref["status"] == 0
    assert np.float32(res["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32)
    assert res["n_tr_indels"] == ref["n_tr_indels"]
    assert np.array_equal(res["map1"], ref["map1"]) and np.array_equal(res["map2"], ref["map2"])


FAMILIES = [
    dict(skip_frac=0.0, drop_chain_frac=0.0),                 # pure chains (leaf vs leaf)
    dict(skip_frac=0.0, drop_chain_frac=0.0, onehot_frac=1.0),   # ... with one-hot rows throughout: the emission kernel's score table per (symbol, column)
    dict(skip_frac=0.0, drop_chain_frac=0.0, chain_cost_frac=0.2),   # chain-only with costs on some chain edges (lean sweep, R rows per lane)
    dict(skip_frac=0.1, onehot_frac=0.97),                    # almost one-hot: workgroups with and without other rows
    dict(skip_frac=0.2),                                      # merged-graph like skip edges: near window + LDS history
    dict(skip_frac=0.3, repeat_frac=0.05),                    # + tandem-repeat edges
    dict(skip_frac=0.4, skip_span=3, skip_max=2),             # only near predecessors (distance 2 and 3): the register window alone
    dict(skip_frac=0.95, skip_max=9, drop_chain_frac=0.0),    # dense extras: more far edges per node than the history serves (generic path)
    dict(skip_frac=0.1, skip_span=70, repeat_frac=0.03, repeat_span=90),   # far edges: beyond the LDS history and the traceback tile
    dict(skip_frac=0.25, skip_span=27, skip_max=2),           # distances up to the on-chip limit: deepest history (64 steps), virtual lanes
    # heavy-tailed graphs (MODE 2 with the long / remote entries of the far helpers):
    dict(skip_frac=0.3, skip_span=150, skip_max=5),           # several long edges per node: long slots 7, 6, 5, remote rows, > 3 long: generic
    dict(skip_frac=0.5, skip_span=27, skip_max=14),           # 9-14 on-chip entries per node: overflow table (and its 48-record limit), row CSR
    dict(skip_frac=0.4, skip_span=60, skip_max=10, repeat_frac=0.05, repeat_span=120),   # everything mixed
]


@pytest.mark.parametrize("dim", [20, 61])
@pytest.mark.parametrize("kw", FAMILIES)
def test_random_jobs_bit_exact(ctx, kw, dim):
    from prographmsa_amd import jobs as J
    sizes = [(2, 2), (3, 2), (2, 5), (3, 3), (7, 4), (40, 33), (64, 64), (65, 66), (66, 65), (130, 97), (200, 310), (517, 129)]
    if kw.get("skip_frac", 1) == 0.0:   # chain-only: more than eight bands of 128 / 256 rows (the worker's wavefronts take a second band), ring wrap-around
        sizes = sizes + [(129, 130), (257, 70), (1100, 700), (2500, 90), (90, 2300)]
    if dim == 61:   # the 61-state alphabet (codons): fewer sizes, same families incl. skip and repeat edges
        sizes = [(3, 2), (7, 4), (65, 66), (130, 97), (200, 310)]
    if kw.get("skip_span", 0) >= 60 or kw.get("skip_max", 0) >= 10:   # several bands with remote rows and long columns in flight
        sizes = sizes + [(700, 650)]
    js = [J.random_job(1000 + i, n1, n2, dim=dim, **kw) for i, (n1, n2) in enumerate(sizes)]
    b = J.Batch(ctx, js, keep_matrices=True)
    b.run()
    res = b.fetch()
    for i, j in enumerate(js):
        _cmp_job(b, i, j, res[i])
    b.close()
    # the product's form of the batch (no test hook: chain-only jobs keep decision bytes instead of the matrices): same results
    b = J.Batch(ctx, js)
    b.run()
    res2 = b.fetch()
    for r, r2 in zip(res, res2):
        assert r["status"] == r2["status"] == 0 and r["n_tr_indels"] == r2["n_tr_indels"]
        assert np.float32(r["score"]).view(np.uint32) == np.float32(r2["score"]).view(np.uint32)
        assert np.array_equal(r["map1"], r2["map1"]) and np.array_equal(r["map2"], r2["map2"])
    if kw.get("skip_frac", 1) == 0.0 and len(js) > 12:
        import prographmsa_amd as pg
        with pytest.raises(pg.PgmError):
            b.read_matrices(12)   # not kept
    b.close()


def test_large_heavy_tailed_jobs_bit_exact(ctx):
    """Jobs of the size of the roots of configs 3-5 with heavy-tailed edge lengths: dozens of MODE 2 bands with long column
    entries, remote row entries and overflow columns in flight at once (tools/probe_big.py goes to 6000 x 5000)."""
    from prographmsa_amd import jobs as J
    js = [J.random_job(77, 2300, 2200, skip_frac=0.3, skip_span=150, skip_max=5),
          J.random_job(78, 1500, 2600, skip_frac=0.5, skip_span=27, skip_max=14),
          J.random_job(79, 2600, 1400, skip_frac=0.25, skip_span=90, skip_max=6, repeat_frac=0.03, repeat_span=120)]
    b = J.Batch(ctx, js, keep_matrices=True)
    b.run()
    res = b.fetch()
    for i, j in enumerate(js):
        _cmp_job(b, i, j, res[i])
    b.close()


def test_critical_path_kernel_and_wide_bands_bit_exact(ctx):
    """Jobs of pgm_crit_kernel (20 and more bands, every predecessor near or within the on-chip history: the chain terms on one wavefront,
    everything else on fifteen others) and of pgm_band_kernel's wide workers (a 32-step history, fewer than 20 bands), 20 and 61 states:
    all four DP matrices, scores and mappings against the oracle; in one batch, so that workers take several bands one after another."""
    from prographmsa_amd import jobs as J
    for dim in (20, 61):
        js = [J.random_job(7100 + dim, 1500, 1400, dim=dim, skip_frac=0.25, skip_span=27, skip_max=2, drop_chain_frac=0.0),   # remote rows: the helper sweep of the fill kernel
              J.random_job(7200 + dim, 1300, 700, dim=dim, skip_frac=0.1, skip_span=9, skip_max=3, drop_chain_frac=0.0),    # near and a few far edges, 21 bands: pgm_crit_kernel
              J.random_job(7300 + dim, 1000, 1100, dim=dim, skip_frac=0.2, skip_span=20, skip_max=2, drop_chain_frac=0.0),
              J.random_job(7400 + dim, 1290, 90, dim=dim, skip_frac=0.5, skip_span=5, skip_max=3, drop_chain_frac=0.0),     # short columns: the sweep is mostly ramp (pgm_crit_kernel)
              J.random_job(7500 + dim, 400, 380, dim=dim, skip_frac=0.2),                                                   # (pgm_crit_kernel)
              # few far edges, a 32-step history, chains well below the longest job's: the wide workers of pgm_band_kernel
              J.random_job(7600 + dim, 450, 400, dim=dim, skip_frac=0.03, skip_span=9, skip_max=2, drop_chain_frac=0.0),
              J.random_job(7700 + dim, 600, 300, dim=dim, skip_frac=0.02, skip_span=12, skip_max=2, drop_chain_frac=0.0),
              J.random_job(7800 + dim, 300, 500, dim=dim, skip_frac=0.04, skip_span=10, skip_max=3, drop_chain_frac=0.0)]
        b = J.Batch(ctx, js, keep_matrices=True)
        b.run()
        res = b.fetch()
        for i, j in enumerate(js):
            _cmp_job(b, i, j, res[i])
        b.close()


def test_job_modes_of_small_and_large_batches_bit_exact(ctx):
    """A batch that would not fill the device as MODE 2 sweeps sends every job of 8 bands or more to the critical-path kernel
    (pgm_align_batch_create: a guide-tree level on its own); the same jobs among enough others stay on the band kernel's wavefronts.
    Both ways: all four DP matrices, scores and mappings of those jobs against the oracle, and the fillers' mappings."""
    from prographmsa_amd import jobs as J
    import oracle_lib
    probes = [J.random_job(8100, 700, 650, skip_frac=0.2, skip_span=20, skip_max=2, drop_chain_frac=0.0),
              J.random_job(8101, 640, 720, skip_frac=0.1, skip_span=9, skip_max=3, drop_chain_frac=0.0),
              J.random_job(8102, 580, 600, skip_frac=0.3, repeat_frac=0.05),
              J.random_job(8103, 900, 300, skip_frac=0.03, skip_span=9, skip_max=2, drop_chain_frac=0.0)]
    fillers = [J.random_job(8200 + k, 900, 900, skip_frac=0.15, skip_span=12, skip_max=2, drop_chain_frac=0.0) for k in range(64)]
    for js in (probes, probes + fillers):
        b = J.Batch(ctx, js, keep_matrices=True)
        b.run()
        res = b.fetch()
        for i in range(len(probes)):
            _cmp_job(b, i, js[i], res[i])
        for i in range(len(probes), len(js), 9):
            ref = oracle_lib.align_graphs(js[i])
            assert res[i]["status"] == 0 and np.array_equal(res[i]["map1"], ref["map1"]) and np.array_equal(res[i]["map2"], ref["map2"]), i
        b.close()


def test_handoff_timeout_in_the_critical_path_kernel(ctx):
    """The same time-out path as below for a job swept by pgm_crit_kernel: band 3 never publishes its progress, the chain wavefront of band 4
    gives up, raises the abort flag, every wavefront of every worker leaves, the batch reports PGM_ERR_DEVICE — and runs clean afterwards."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    import oracle_lib
    js = [J.random_job(7600, 1400, 600, skip_frac=0.2, skip_span=20, skip_max=2, drop_chain_frac=0.0)]
    b = J.Batch(ctx, js)
    pg.check(pg.lib.pgm_align_batch_test_stall(b.handle, 0, 3, 2000))
    b.run()
    rc = pg.lib.pgm_align_batch_fetch(ctx.handle, b.handle, b.cj.out)
    assert rc == pg.PGM_ERR_DEVICE and b"timed out" in pg.lib.pgm_last_error()
    pg.check(pg.lib.pgm_align_batch_test_stall(b.handle, 0xFFFFFFFF, 0, 0))
    b.run()
    res = b.fetch()
    ref = oracle_lib.align_graphs(js[0])
    assert res[0]["status"] == 0 and np.array_equal(res[0]["map1"], ref["map1"]) and np.array_equal(res[0]["map2"], ref["map2"])
    b.close()


def test_relaunch_and_fetch(ctx):
    """run, run, fetch and run, fetch, fetch: the result records in the pinned block are reset by every launch and copied
    out by fetch while the kernel is still running (status word last); every fetch returns the same results."""
    from prographmsa_amd import jobs as J
    js = [J.random_job(300 + i, 260 + 37 * i, 300 - 11 * i, skip_frac=0.2) for i in range(12)]
    b = J.Batch(ctx, js, keep_matrices=True)
    b.run(); b.run()
    r1 = b.fetch()
    b.run()
    r2 = b.fetch()
    r3 = b.fetch()
    for a in (r2, r3):
        for x, y in zip(r1, a):
            assert np.array_equal(x["map1"], y["map1"]) and np.array_equal(x["map2"], y["map2"])
            assert np.float32(x["score"]).view(np.uint32) == np.float32(y["score"]).view(np.uint32) and x["status"] == y["status"] == 0
    for i, j in enumerate(js):
        _cmp_job(b, i, j, r1[i])
    b.close()


def test_one_call_entry_point(ctx):
    import oracle_lib
    from prographmsa_amd import jobs as J
    js = [J.random_job(7 + i, 90 + 13 * i, 120 - 7 * i, skip_frac=0.15) for i in range(5)]
    res = J.align_graphs_batch(ctx, js)
    for j, r in zip(js, res):
        ref = oracle_lib.align_graphs(j)
        assert np.array_equal(r["map1"], ref["map1"]) and np.array_equal(r["map2"], ref["map2"])
        assert np.float32(r["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32)


def test_invalid_graphs_are_rejected(ctx):
    """PGM_ERR_INVALID: an edge to a later node (Graph.h:43 guarantees from < to) and a dimension mismatch."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    good = J.random_job(5, 30, 28, skip_frac=0.2)
    bad = J.random_job(6, 30, 28, skip_frac=0.2)
    bad.g1.e_col[int(bad.g1.e_rowptr[10])] = 17          # node 10 <- node 17
    cj = J.CJobs([good, bad])
    rc = pg.lib.pgm_align_graphs_batch(ctx.handle, cj.n, cj.g1, cj.g2, cj.m, cj.sc, cj.out)
    assert rc == pg.PGM_ERR_INVALID and b"job 1" in pg.lib.pgm_last_error()
    mism = J.random_job(7, 30, 28, skip_frac=0.2)
    other = J.random_job(8, 30, 28, dim=61, skip_frac=0.2)
    mism.g2 = other.g2                                    # 20-state graph against a 61-state graph
    cj = J.CJobs([mism])
    rc = pg.lib.pgm_align_graphs_batch(ctx.handle, cj.n, cj.g1, cj.g2, cj.m, cj.sc, cj.out)
    assert rc == pg.PGM_ERR_INVALID
    # the context is still usable afterwards
    assert J.align_graphs_batch(ctx, [good])[0]["status"] == 0


def test_handoff_timeout_aborts_the_batch(ctx):
    """PGM_ERR_DEVICE: band 0 of job 0 never publishes its progress (test knob), the band below times out after a
    shortened spin limit, raises the abort flag and every job of the batch reports a device error instead of hanging."""
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    js = [J.random_job(21, 200, 150, skip_frac=0.2), J.random_job(22, 90, 80, skip_frac=0.0, drop_chain_frac=0.0)]
    b = J.Batch(ctx, js)
    pg.check(pg.lib.pgm_align_batch_test_stall(b.handle, 0, 0, 2000))
    b.run()
    rc = pg.lib.pgm_align_batch_fetch(ctx.handle, b.handle, b.cj.out)
    assert rc == pg.PGM_ERR_DEVICE and b"timed out" in pg.lib.pgm_last_error()
    pg.check(pg.lib.pgm_align_batch_test_stall(b.handle, 0xFFFFFFFF, 0, 0))
    b.run()                                               # the same batch runs clean afterwards
    res = b.fetch()
    import oracle_lib
    for j, r in zip(js, res):
        ref = oracle_lib.align_graphs(j)
        assert r["status"] == 0 and np.array_equal(r["map1"], ref["map1"]) and np.array_equal(r["map2"], ref["map2"])
    b.close()


def test_job_timeline(ctx):
    """pgm_align_batch_job_times: every job of a mixed batch (chain-only jobs in the lean kernel, bands one per wavefront, a MODE 2 job
    with a pre-linked corridor) has a sweep end and a later publication stamp, all within the launch; pgm_align_batch_stage_times
    counts exactly the fetched launches."""
    import ctypes as C
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J
    js = [J.random_job(900 + i, n1, n2, **kw) for i, (n1, n2, kw) in enumerate([
        (300, 280, dict(skip_frac=0.0, drop_chain_frac=0.0)), (500, 510, dict(skip_frac=0.2)), (260, 300, dict(skip_frac=0.3, repeat_frac=0.05)),
        (1400, 1350, dict(skip_frac=0.25, skip_span=27, skip_max=2)), (90, 80, dict(skip_frac=0.0, drop_chain_frac=0.0))])]
    b = J.Batch(ctx, js)
    for _ in range(3):
        b.run(); res = b.fetch()
    assert all(r["status"] == 0 for r in res)
    t = np.zeros(2 * len(js), np.uint64)
    pg.check(pg.lib.pgm_align_batch_job_times(ctx.handle, b.handle, t.ctypes.data_as(C.POINTER(C.c_uint64))))
    t = t.reshape(-1, 2).astype(np.int64)
    assert (t > 0).all() and (t[:, 1] >= t[:, 0]).all()
    assert (t.max() - t.min()) < 100 * 100000          # within 100 ms (10 ns ticks)
    ms = b.stage_times(reset=True)
    assert ms[3] == 3 and ms[2] > 0 and b.stage_times()[3] == 0
    import oracle_lib
    for j, r in zip(js, res):
        ref = oracle_lib.align_graphs(j)
        assert np.array_equal(r["map1"], ref["map1"]) and np.array_equal(r["map2"], ref["map2"])
    b.close()


def test_empty_batch(ctx):
    from prographmsa_amd import jobs as J
    assert J.align_graphs_batch(ctx, []) == []


def test_sequence_graph_jobs_score_table_after_other_batches(ctx):
    """Chain-only jobs of two SEQUENCE graphs (every profile column one-hot, uniform or empty) take their scores from a per-job class
    table instead of the score matrix, and pgm_prep_kernel only works on the slices that hold a class's representative node or the END
    node (PgmJob::cls1, pgm_classify_kernel, pgm_lean_kernel).  Batches of such jobs on a context whose cached buffers earlier batches
    of other shapes have used (round 4: the END node's edge record of a skipped slice was read stale — found by tools/fuzz_e2e.py on
    the second pass of a run), 20 and 61 states, residues without a value: scores and mappings against the oracle."""
    from prographmsa_amd import jobs as J
    import oracle_lib
    def seqjob(seed, L1, L2, D):
        rng = np.random.default_rng(seed)
        job = J.random_job(seed, L1 + 2, L2 + 2, dim=D, skip_frac=0.0, drop_chain_frac=0.0)
        for g, L in ((job.g1, L1), (job.g2, L2)):
            m = np.zeros((L + 2, D))
            s = rng.integers(0, D, L)
            for i, v in enumerate(s):
                if rng.random() < 0.05: m[i + 1, :] = 1.0 / D      # a residue without a value
                else: m[i + 1, v] = 1.0
            g.sites = m.reshape(-1)
        return job
    for D in (20, 61):
        for rnd in range(3):
            mixed = [J.random_job(9100 + 10 * rnd + k, 300 + 37 * k, 280 + 41 * k, dim=D, skip_frac=0.2) for k in range(3)] + \
                    [seqjob(9200 + 10 * rnd + k, 150 + 61 * k, 170 + 53 * k, D) for k in range(3)]
            only = [seqjob(9300 + 10 * rnd + k, 120 + 47 * k + 13 * rnd, 140 + 43 * k + 7 * rnd, D) for k in range(6)]
            for js in (mixed, only):
                b = J.Batch(ctx, js); b.run(); res = b.fetch(); b.close()
                for i, j in enumerate(js):
                    ref = oracle_lib.align_graphs(j)
                    assert np.float32(res[i]["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32), (D, rnd, i)
                    assert np.array_equal(res[i]["map1"], ref["map1"]) and np.array_equal(res[i]["map2"], ref["map2"]), (D, rnd, i)
