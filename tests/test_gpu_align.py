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


@pytest.mark.parametrize("kw", [
    dict(skip_frac=0.0, drop_chain_frac=0.0),                 # pure chains (leaf vs leaf)
    dict(skip_frac=0.2),                                      # merged-graph like skip edges
    dict(skip_frac=0.3, repeat_frac=0.05),                    # + tandem-repeat edges
    dict(skip_frac=0.95, skip_max=9, drop_chain_frac=0.0),    # dense extras: > 128 row extras per band, > 7 per node (generic path)
    dict(skip_frac=0.1, skip_span=70, repeat_frac=0.03, repeat_span=90),   # far edges: beyond the LDS history and the traceback tile
])
def test_random_jobs_bit_exact(ctx, kw):
    from prographmsa_amd import jobs as J
    sizes = [(2, 2), (3, 2), (2, 5), (3, 3), (7, 4), (40, 33), (64, 64), (65, 66), (66, 65), (130, 97), (200, 310), (517, 129)]
    js = [J.random_job(1000 + i, n1, n2, **kw) for i, (n1, n2) in enumerate(sizes)]
    b = J.Batch(ctx, js)
    b.run()
    res = b.fetch()
    for i, j in enumerate(js):
        _cmp_job(b, i, j, res[i])
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


def test_empty_batch(ctx):
    from prographmsa_amd import jobs as J
    assert J.align_graphs_batch(ctx, []) == []
