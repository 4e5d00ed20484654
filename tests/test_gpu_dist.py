"""GPU parity of the guide-tree tail: batched ML distances and the pair counts of an alignment.

pgm_prealigned_counts_batch is integer work: bit-exact against the oracle's pgmo_prealigned_counts
(src/DistanceFactoryPrealigned.h:34-90).  pgm_mldist_batch against the oracle's pgmo_mldist on synthetic reversible models.  pgm_mldist_batch keeps the host estimator's operation order (host/distance.cpp,
the mirror of src/DistanceFactoryML.h:66-190) but uses the device library's exp / log: tolerance 1e-12 relative, checked
through the product driver (`pgmsa --dump_dist`, host estimator vs PGM_DEVICE_MLDIST=1) on the committed families."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scan(r1, r2, D):
    """The reference's loop over the columns of two aligned rows: the oracle's pgmo_prealigned_counts (oracle/pgm_oracle.c,
    src/DistanceFactoryPrealigned.h:49-79) on one pair."""
    import oracle_lib
    c, g = oracle_lib.prealigned_counts(D, np.array([r1, r2], np.int8), [0], [1])
    return c, int(g[0])


@pytest.mark.parametrize("D,nrows,L", [(20, 7, 1), (20, 9, 63), (20, 12, 64), (20, 10, 65), (20, 8, 1000), (61, 6, 333), (20, 40, 3274)])
def test_prealigned_pair_counts_bit_exact(ctx, D, nrows, L):
    import prographmsa_amd as pg
    rng = np.random.default_rng(100 + L)
    rows = rng.integers(0, D, (nrows, L)).astype(np.int8)
    rows[rng.random((nrows, L)) < 0.25] = -1                       # gaps, in runs and alone
    for r in range(nrows):
        for _ in range(3):
            a = int(rng.integers(0, L)); rows[r, a:a + int(rng.integers(1, 40))] = -1
    rows[rng.random((nrows, L)) < 0.02] = -2                       # residues without a value
    if L > 10:
        rows[0, :] = -1                                            # an all-gap row
        rows[1, : L // 2] = -1
    pairs = [(i, j) for i in range(nrows) for j in range(i + 1, nrows)]
    pi = np.array([p[0] for p in pairs], np.uint32)
    pj = np.array([p[1] for p in pairs], np.uint32)
    counts = np.zeros(len(pairs) * D * D, np.int32)
    gaps = np.zeros(len(pairs), np.uint32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    pg.check(pg.lib.pgm_prealigned_counts_batch(ctx.handle, D, nrows, L, P(rows, C.c_int8), len(pairs), P(pi, C.c_uint32), P(pj, C.c_uint32),
                                                P(counts, C.c_int32), P(gaps, C.c_uint32)))
    for k, (i, j) in enumerate(pairs):
        rc, rg = _scan(rows[i].tolist(), rows[j].tolist(), D)
        assert np.array_equal(counts[k * D * D:(k + 1) * D * D], rc), (i, j)
        assert gaps[k] == rg, (i, j, gaps[k], rg)


def _dump(args, fasta, tmp_path, device):
    import prographmsa_amd as pg
    path = str(tmp_path / ("dist_%d.bin" % device))
    env = dict(os.environ)
    env.pop("PGM_DEVICE_MLDIST", None)
    env.pop("PGM_HOST_COUNTS", None)
    if device:
        env["PGM_DEVICE_MLDIST"] = "1"
    else:
        env["PGM_HOST_COUNTS"] = "1"     # (the pair counts of an alignment are on the device by default: bit-exact integers)
    r = subprocess.run([pg.PGMSA_PATH] + args + ["--dump_dist", path, fasta], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    buf = open(path, "rb").read()
    mats, off = [], 0
    while off < len(buf):
        n = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
        d = np.frombuffer(buf, np.float64, n * n, off); off += 8 * n * n
        v = np.frombuffer(buf, np.float64, n * n, off); off += 8 * n * n
        mats.append((d, v))
    return r.stdout, mats


@pytest.mark.parametrize("case,args,nmat", [
    ("c1.fa", ["-a", "-m", "-T", "-i", "0"], 1),            # alignPair counts -> ML distances (-m)
    ("c2.fa", ["-a", "-m", "-T", "-i", "0"], 1),
    ("c1.fa", ["-a", "-M", "-T", "-i", "0"], 1),            # ... with the gap term of --mldist_gap
    ("c1.fa", ["--fasta", "-a", "-m"], 2),                  # + the distances induced by the alignment of the first round (prealigned counts on the device)
    ("c2.fa", ["--fasta", "-a"], 2),                        # p-distances: the estimator's non-ML branch
])
def test_device_distances_match_the_host_estimator(tmp_path, case, args, nmat):
    fasta = os.path.join(GOLD, case)
    out_h, mh = _dump(args, fasta, tmp_path, 0)
    out_d, md = _dump(args, fasta, tmp_path, 1)
    assert len(mh) == len(md) and len(mh) >= nmat
    worst, same, total = 0.0, 0, 0
    for (dh, vh), (dd, vd) in zip(mh, md):
        for a, b in ((dh, dd), (vh, vd)):
            worst = max(worst, float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))))
            same += int((a.view(np.uint64) == b.view(np.uint64)).sum()); total += a.size
    assert worst <= 1e-12, worst
    print("device vs host distances: max rel diff %.2e, %d of %d values bit-identical" % (worst, same, total))
    if "-m" not in args and "-M" not in args:
        assert same == total and out_h == out_d      # no exp / log on this path: bit-identical, same output


def test_c3_prealigned_counts_at_full_size(ctx, tmp_path):
    """The 256-row alignment of the headline family (3274 columns, 32 640 pairs): device pair counts vs the scan, on a sample."""
    import prographmsa_amd as pg
    fa = tmp_path / "c3.fa"
    fa.write_text(gen.fasta(gen.gen(256, 1000, 3)))
    r = subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(GOLD, "c3.tree"), str(fa)], capture_output=True, text=True, check=True)
    seqs = [ln for ln in r.stdout.splitlines() if not ln.startswith(">")]
    order = "ACDEFGHIKLMNPQRSTVWY"
    L = len(seqs[0])
    rows = np.array([[order.index(c) if c in order else (-1 if c == "-" else -2) for c in s] for s in seqs], np.int8)
    pairs = [(i, j) for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
    pi = np.array([p[0] for p in pairs], np.uint32); pj = np.array([p[1] for p in pairs], np.uint32)
    counts = np.zeros(len(pairs) * 400, np.int32); gaps = np.zeros(len(pairs), np.uint32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    pg.check(pg.lib.pgm_prealigned_counts_batch(ctx.handle, 20, len(seqs), L, P(rows, C.c_int8), len(pairs), P(pi, C.c_uint32), P(pj, C.c_uint32),
                                                P(counts, C.c_int32), P(gaps, C.c_uint32)))
    rng = np.random.default_rng(1)
    for k in rng.integers(0, len(pairs), 60):
        rc, rg = _scan(rows[pairs[k][0]].tolist(), rows[pairs[k][1]].tolist(), 20)
        assert np.array_equal(counts[k * 400:(k + 1) * 400], rc) and gaps[k] == rg


@pytest.mark.parametrize("flags", [(1, 0), (0, 1), (0, 0)])
def test_mldist_kernel_against_the_oracle(ctx, flags):
    """pgm_mldist_batch vs oracle/pgm_oracle.c pgmo_mldist (DistanceFactoryML.h:66-190) on a random reversible 20-state model in
    eigen form and pair counts drawn from P(d) at distances 0.02 .. 3 (incl. identical pairs, empty pairs, saturated pairs):
    1e-12 relative (device exp / log differ from glibc's in the last bit); the p-distance branch (no -m / -M) bit for bit."""
    import oracle_lib
    import prographmsa_amd as pg
    rng = np.random.default_rng(77)
    D = 20
    pi = rng.dirichlet(np.ones(D) * 5)
    S = rng.gamma(0.5, 1.0, (D, D)); S = (S + S.T) / 2; np.fill_diagonal(S, 0)
    Q = S * pi[None, :]
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -(pi * np.diag(Q)).sum()
    sig, V = np.linalg.eig(Q)
    sig, V = sig.real, V.real
    Vi = np.linalg.inv(V)
    npairs = 300
    counts = np.zeros((npairs, D, D), np.int32)
    gaps = rng.integers(0, 30, npairs).astype(np.uint32)
    seqlen = rng.uniform(50, 1200, npairs)
    for p in range(npairs):
        d = float(rng.choice([0.02, 0.1, 0.3, 0.8, 1.5, 3.0]))
        Pd = (V * np.exp(sig * d)[None, :]) @ Vi
        n = int(rng.integers(0, 900))
        if p % 17 == 0:
            n = 0                                   # empty pair
        a = rng.choice(D, n, p=pi)
        if p % 13 == 0:
            b = a.copy()                            # identical sequences
        else:
            b = np.array([rng.choice(D, p=np.clip(Pd[x], 0, None) / np.clip(Pd[x], 0, None).sum()) for x in a], int) if n else a
        for x, y in zip(a, b):
            counts[p, y, x] += 1                    # counts(c1, c2) column-major
    m = pg.pgm_mldist_model()
    keep = [np.asfortranarray(Q), np.asfortranarray(V), np.asfortranarray(Vi), np.ascontiguousarray(sig)]
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    m.dim = D
    m.Q, m.V, m.Vi, m.sigma = P(keep[0], C.c_double), P(keep[1], C.c_double), P(keep[2], C.c_double), P(keep[3], C.c_double)
    m.dist_max, m.var_max, m.var_min, m.cutoff_dist, m.min_dist, m.max_dist, m.indel_rate = 2.2, 1e3, 1e-5, 2.2, 0.05, 2.2, 0.0093359375
    m.mldist, m.mldist_gap = flags
    cflat = np.ascontiguousarray(counts.reshape(-1))
    dist = np.zeros(npairs); var = np.zeros(npairs)
    pg.check(pg.lib.pgm_mldist_batch(ctx.handle, C.byref(m), npairs, P(cflat, C.c_int32), P(gaps, C.c_uint32), P(seqlen, C.c_double), P(dist, C.c_double), P(var, C.c_double)))
    rd, rv = oracle_lib.mldist(m, cflat, gaps, seqlen)
    if flags == (0, 0):
        assert np.array_equal(dist.view(np.uint64), rd.view(np.uint64)) and np.array_equal(var.view(np.uint64), rv.view(np.uint64))
    else:
        assert np.max(np.abs(dist - rd) / np.maximum(np.abs(rd), 1e-300)) <= 1e-12
        assert np.max(np.abs(var - rv) / np.maximum(np.abs(rv), 1e-300)) <= 1e-9   # (the variance is -1 / f'' of the last Newton step: a difference of two large sums)


@pytest.mark.parametrize("nseq,ncols", [(1, 400), (7, 400), (64, 400), (257, 400), (33, 3721)])
def test_kmer_cosine_bit_exact(ctx, nseq, ncols):
    """pgm_kmer_cosine_kernel vs the oracle's pgmo_kmer_cosine (DistanceFactoryAngle.h:100): same order of the fp64 operations, bit for bit."""
    import oracle_lib
    import prographmsa_amd as pg
    rng = np.random.default_rng(nseq * 1000 + ncols)
    counts = rng.poisson(0.6, (nseq, ncols)).astype(np.int32)
    counts[:, 0] += 1                                   # no all-zero row (a sequence shorter than K would have one: 1/0 on both sides)
    out = np.zeros(nseq * nseq)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    pg.check(pg.lib.pgm_kmer_cosine(ctx.handle, nseq, ncols, P(counts, C.c_int32), P(out, C.c_double)))
    ref = oracle_lib.kmer_cosine(counts)
    assert np.array_equal(out.view(np.uint64), ref.view(np.uint64))
