"""Randomised parity campaign of the all-pairs alignPair kernel (GPU box): random families (lengths 0..1600, mutation and indel
rates, 20 / 61 states, random symmetric score tables and gap penalties) through pgm_nw_pairs_batch and the oracle; the pair
counts and gap counts must be identical.  usage: tools/fuzz_nw.py SECONDS [SEED]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prographmsa_amd as pg
import oracle_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
ctx = pg.Context(0)
P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
t_end = time.time() + budget
it = npairs_total = nbad = 0
cells = 0
while time.time() < t_end:
    it += 1
    dim = int(rng.choice([20, 20, 61]))
    nseq = int(rng.integers(2, 14))
    Lmax = int(rng.choice([12, 70, 300, 600, 1600]))
    base = rng.integers(0, dim, Lmax + 8)
    mut, indel = float(rng.choice([0.0, 0.1, 0.4, 1.0])), float(rng.choice([0.0, 0.02, 0.1]))
    seqs = []
    for _ in range(nseq):
        L = int(rng.integers(0, Lmax + 1))
        s = base[:L].copy()
        m = rng.random(L) < mut
        s[m] = rng.integers(0, dim + 1, int(m.sum()))
        keep = rng.random(L) >= indel
        seqs.append(np.minimum(s[keep], dim).astype(np.int8))
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint32)
    syms = np.ascontiguousarray(np.concatenate(seqs) if offs[-1] else np.zeros(1, np.int8), np.int8)
    pairs = [(i, j) for i in range(nseq) for j in range(nseq) if i != j]
    pi = np.array([p[0] for p in pairs], np.uint32); pj = np.array([p[1] for p in pairs], np.uint32)
    sc = rng.integers(-6, 4, (dim + 1, dim + 1)).astype(np.int32)
    sc = np.minimum(sc, sc.T)
    sc[np.arange(dim + 1), np.arange(dim + 1)] = rng.integers(1, 14, dim + 1)
    sc = np.ascontiguousarray(sc.reshape(-1))
    go, ge = -int(rng.integers(1, 16)), -int(rng.integers(0, 4))
    n = len(pairs)
    counts = np.full(n * dim * dim, -1, np.int32); gaps = np.zeros(n, np.uint32)
    pg.check(pg.lib.pgm_nw_pairs_batch(ctx.handle, dim, P(sc, C.c_int32), go, ge, nseq, P(syms, C.c_int8), P(offs, C.c_uint32), n,
                                       P(pi, C.c_uint32), P(pj, C.c_uint32), P(counts, C.c_int32), P(gaps, C.c_uint32)))
    co, gr = oracle_lib.nw_pairs(dim, sc, go, ge, syms, offs, pi, pj)
    same = np.array_equal(counts.reshape(n, dim * dim), co) and np.array_equal(gaps, gr)
    if not same:
        nbad += 1
        print("DIFF iteration %d: dim %d nseq %d Lmax %d mut %.2f indel %.2f go %d ge %d" % (it, dim, nseq, Lmax, mut, indel, go, ge), flush=True)
    npairs_total += n
    cells += sum(len(seqs[i]) * len(seqs[j]) for i, j in pairs)
    if it % 50 == 0:
        print("... %d families, %d pairs, %.2e cells, %d differing" % (it, npairs_total, cells, nbad), flush=True)
print("FUZZ-NW %s: %d families, %d pairs, %.3e cells, %d differing (seed %d)" % ("OK" if nbad == 0 else "FAILED", it, npairs_total, cells, nbad, seed0))
sys.exit(0 if nbad == 0 else 1)
