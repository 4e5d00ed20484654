"""Randomised parity campaign (GPU box): batches of random jobs — random sizes, edge densities, edge lengths, repeat edges,
dropped chain edges, alphabet sizes — through the HIP path and the oracle; scores, mappings, n_tr_indels and (for a sample)
the four DP matrices must be bit-identical.  usage: tools/fuzz_align.py SECONDS [SEED]   (FUZZ_CRIT=1: biased towards pgm_crit_kernel / wide bands)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
import oracle_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
ctx = pg.Context(0)
pool = ThreadPoolExecutor(8)       # the oracle is a C library: ctypes releases the GIL
t_end = time.time() + budget
njobs = nbad = ncells = it = 0
while time.time() < t_end:
    it += 1
    kw = dict(skip_frac=float(rng.choice([0.0, 0.05, 0.2, 0.5, 0.95])), skip_span=int(rng.choice([2, 3, 9, 27, 28, 29, 60, 150, 400])),
              skip_max=int(rng.choice([1, 2, 5, 9, 14])), repeat_frac=float(rng.choice([0.0, 0.0, 0.03, 0.1])),
              repeat_span=int(rng.choice([10, 90, 300])), drop_chain_frac=float(rng.choice([0.0, 0.03, 0.2])),
              onehot_frac=float(rng.choice([0.0, 0.5, 1.0])))
    dim = int(rng.choice([20, 20, 20, 61, 4]))
    big = rng.random() < 0.15
    if os.environ.get("FUZZ_CRIT"):   # bias towards the jobs of pgm_crit_kernel and of the wide band workers: many bands or a 32-step history, no long edges, no dropped chain edges
        kw.update(skip_span=int(rng.choice([2, 3, 9, 14, 20, 27])), skip_max=int(rng.choice([1, 2, 3, 5])), drop_chain_frac=0.0,
                  skip_frac=float(rng.choice([0.02, 0.05, 0.2, 0.5])), repeat_span=int(rng.choice([10, 20])))
        big = rng.random() < 0.6
    sizes = [(int(rng.integers(2, 2600 if big else 700)), int(rng.integers(2, 2600 if big else 700))) for _ in range(3 if big else int(rng.integers(1, 24)))]
    js = [J.random_job(int(rng.integers(1 << 30)), n1, n2, dim=dim, **kw) for n1, n2 in sizes]
    b = J.Batch(ctx, js, keep_matrices=True)
    b.run()
    res = b.fetch()
    want = [k == 0 for k in range(len(js))]          # matrices of the first job of every batch
    refs = list(pool.map(lambda a: oracle_lib.align_graphs(a[0], want_matrices=a[1]), zip(js, want)))
    for i, (j, r, ref) in enumerate(zip(js, res, refs)):
        ok = (r["status"] == ref["status"] and np.float32(r["score"]).view(np.uint32) == np.float32(ref["score"]).view(np.uint32)
              and r["n_tr_indels"] == ref["n_tr_indels"] and np.array_equal(r["map1"], ref["map1"]) and np.array_equal(r["map2"], ref["map2"]))
        if ok and want[i]:
            mats = b.read_matrices(i)
            for k in range(4):
                a = mats[k][: j.g1.n - 1, : j.g2.n - 1].view(np.uint32); c = ref["mats"][k][: j.g1.n - 1, : j.g2.n - 1].view(np.uint32)
                ok = ok and bool((a == c).all())
        if not ok:
            nbad += 1
            print("DIFF iteration %d job %d: %dx%d dim %d %r" % (it, i, j.g1.n, j.g2.n, dim, kw), flush=True)
        njobs += 1
        ncells += (j.g1.n - 2) * (j.g2.n - 2)
    b.close()
    if it % 20 == 0:
        print("... %d batches, %d jobs, %.2e cells, %d differing" % (it, njobs, ncells, nbad), flush=True)
print("FUZZ %s: %d batches, %d jobs, %.3e cells, %d differing (seed %d)" % ("OK" if nbad == 0 else "FAILED", it, njobs, ncells, nbad, seed0))
sys.exit(0 if nbad == 0 else 1)
