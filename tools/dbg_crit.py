"""Debug: one family of tests/test_gpu_align.py through the HIP path and the oracle, every differing cell counted (GPU box).
Usage: python tools/dbg_crit.py FAMILY_INDEX [DIM] [REPEATS]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
import test_gpu_align as T
k = int(sys.argv[1]); dim = int(sys.argv[2]) if len(sys.argv) > 2 else 20; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = T.FAMILIES[k]
sizes = [(2, 2), (3, 2), (2, 5), (3, 3), (7, 4), (40, 33), (64, 64), (65, 66), (66, 65), (130, 97), (200, 310), (517, 129)]
if os.environ.get("DBG_SIZES"): sizes = [tuple(int(v) for v in s.split("x")) for s in os.environ["DBG_SIZES"].split(",")]
js = [J.random_job(1000 + i, n1, n2, dim=dim, **kw) for i, (n1, n2) in enumerate(sizes)]
ctx = pg.Context(0)
refs = [oracle_lib.align_graphs(j, want_matrices=True) for j in js]
for rep in range(reps):
    b = J.Batch(ctx, js, keep_matrices=True)
    b.run(); res = b.fetch()
    for i, j in enumerate(js):
        mats = b.read_matrices(i)
        n1, n2 = j.g1.n, j.g2.n
        for m in range(4):
            a = mats[m][: n1 - 1, : n2 - 1]; r = refs[i]["mats"][m][: n1 - 1, : n2 - 1]
            bad = a.view(np.uint32) != r.view(np.uint32)
            if bad.any():
                pos = np.argwhere(bad)
                print("rep %d job %d (%dx%d) matrix %s: %d cells differ, first %s  got %r want %r; steps (x + y %% 64) of the first: %s" % (
                    rep, i, n1, n2, "MXYW"[m], bad.sum(), pos[:4].tolist(), a[bad][:4], r[bad][:4], [(int(p[1]) + int(p[0]) % 64) for p in pos[:4]]), flush=True)
    b.close()
print("done")
