"""Parity probe at larger sizes (GPU): random jobs of given sizes vs the oracle; prints the first difference per job."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
import oracle_lib

ctx = pg.Context(0)
cases = [
    ("chain 1002x1002 x1", [dict(n1=1002, n2=1002, skip_frac=0.0, drop_chain_frac=0.0)]),
    ("chain 1002x1002 x8", [dict(n1=1002, n2=1002, skip_frac=0.0, drop_chain_frac=0.0)] * 8),
    ("light 1030x1020 x8", [dict(n1=1030, n2=1020, skip_frac=0.03, skip_span=4)] * 8),
    ("heavy 2100x2150 x2", [dict(n1=2100, n2=2150, skip_frac=0.5, skip_span=14, skip_max=3)] * 2),
    ("chain 300x2000 x64", [dict(n1=300, n2=2000, skip_frac=0.0, drop_chain_frac=0.0)] * 64),
    # heavy-tailed graphs at the size of the roots of configs 4 / 5 and beyond (MODE 2 with long / remote entries, overflow table)
    ("tails 3700x3600 x1", [dict(n1=3700, n2=3600, skip_frac=0.3, skip_span=150, skip_max=5)]),
    ("dense 2500x2600 x2", [dict(n1=2500, n2=2600, skip_frac=0.5, skip_span=27, skip_max=14)] * 2),
    ("mixed 6000x5000 x1", [dict(n1=6000, n2=5000, skip_frac=0.25, skip_span=90, skip_max=6, repeat_frac=0.03, repeat_span=120)]),
]
for name, specs in cases:
    js = [J.random_job(500 + i, sp["n1"], sp["n2"], **{k: v for k, v in sp.items() if k not in ("n1", "n2")}) for i, sp in enumerate(specs)]
    b = J.Batch(ctx, js, keep_matrices=True)
    b.run()
    res = b.fetch()
    bad = 0
    for i, j in enumerate(js):
        ref = oracle_lib.align_graphs(j, want_matrices=True)
        mats = b.read_matrices(i)
        n1, n2 = j.g1.n, j.g2.n
        for k, nm in enumerate("MXYW"):
            a = mats[k][: n1 - 1, : n2 - 1].view(np.uint32); r = ref["mats"][k][: n1 - 1, : n2 - 1].view(np.uint32)
            if not (a == r).all():
                w = np.argwhere(a != r)
                print("  %s job %d: matrix %s differs in %d cells, first %s, rows %d..%d cols %d..%d" % (name, i, nm, len(w), w[0], w[:, 0].min(), w[:, 0].max(), w[:, 1].min(), w[:, 1].max()))
                bad += 1
                break
        else:
            ok = res[i]["status"] == 0 and np.array_equal(res[i]["map1"], ref["map1"]) and np.array_equal(res[i]["map2"], ref["map2"])
            if not ok:
                print("  %s job %d: matrices equal, traceback differs (status %d, len %d vs %d)" % (name, i, res[i]["status"], len(res[i]["map1"]), len(ref["map1"])))
                bad += 1
    print("%s: %d of %d jobs differ" % (name, bad, len(js)), flush=True)
    b.close()
