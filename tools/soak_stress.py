"""Soak on the synthetic stress jobs of tests/test_gpu_align.py (dense extras, far edges, repeats, heavy tails): N launches, all identical."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
kws = [dict(skip_frac=0.2), dict(skip_frac=0.3, repeat_frac=0.05), dict(skip_frac=0.95, skip_max=9, drop_chain_frac=0.0),
       dict(skip_frac=0.1, skip_span=70, repeat_frac=0.03, repeat_span=90), dict(skip_frac=0.0, drop_chain_frac=0.0),
       # heavy-tailed graphs: long / remote entries of the far helpers, overflow table, generic fall-backs
       dict(skip_frac=0.3, skip_span=150, skip_max=5), dict(skip_frac=0.5, skip_span=27, skip_max=14),
       dict(skip_frac=0.4, skip_span=60, skip_max=10, repeat_frac=0.05, repeat_span=120)]
sizes = [(2, 2), (3, 2), (7, 4), (40, 33), (64, 64), (65, 66), (130, 97), (200, 310), (517, 129), (300, 700), (900, 450)]
jobs = [J.random_job(5000 + 37 * a + i, n1, n2, **kw) for a, kw in enumerate(kws) for i, (n1, n2) in enumerate(sizes)]
ctx = pg.Context(0)
b = J.Batch(ctx, jobs)
def snapshot():
    b.run(); b.fetch_raw()
    o = b.cj.out
    return (np.array([o[i].score for i in range(b.cj.n)], np.float32).view(np.uint32), np.array([o[i].len for i in range(b.cj.n)]),
            np.array([o[i].status for i in range(b.cj.n)]),
            np.concatenate([b.cj.maps1[i][:o[i].len] for i in range(b.cj.n)]), np.concatenate([b.cj.maps2[i][:o[i].len] for i in range(b.cj.n)]))
ref = snapshot()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
bad = sum(0 if all(np.array_equal(a, c) for a, c in zip(ref, snapshot())) else 1 for _ in range(n))
print("SOAK-STRESS", "OK" if bad == 0 else "FAILED (%d)" % bad, n, "launches of", len(jobs), "jobs; statuses", np.bincount(ref[2]))
