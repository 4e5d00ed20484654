"""Soak: the headline batch N times; every launch must reproduce the first launch's scores and mappings exactly (GPU box)."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
dump = os.path.join(tmp, "jobs.bin")
subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True)
jobs = J.load_jobs(dump)
ctx = pg.Context(0)
b = J.Batch(ctx, jobs)
def snapshot():
    b.run(); b.fetch_raw()
    o = b.cj.out
    return (np.array([o[i].score for i in range(b.cj.n)], np.float32).view(np.uint32), np.array([o[i].len for i in range(b.cj.n)]),
            np.concatenate([b.cj.maps1[i][:o[i].len] for i in range(b.cj.n)]), np.concatenate([b.cj.maps2[i][:o[i].len] for i in range(b.cj.n)]))
ref = snapshot()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
bad = 0
t0 = time.time()
for k in range(n):
    cur = snapshot()
    if not all(np.array_equal(a, c) for a, c in zip(ref, cur)):
        bad += 1
        print("launch %d differs" % k, flush=True)
    if (k + 1) % 250 == 0:
        print("%d launches, %d differing, %.1f s" % (k + 1, bad, time.time() - t0), flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", n, "launches")
