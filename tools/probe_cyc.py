"""Cycle breakdown per block of the chain-only fill kernel (PGM_FILL_DBG=4 build variant) on the leaf level (GPU box)."""
import os, subprocess, sys, tempfile
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
by = sorted(jobs, key=lambda j: j.cells)
for n, w in ((1, 21), (16, 16), (32, 32), (128, 128), (128, 1024)):
    os.environ["PGM_FILL_WORKERS"] = str(w)
    b = J.Batch(ctx, by[:n])
    os.environ["PGM_FILL_DBG"] = "0"
    b.run(); b.fetch()
    os.environ["PGM_FILL_DBG"] = "4"
    p, e, f, t = b.time(1)
    res = b.fetch()
    os.environ["PGM_FILL_DBG"] = "0"
    m = res[n // 2]["map1"][:84].reshape(21, 4)
    print("jobs=%d workers<=%d fill=%.3f ms; job %d: cycles/block [start-section, of which wait_prev, 8 steps, publish] per band:" % (n, w, f, n // 2))
    print(m.T, flush=True)
    b.close()
