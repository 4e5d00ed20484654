"""Cycle breakdown of the helper fill kernel (PGM_FILL_DBG=8 build variant) on the root job (GPU box)."""
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
b = J.Batch(ctx, by[-1:])
b.run(); b.fetch()
os.environ["PGM_FILL_DBG"] = "8"
p, e, f, t = b.time(1)
res = b.fetch()
os.environ["PGM_FILL_DBG"] = "0"
nb = (by[-1].g1.n - 1 + 47) // 48
m = res[0]["map1"][:8 * nb].reshape(nb, 8)
print("root job fill=%.3f ms; cycles/step per band: [main wait, main total, h1 wait, h1 total, h2 wait, h2 total, h3 wait, h3 total]" % f)
print(m[::3].T, flush=True)
