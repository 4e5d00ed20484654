"""Cycle breakdown of the lock-step fill (PGM_FILL_DBG=8 build variant) for one job of each tree level, alone (GPU box):
per band [main wait, main total, h1 wait, h1 total, h2 wait, h2 total, h3 wait, h3 total] in cycles per step."""
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
for name, k in (("level 2", 130), ("level 3", 200), ("level 4", 230), ("level 5", 244), ("level 6", 250), ("level 7", 253), ("root", 254)):
    j = by[k]
    b = J.Batch(ctx, [j])
    b.run(); b.fetch()
    os.environ["PGM_FILL_DBG"] = "8"
    f = b.time(1)[2]
    res = b.fetch()
    os.environ["PGM_FILL_DBG"] = "0"
    nb = (j.g1.n - 1 + 47) // 48
    m = res[0]["map1"][:8 * nb].reshape(nb, 8).astype(float)
    ex1 = j.g1.e_col.size - (j.g1.n - 1); ex2 = j.g2.e_col.size - (j.g2.n - 1)
    print("%-8s %4d x %4d  extra edges %4d / %4d  fill %.3f ms  mean cycles/step: main wait %4.0f total %4.0f | h1 busy %4.0f | h2 busy %4.0f | h3 busy %4.0f"
          % (name, j.g1.n, j.g2.n, ex1, ex2, f, m[:, 0].mean(), m[:, 1].mean(), (m[:, 3] - m[:, 2]).mean(), (m[:, 5] - m[:, 4]).mean(), (m[:, 7] - m[:, 6]).mean()), flush=True)
