"""Chain-only ("pack") items of the fill kernel: cycles per step alone and under load, and the throughput of the 128
leaf-level jobs of the headline batch alone (GPU box).  PGM_FILL_DBG=8 selects the cycle-counting kernel variant."""
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
leaves = by[:128]
levels = {"1 leaf job": leaves[:1], "128 leaf jobs": leaves, "64 level-2 jobs": by[128:192], "all but the 3 largest": by[:-3], "all 255": by}
for name, sel in levels.items():
    b = J.Batch(ctx, sel)
    b.run(); b.fetch()
    ts = sorted(b.time(1)[2] for _ in range(5))
    cells = sum(j.cells for j in sel)
    os.environ["PGM_FILL_DBG"] = "8"
    b.time(1)
    res = b.fetch()
    os.environ["PGM_FILL_DBG"] = "0"
    cyc = []
    for j, r in zip(sel, res):
        nb = (j.g1.n - 1 + 47) // 48
        m = r["map1"][:8 * nb].reshape(nb, 8)
        cyc.append(m[:, :2])
    cyc = np.concatenate(cyc)
    print("%-24s cells %.3e  fill+traceback %.3f ms (median of 5)  %.1f GCUPS   sweeper cycles/step: wait %.0f total %.0f (mean over bands)"
          % (name, cells, ts[2], cells / ts[2] / 1e6, cyc[:, 0].mean(), cyc[:, 1].mean()), flush=True)
