"""Cycles per lock-step tick of the root job's bands, alone and inside the full 255-job batch (instrumented fill kernel):
tells contention (more cycles per tick) from clock throttling (same cycles, longer wall time)."""
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
ri = int(np.argmax([j.cells for j in jobs]))
for name, js, idx in (("root alone", [jobs[ri]], 0), ("root inside the 255-job batch", jobs, ri)):
    b = J.Batch(ctx, js)
    b.run(); b.fetch()
    os.environ["PGM_FILL_DBG"] = "8"
    p, e, f, t = b.time(1)
    res = b.fetch()
    os.environ["PGM_FILL_DBG"] = "0"
    nb = (js[idx].g1.n - 1 + 47) // 48
    m = res[idx]["map1"][:8 * nb].reshape(nb, 8)
    print("%s: fill kernel %.3f ms (no tracebacks); main wavefront cycles/tick: mean %.0f (wait %.0f), first/last band %d / %d" % (
        name, f, m[:, 1].mean(), m[:, 0].mean(), m[0, 1], m[-1, 1]), flush=True)
    b.close()
