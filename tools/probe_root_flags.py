"""Root job of the headline batch alone under the strip-down switches of the tools build (PGM_TEST_NOSTORE bits): where a MODE 2 step's time goes.
Usage (GPU box): PGM_TOOLS_LIB=1 python tools/probe_root_flags.py [flags ...]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
dump = os.path.join(tmp, "jobs.bin")
subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True)
jobs = sorted(J.load_jobs(dump), key=lambda j: j.cells)
root = jobs[-1:]
ctx = pg.Context(0)
os.environ["PGM_FILL_DBG"] = "8"   # the fill alone, no traceback
for fl in (sys.argv[1:] or ["0", "2", "6", "14", "78", "79", "207"]):
    os.environ["PGM_TEST_NOSTORE"] = fl
    b = J.Batch(ctx, root)
    b.run(); b.fetch_raw()
    t = sorted(b.time(1)[2] for _ in range(5))[2]
    nb = (root[0].g1.n - 1 + 63) // 64
    steps = (nb - 1) * 78 + root[0].g2.n - 1 + 63
    print("flags %5s: fill %.3f ms = %.3f us per step of the chain (%d steps)" % (fl, t, t * 1e3 / steps, steps), flush=True)
    b.close()
