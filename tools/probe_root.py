"""Root job and second-largest job of the headline batch alone, release library: fill (sweep + traceback) time, median of 7 (GPU box)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
dump = os.environ.get("PROBE_DUMP", os.path.join(tmp, "jobs.bin"))   # (PROBE_DUMP: made once, by a library whose results are valid)
if not os.path.exists(dump):
    subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True)
jobs = sorted(J.load_jobs(dump), key=lambda j: j.cells)
ctx = pg.Context(0)
for name, js in (("root", jobs[-1:]), ("2nd", jobs[-2:-1]), ("level 6", jobs[-7:-6]), ("all", jobs)):
    b = J.Batch(ctx, js)
    b.run()
    if not os.environ.get("PROBE_NOFETCH"):
        b.fetch_raw()
    ts = sorted(b.time(1)[2] for _ in range(7))
    print("%-8s fill %.3f ms (min %.3f)" % (name, ts[3], ts[0]), flush=True)
    b.close()
