"""Fill time of the leaf level (128 chain-only jobs) and of the whole batch against the number of workers (GPU box)."""
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
jobs = J.load_jobs(dump)
ctx = pg.Context(0)
by = sorted(jobs, key=lambda j: j.cells)
for name, js in (("leaf128", by[:128]), ("all", jobs)):
    for w in sys.argv[1:]:
        os.environ["PGM_FILL_WORKERS"] = w
        b = J.Batch(ctx, js)
        b.run(); b.fetch()
        p, e, f, t = b.time(3)
        print("workers<=%s %-8s fill=%8.3f ms tb=%.3f" % (w, name, f, t), flush=True)
        b.close()
