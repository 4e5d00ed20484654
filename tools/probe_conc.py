"""Fill time of n chain-only leaf jobs run by exactly n workers (one worker per job): per-step time against concurrency,
for the single-wavefront kernel and (PGM_FILL_HELPERS=2) the helper kernel's main wavefront alone."""
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
leaf = [j for j in jobs if j.g1.e_col.size == j.g1.n - 1 and j.g2.e_col.size == j.g2.n - 1 and j.g1.r_col is None and j.g2.r_col is None]
print("chain-only jobs:", len(leaf))
for h in ("1", "2"):
    os.environ["PGM_FILL_HELPERS"] = h
    for n, w in ((1, 1), (16, 16), (32, 32), (128, 128), (128, 512), (128, 1024)):
        os.environ["PGM_FILL_WORKERS"] = str(w)
        js = leaf[:n]
        b = J.Batch(ctx, js)
        b.run(); b.fetch()
        p, e, f, t = b.time(3)
        steps = max(((j.g1.n - 1 + 47) // 48) * (j.g2.n - 1 + 63) for j in js)
        print("helpers=%s n=%3d workers<=%4d fill=%8.3f ms  (%.3f us/step if one worker per job)" % (h, n, w, f, f * 1e3 / steps), flush=True)
        b.close()
