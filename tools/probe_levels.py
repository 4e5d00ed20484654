"""Fill pipeline time of each guide-tree level of the 256 x 1000 workload as its own batch (what the product driver issues),
for different caps of persistent workers (GPU box)."""
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
jobs = J.load_jobs(dump)   # dumped level by level: 128, 64, 32, ... jobs
ctx = pg.Context(0)
levels, i, n = [], 0, 128
while n >= 1:
    levels.append(jobs[i:i + n]); i += n; n //= 2
for w in sys.argv[1:]:
    os.environ["PGM_FILL_WORKERS"] = w
    tot = 0.0
    line = []
    for js in levels:
        b = J.Batch(ctx, js)
        b.run(); b.fetch()
        p, e, f, t = b.time(3)
        line.append("%d:%.2f" % (len(js), p + e + f)); tot += p + e + f
        b.close()
    print("workers<=%s  kernels per level (jobs:ms) %s  total %.2f ms" % (w, " ".join(line), tot), flush=True)
