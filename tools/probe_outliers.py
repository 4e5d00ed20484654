"""Distribution of the run + fetch time of the headline batch over many launches (GPU box): looks for rare slow launches."""
import os, subprocess, sys, tempfile, time
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
b = J.Batch(ctx, jobs)
for _ in range(3):
    b.run(); b.fetch_raw()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ts = []
for _ in range(n):
    t0 = time.perf_counter(); b.run(); b.fetch_raw(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("n=%d  min %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f ms;  launches slower than 1.5 x median: %d" % (
    n, ts.min(), np.median(ts), np.quantile(ts, 0.9), np.quantile(ts, 0.99), ts.max(), int((ts > 1.5 * np.median(ts)).sum())))
print("slow ones:", np.round(ts[ts > 1.5 * np.median(ts)], 2), "at", np.where(ts > 1.5 * np.median(ts))[0])
