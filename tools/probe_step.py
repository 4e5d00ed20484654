"""Where a step of the headline batch spends its wall time (GPU box): pgm_align_batch_run (launches) and pgm_align_batch_fetch
(poll + copy-out) separately, over 200 steps; several processes in a row show the run-to-run spread of a box."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import numpy as np
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
tr, tf = [], []
for _ in range(200):
    t0 = time.perf_counter(); b.run(); t1 = time.perf_counter(); b.fetch_raw(); t2 = time.perf_counter()
    tr.append((t1 - t0) * 1e3); tf.append((t2 - t1) * 1e3)
tr, tf = np.array(tr), np.array(tf)
a, e, f, c = b.time(5)
print("step %.3f ms = run call %.3f (min %.3f max %.3f) + fetch %.3f (min %.3f max %.3f); kernels by events: prep %.3f emission %.3f fill %.3f" % (
    (tr + tf).mean(), tr.mean(), tr.min(), tr.max(), tf.mean(), tf.min(), tf.max(), a, e, f), flush=True)
