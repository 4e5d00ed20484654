"""Host-side cost of the staged API on the headline batch (GPU box): create (flatten + upload), run, fetch, destroy."""
import os, subprocess, sys, tempfile, time
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
cj = J.CJobs(jobs)
import ctypes as C
for rep in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter()
    pg.check(pg.lib.pgm_align_batch_create(ctx.handle, cj.n, cj.g1, cj.g2, cj.m, cj.sc, C.byref(h)))
    t1 = time.perf_counter()
    pg.check(pg.lib.pgm_align_batch_run(ctx.handle, h))
    rc = pg.lib.pgm_align_batch_fetch(ctx.handle, h, cj.out)
    t2 = time.perf_counter()
    pg.lib.pgm_align_batch_destroy(ctx.handle, h)
    t3 = time.perf_counter()
    print("rep %d: create %.1f ms, run+fetch %.1f ms, destroy %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
