"""Whole headline batch: fill time (median of 5 single launches) under the environment given (GPU box; PGM_TOOLS_LIB=1 for the switches)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
dump = os.path.join(tmp, "jobs.bin")
env = {k: v for k, v in os.environ.items() if not k.startswith("PGM_") or k in ("PGM_TOOLS_LIB",)}
subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True, env=env)
jobs = J.load_jobs(dump)
ctx = pg.Context(0)
b = J.Batch(ctx, jobs)
b.run(); b.fetch_raw()
ts = sorted(b.time(1) for _ in range(7))
print("env %s: prep %.3f emission %.3f fill %.3f ms (min %.3f)" % ({k: v for k, v in os.environ.items() if k.startswith("PGM_")}, ts[3][0], ts[3][1], sorted(t[2] for t in ts)[3], min(t[2] for t in ts)), flush=True)
