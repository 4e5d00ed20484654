"""Times the fill kernel on selected jobs of the 256x1000 workload (GPU box)."""
import json, os, subprocess, sys, tempfile
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
def run(name, js, reps=3):
    b = J.Batch(ctx, js)
    b.run(); b.fetch()
    p, e, f, t = b.time(reps)
    cells = sum(j.cells for j in js)
    nb = [(j.g1.n - 1 + 47) // 48 for j in js]
    ts = [j.g2.n - 1 + 63 for j in js]
    wsteps = sum(a * c for a, c in zip(nb, ts))
    print("%-28s jobs=%3d cells=%.3e fill=%8.3f ms  %.2f GCUPS  wave-steps=%.3e  max job steps(5 waves)=%d  prep=%.3f emis=%.3f tb=%.3f" % (
        name, len(js), cells, f, cells / f / 1e6, wsteps, max(-(-a // 5) * c for a, c in zip(nb, ts)), p, e, t), flush=True)
    b.close()
run("1 leaf job", by[:1])
run("1 median job", [by[len(by) // 2]])
run("root job", by[-1:])
run("2nd largest", by[-2:-1])
run("job #200 (mid tree)", [by[200]])
run("128 smallest (leaf level)", by[:128])
run("all 255", jobs)
