"""Times the fill kernel on the root job of the 256x1000 workload (GPU box); PGM_FILL_DBG selects experiment variants."""
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
for name, js in (("root", by[-1:]), ("leaf128", by[:128]), ("all", jobs)):
    b = J.Batch(ctx, js)
    for d in sys.argv[1:]:
        os.environ["PGM_FILL_DBG"] = d    # experiment variants compute wrong cells; the library skips the traceback for them
        p, e, f, t = b.time(3)
        print("DBG=%s %-5s fill=%8.3f ms tb=%.3f" % (d, name, f, t), flush=True)
        if d == "4" and name == "leaf128":
            res = b.fetch()
            for jj in (0, 64, 127):
                print(" job", jj, "per band [start-section, wait_prev, 8 steps, publish] cycles/block:")
                print(res[jj]["map1"][:84].reshape(21, 4).T)
    os.environ["PGM_FILL_DBG"] = "0"
    b.close()
