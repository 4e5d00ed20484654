import sys, os, subprocess, tempfile, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
import oracle_lib
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "c3.fa")
open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
dump = os.path.join(tmp, "jobs.bin")
subprocess.run([oracle_lib.PGMSA_ORACLE, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True)
jobs = J.load_jobs(dump)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which == "chain":
    jobs = jobs[:128]
elif which == "merged":
    jobs = jobs[128:]
ctx = pg.Context(0)
b = J.Batch(ctx, jobs, keep_matrices=True)
b.run()
res = b.fetch()
bad = [i for i, (j, r) in enumerate(zip(jobs, res)) if r["status"] != 0]
print(which, "bad jobs:", len(bad), bad[:10], flush=True)
for i in bad[:2]:
    j = jobs[i]
    ref = oracle_lib.align_graphs(j, want_matrices=True)
    mats = b.read_matrices(i)
    n1, n2 = j.g1.n, j.g2.n
    diff = np.zeros((n1 - 1, n2 - 1), bool)
    for k in range(4):
        diff |= mats[k][: n1 - 1, : n2 - 1].view(np.uint32) != ref["mats"][k][: n1 - 1, : n2 - 1].view(np.uint32)
    w = np.argwhere(diff)
    print(" job", i, n1, n2, "bad cells", len(w))
    bands = collections.Counter((w[:, 0] // 64).tolist())
    print("  bands:", sorted(bands.items()))
    t = w[:, 0] % 64 + w[:, 1]
    print("  t%8:", sorted(collections.Counter((t % 8).tolist()).items()))
    print("  lanes:", sorted(collections.Counter((w[:, 0] % 64).tolist()).items())[:70])
    # the earliest bad cell in sweep order within the first bad band
    b0 = min(bands)
    wb = w[w[:, 0] // 64 == b0]
    tb = wb[:, 0] % 64 + wb[:, 1]
    o = np.argsort(tb, kind="stable")
    for q in o[:12]:
        y, x = wb[q]
        print("   band %d step %d lane %d (y %d x %d): gpu M %r X %r Y %r W %r | ref M %r X %r Y %r W %r | bits M %d" % (
            b0, tb[q], y % 64, y, x, mats[0][y, x], mats[1][y, x], mats[2][y, x], mats[3][y, x],
            ref["mats"][0][y, x], ref["mats"][1][y, x], ref["mats"][2][y, x], ref["mats"][3][y, x], int(mats[0][y, x].view(np.uint32))))
b.close()
