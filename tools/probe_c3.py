"""The 255 alignGraphs jobs of the headline pass (dumped by the oracle driver on the CPU) through the GPU in one batch,
each compared with the oracle.  Usage: probe_c3.py [level-batched]"""
import sys, os, subprocess, tempfile
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
ctx = pg.Context(0)
b = J.Batch(ctx, jobs, keep_matrices=True)
for rep in range(3):
    b.run()
    res = b.fetch()
    bad = []
    for i, (j, r) in enumerate(zip(jobs, res)):
        ref = oracle_lib.align_graphs(j)
        if not (r["status"] == 0 and np.array_equal(r["map1"], ref["map1"]) and np.array_equal(r["map2"], ref["map2"])):
            bad.append(i)
    print("rep %d: %d of %d jobs differ: %s" % (rep, len(bad), len(jobs), bad[:20]), flush=True)
    for i in bad[:2]:
        j = jobs[i]
        ref = oracle_lib.align_graphs(j, want_matrices=True)
        mats = b.read_matrices(i)
        n1, n2 = j.g1.n, j.g2.n
        print("  job %d: %d x %d, status %d" % (i, n1, n2, res[i]["status"]))
        for k, nm in enumerate("MXYWS"):
            a = mats[k][1: n1 - 1, 1: n2 - 1].view(np.uint32); r = ref["mats"][k][1: n1 - 1, 1: n2 - 1].view(np.uint32)
            if not (a == r).all():
                w = np.argwhere(a != r) + 1
                print("    matrix %s differs in %d cells, first %s, rows %d..%d cols %d..%d" % (nm, len(w), w[0], w[:, 0].min(), w[:, 0].max(), w[:, 1].min(), w[:, 1].max()))
                y, x = w[0]
                print("      gpu %r oracle %r" % (mats[k][y, x], ref["mats"][k][y, x]))
b.close()
