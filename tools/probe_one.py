"""Runs the fill pipeline once on one selected job of the 256x1000 workload (for rocprofv3 --pmc)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
which = sys.argv[1] if len(sys.argv) > 1 else "root"
dump = sys.argv[2]
jobs = J.load_jobs(dump)
by = sorted(jobs, key=lambda j: j.cells)
sel = {"root": by[-1:], "leaf": by[:1], "mid": [by[200]], "all": jobs}[which]
ctx = pg.Context(0)
b = J.Batch(ctx, sel)
b.run(); b.fetch()
print(which, "cells", sum(j.cells for j in sel))
