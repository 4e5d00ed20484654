"""Which job does the headline batch wait for?  Per-job timeline of one launch (pgm_align_batch_job_times, release library): the
moment each job's last band was complete and the moment its traceback was published, relative to the end of the launch (GPU box)."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
tmp = tempfile.mkdtemp()
cfg = os.environ.get("PROBE_CFG", "c3")   # c3: headline batch; c4 / c5: the heavy-tailed configs
fam, flags = {"c3": (lambda: gen.gen(256, 1000, 3), ["--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree")]),
              "c4": (lambda: gen.gen_codon(128, 1000, 4), ["--codon", "--fasta", "-t", os.path.join(ROOT, "tests/golden/c4.tree")]),
              "c5": (lambda: gen.gen(1024, 600, 6), ["--fasta", "-t", os.path.join(ROOT, "tests/golden/c5.tree")])}[cfg]
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(fam()))
dump = os.environ.get("PROBE_DUMP", os.path.join(tmp, "jobs.bin"))   # (PROBE_DUMP: made once, by a library whose results are valid — ab_jobtimes.sh)
if not os.path.exists(dump):
    subprocess.run([pg.PGMSA_PATH] + flags + ["--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True, env=dict(os.environ, PGM_HOST_PROFILE="1" if os.environ.get("PROBE_LISTS") else "0"))
jobs = J.load_jobs(dump)
if os.environ.get("PROBE_TOP"):    # the largest jobs alone (7: the MODE 2 jobs; 127: everything but the leaf level)
    jobs = sorted(jobs, key=lambda j: -j.cells)[:int(os.environ["PROBE_TOP"])]
if os.environ.get("PROBE_SET"):    # the root with one kind of company: "lean" (the leaf level), "mid" (the jobs below 1216 rows), "big" (the other six)
    srt = sorted(jobs, key=lambda j: -j.cells)
    jobs = srt[:1] + {"lean": srt[127:], "mid": srt[7:127], "big": srt[1:7], "midlean": srt[7:]}[os.environ["PROBE_SET"]]
ctx = pg.Context(0)
b = J.Batch(ctx, jobs)
for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    b.run(); b.fetch_raw()
    ms = b.stage_times()
    t = np.zeros(2 * len(jobs), np.uint64)
    pg.check(pg.lib.pgm_align_batch_job_times(ctx.handle, b.handle, t.ctypes.data_as(C.POINTER(C.c_uint64))))
    t = t.reshape(-1, 2).astype(np.float64) / 100.0          # us
    end = t.max()
    fill_us = ms[2] * 1e3
    sw, dn = fill_us - (end - t[:, 0]), fill_us - (end - t[:, 1])   # relative to the start of the fill stage
    chain = np.array([j.g1.e_col.size == j.g1.n - 1 and j.g2.e_col.size == j.g2.n - 1 for j in jobs])
    n1 = np.array([j.g1.n for j in jobs])
    print("launch %d: fill stage %.0f us" % (rep, fill_us))
    for name, m in (("chain-only (lean kernel)", chain), ("< 1216 rows", ~chain & (n1 < 1217)), (">= 1216 rows", ~chain & (n1 >= 1217))):
        m = np.asarray(m)
        if m.any():
            print("  %-26s %3d jobs: last sweep ends %5.0f us (median %5.0f), last traceback published %5.0f us (median %5.0f)" % (name, m.sum(), sw[m].max(), np.median(sw[m]), dn[m].max(), np.median(dn[m])))
    for i in np.argsort(-dn)[:6]:
        print("    job %3d (%d x %d): sweeps end %5.0f us, published %5.0f us" % (i, jobs[i].g1.n, jobs[i].g2.n, sw[i], dn[i]))
    if os.environ.get("PROBE_ALL"):   # every job that is not chain-only, largest first
        for i in sorted(np.nonzero(~chain)[0], key=lambda i: -jobs[i].cells):
            print("    job %3d (%d x %d): sweeps end %5.0f us, published %5.0f us" % (i, jobs[i].g1.n, jobs[i].g2.n, sw[i], dn[i]))
