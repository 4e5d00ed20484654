"""Timeline of the fill kernel's band queue on the headline batch (GPU box): PGM_FILL_TRACE dump -> utilisation summary."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
tmp = tempfile.mkdtemp()
trace = os.path.join(tmp, "trace.bin")
import prographmsa_amd as pg
from prographmsa_amd import jobs as J
cfg = os.environ.get("PROBE_CFG", "c3")   # c3: headline batch; c4 / c5: the heavy-tailed configs
fam, flags = {"c3": (lambda: gen.gen(256, 1000, 3), ["--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree")]),
              "c4": (lambda: gen.gen_codon(128, 1000, 4), ["--codon", "--fasta", "-t", os.path.join(ROOT, "tests/golden/c4.tree")]),
              "c5": (lambda: gen.gen(1024, 600, 6), ["--fasta", "-t", os.path.join(ROOT, "tests/golden/c5.tree")])}[cfg]
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(fam()))
dump = os.path.join(tmp, "jobs.bin")
subprocess.run([pg.PGMSA_PATH] + flags + ["--dump_jobs", dump, "-o", os.path.join(tmp, "o.fa"), fa], check=True)
jobs = J.load_jobs(dump)
if os.environ.get("PROBE_LEAF"):   # the chain-only jobs (leaf level) alone
    jobs = sorted(jobs, key=lambda j: j.cells)[:int(os.environ["PROBE_LEAF"])]
if os.environ.get("PROBE_TOP"):    # the largest jobs alone (1: the root)
    jobs = sorted(jobs, key=lambda j: -j.cells)[:int(os.environ["PROBE_TOP"])]
os.environ["PGM_FILL_TRACE"] = trace
for kv in os.environ.get("PROBE_ENV", "").split(","):   # experiment knobs that must not reach the product run above
    if "=" in kv:
        os.environ[kv.split("=")[0]] = kv.split("=")[1]
ctx = pg.Context(0)
b = J.Batch(ctx, jobs)
b.run(); b.fetch(); b.run(); b.fetch()
buf = open(trace, "rb").read()
n = np.frombuffer(buf, np.uint32, 1)[0]
items = np.frombuffer(buf, np.uint32, 4 * n, 4).reshape(n, 4)[:, :2]
raw = np.frombuffer(buf, np.uint64, 6 * n, 4 + 16 * n).reshape(n, 6)
tr = raw[:, :4].astype(np.float64)
t0 = tr[:, 1].min()
start, bend, tend = (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0, np.where(tr[:, 3] > 0, (tr[:, 3] - t0) / 100.0, 0)   # us
end = np.maximum(bend, tend)
wait_us = (raw[:, 0] >> np.uint64(16)).astype(np.float64) / 100.0
tr[:, 0] = (raw[:, 0] & np.uint64(0xffff)).astype(np.float64)
print("items %d, workers %d, makespan %.0f us" % (n, len(set(tr[:, 0])), end.max()))
print("time spent waiting for the previous band (wavefront 0 of each item): %.0f us-worker" % wait_us.sum())
busy = (end - start).sum()
print("sum of item times %.0f us-worker = %.1f%% of workers x makespan" % (busy, 100 * busy / (len(set(tr[:, 0])) * end.max())))
tb = np.where(tend > 0, tend - bend, 0)
print("traceback: %d items, total %.0f us-worker, max %.0f us" % ((tend > 0).sum(), tb.sum(), tb.max()))
sizes = np.array([(j.g1.n, j.g2.n) for j in jobs])
for q in (0.5, 0.75, 0.9, 0.95, 0.99, 1.0):
    print("  %3.0f%% of items started by %.0f us, ended by %.0f us" % (100 * q, np.quantile(start, q), np.quantile(end, q)))
late = np.argsort(-end)[:12]
for i in late:
    j, bd = items[i]
    print("  late item %4d: job %3d (%dx%d, extras=%d) band %2d/%2d  start %.0f  band end %.0f  tb end %.0f" % (i, j, sizes[j][0], sizes[j][1], int(jobs[j].g1.e_col.size != jobs[j].g1.n - 1 or jobs[j].g2.e_col.size != jobs[j].g2.n - 1), bd, (sizes[j][0] - 1 + 63) // 64, start[i], bend[i], tend[i]))
# lean jobs (pgm_lean_kernel): second section of the dump: nlean, job ids in queue order, 6 words per job
off2 = 4 + 16 * n + 8 * 22 * n
if len(buf) >= off2 + 4:
    nl = int(np.frombuffer(buf, np.uint32, 1, off2)[0])
    lj = np.frombuffer(buf, np.uint32, nl, off2 + 4)
    lr = np.frombuffer(buf, np.uint64, 6 * nl, off2 + 4 + 4 * nl).reshape(nl, 6)
    if nl:
        ls, lb, lt = (lr[:, 1].astype(float) - t0) / 100.0, (lr[:, 2].astype(float) - t0) / 100.0, (lr[:, 3].astype(float) - t0) / 100.0
        w, v = lr[:, 5], lr[:, 4]
        wait_us = ((v >> np.uint64(40)) & np.uint64(0xfffff)).astype(float) / 100.0
        win_us = ((v >> np.uint64(20)) & np.uint64(0xfffff)).astype(float) / 100.0
        walk_us = (v & np.uint64(0xfffff)).astype(float) / 100.0
        print("lean kernel: %d jobs on %d workers, first start %.0f us, last end %.0f us; per job: wavefront 0 sweeps %.0f us, then %.0f us until every band is done, walk %.0f us (window switches %.1f us: %.1f windows, %.1f not prefetched; %.1f loop iterations), publish %.0f us; total %.0f us" % (
            nl, len(set(lr[:, 0])), ls.min(), lt.max(), (lb - ls).mean(), wait_us.mean(), walk_us.mean(), win_us.mean(), (w >> np.uint64(32)).astype(float).mean(),
            ((w >> np.uint64(16)) & np.uint64(0xffff)).astype(float).mean(), (w & np.uint64(0xffff)).astype(float).mean(), (lt - lb - wait_us - walk_us).mean(), (lt - ls).mean()))
tbi = np.where(tend > 0)[0]
rel = (raw[tbi, 4] & np.uint64(0xffffffff)).astype(np.float64) / 100.0
stg = (raw[tbi, 4] >> np.uint64(32)).astype(np.float64) / 100.0   # of which: loads -> LDS (the rest is the link pass)
nrel = (raw[tbi, 5] >> np.uint64(32)).astype(np.float64)
slow = (raw[tbi, 5] & np.uint64(0xffff)).astype(np.float64)
grid = ((raw[tbi, 5] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64)   # runs of the walker through pre-linked grid rows
plen = np.array([sizes[items[i, 0]].sum() for i in tbi], dtype=np.float64)
for name, m in (("chain-only", np.array([jobs[items[i, 0]].g1.e_col.size == jobs[items[i, 0]].g1.n - 1 and jobs[items[i, 0]].g2.e_col.size == jobs[items[i, 0]].g2.n - 1 for i in tbi])),):
    for nm, mm in ((name, m), ("merged", ~m)):
        if mm.any():
            print("traceback %-10s: %3d jobs, mean %.0f us, of which tile staging %.0f us in %.0f tiles (%.1f us/tile); slow steps %.0f of ~%.0f nodes; walking %.3f us/node" % (
                nm, mm.sum(), tb[tbi][mm].mean(), rel[mm].mean(), nrel[mm].mean(), (rel[mm] / nrel[mm]).mean(), slow[mm].mean(), plen[mm].mean(), ((tb[tbi][mm] - rel[mm]) / plen[mm]).mean()))
ri = [k for k, i in enumerate(tbi) if items[i, 0] == np.argmax(sizes[:, 0] * sizes[:, 1])][0]
print("root traceback: %.0f us, tile staging %.0f us (loads -> LDS %.0f us, links %.0f us) in %.0f tiles (%.1f us/tile), slow steps %.0f, runs through pre-linked rows %.0f" % (tb[tbi][ri], rel[ri], stg[ri], rel[ri] - stg[ri], nrel[ri], rel[ri] / max(nrel[ri], 1), slow[ri], grid[ri]))
print("all merged: loads -> LDS %.1f us/tile, links %.1f us/tile; chain-only: %.1f / %.1f" % ((stg[~m] / nrel[~m]).mean(), ((rel[~m] - stg[~m]) / nrel[~m]).mean(), (stg[m] / nrel[m]).mean(), ((rel[m] - stg[m]) / nrel[m]).mean()))
# step time per class of job: (band end - start - wait) / (steps of the item's last band + its start offset)
cnt = np.frombuffer(buf, np.uint32, 4 * n, 4).reshape(n, 4)[:, 3]
ext = np.array([int(j.g1.e_col.size != j.g1.n - 1 or j.g2.e_col.size != j.g2.n - 1) for j in jobs])
steps = np.array([sizes[items[i, 0]][1] - 1 + 63 for i in range(n)], dtype=np.float64)
per = (bend - start - wait_us) / steps
for nm, mm in (("chain-only items", ext[items[:, 0]] == 0), ("merged items", ext[items[:, 0]] == 1), ("root items", items[:, 0] == np.argmax(sizes[:, 0] * sizes[:, 1]))):
    if mm.any():
        print("%-18s: %4d items, bands/item %.2f, (band end - start - wait of wavefront 0) / steps: mean %.3f us, median %.3f, p90 %.3f; wait mean %.0f us" % (
            nm, mm.sum(), cnt[mm].mean(), per[mm].mean(), np.median(per[mm]), np.quantile(per[mm], 0.9), wait_us[mm].mean()))
# per-job: first start, last end
for name, sel in (("root", np.argmax(sizes[:, 0] * sizes[:, 1])),):
    m = items[:, 0] == sel
    print("root job: bands %d, first start %.0f, last band end %.0f, tb end %.0f" % (m.sum(), start[m].min(), bend[m].max(), tend[m].max()))
    for i in np.where(m)[0][np.argsort(items[m][:, 1])]:
        if tend[i] == 0 and raw[i, 5] > 0:
            print("   band %2d: %.0f shader cycles in %.0f us = %.2f GHz" % (items[i, 1], float(raw[i, 5]), bend[i] - start[i], float(raw[i, 5]) / (bend[i] - start[i]) / 1e3))
        print("   band %2d: start %6.0f  wait %6.0f  end %6.0f  (running %.0f us = %.3f us/step; of which waiting for the helper %.0f us)" % (items[i, 1], start[i], wait_us[i], bend[i], bend[i] - start[i] - wait_us[i], (bend[i] - start[i] - wait_us[i]) / steps[i], raw[i, 4] / 100.0 if tend[i] == 0 else -1))

# helper wavefronts of the root's bands (MODE 2): share of their time spent polling for the sweep (ticks of 10 ns)
if len(buf) >= 4 + 16 * n + 8 * 22 * n:
    hs = np.frombuffer(buf, np.uint64, 16 * n, 4 + 16 * n + 48 * n).reshape(n, 16).astype(np.float64)
    m = items[:, 0] == np.argmax(sizes[:, 0] * sizes[:, 1])
    tot, wt = hs[m][:, 8:16].mean(0), hs[m][:, 0:8].mean(0)
    print("root bands, helper wavefronts 1..7: mean us in all   " + " ".join("%7.0f" % (v / 100) for v in tot[1:]))
    print("                                   of which polling " + " ".join("%7.0f" % (v / 100) for v in wt[1:]))
