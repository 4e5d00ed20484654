"""alignGraphs jobs as numpy arrays + ctypes views for the C ABI (tests / bench plumbing).

Dump format written by the host driver (`pgmsa --dump_jobs FILE`, host/graph_align.cpp): per job
  u32 magic 'PGMJ'; graph1; graph2; M (dim*dim f64, column-major); pi (dim f64); pgm_scores (10 f32)
graph := u32 n, dim, nnz_e, nnz_r; sites (dim*n f64); e_rowptr (n+1 i32); e_col (u32); e_val (f32);
         if nnz_r: r_rowptr (n+1 i32); r_col (u32); r_units (u32)
"""
import ctypes as C

import numpy as np

from . import (PGM_BATCH_KEEP_MATRICES, PGM_GAP, check, lib, pgm_align_out, pgm_graph, pgm_model, pgm_scores)

MAGIC = 0x4A4D4750


class Graph:
    def __init__(self, n, dim, sites, e_rowptr, e_col, e_val, r_rowptr=None, r_col=None, r_units=None):
        self.n, self.dim = int(n), int(dim)
        self.sites = np.ascontiguousarray(sites, dtype=np.float64).reshape(-1)
        self.e_rowptr = np.ascontiguousarray(e_rowptr, dtype=np.int32)
        self.e_col = np.ascontiguousarray(e_col, dtype=np.uint32)
        self.e_val = np.ascontiguousarray(e_val, dtype=np.float32)
        has_r = r_col is not None and len(r_col) > 0
        self.r_rowptr = np.ascontiguousarray(r_rowptr, dtype=np.int32) if has_r else None
        self.r_col = np.ascontiguousarray(r_col, dtype=np.uint32) if has_r else None
        self.r_units = np.ascontiguousarray(r_units, dtype=np.uint32) if has_r else None
        assert self.sites.size == self.n * self.dim and self.e_rowptr.size == self.n + 1

    def c(self):
        def p(a, t):
            return a.ctypes.data_as(C.POINTER(t)) if a is not None and a.size else C.POINTER(t)()
        g = pgm_graph()
        g.n, g.dim = self.n, self.dim
        g.sites = p(self.sites, C.c_double)
        g.e_rowptr = p(self.e_rowptr, C.c_int32)
        g.e_col = p(self.e_col, C.c_uint32)
        g.e_val = p(self.e_val, C.c_float)
        g.r_rowptr = p(self.r_rowptr, C.c_int32)
        g.r_col = p(self.r_col, C.c_uint32)
        g.r_units = p(self.r_units, C.c_uint32)
        return g


class Job:
    def __init__(self, g1, g2, M, pi, scores):
        self.g1, self.g2 = g1, g2
        self.M = np.ascontiguousarray(M, dtype=np.float64).reshape(-1)
        self.pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(-1)
        self.scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(10)

    @property
    def cells(self):
        return (self.g1.n - 2) * (self.g2.n - 2)


def _read_graph(buf, off):
    n, dim, nnz_e, nnz_r = np.frombuffer(buf, np.uint32, 4, off)
    off += 16
    sites = np.frombuffer(buf, np.float64, int(n) * int(dim), off); off += 8 * int(n) * int(dim)
    rp = np.frombuffer(buf, np.int32, int(n) + 1, off); off += 4 * (int(n) + 1)
    col = np.frombuffer(buf, np.uint32, int(nnz_e), off); off += 4 * int(nnz_e)
    val = np.frombuffer(buf, np.float32, int(nnz_e), off); off += 4 * int(nnz_e)
    rrp = rcol = ru = None
    if nnz_r:
        rrp = np.frombuffer(buf, np.int32, int(n) + 1, off); off += 4 * (int(n) + 1)
        rcol = np.frombuffer(buf, np.uint32, int(nnz_r), off); off += 4 * int(nnz_r)
        ru = np.frombuffer(buf, np.uint32, int(nnz_r), off); off += 4 * int(nnz_r)
    return Graph(n, dim, sites, rp, col, val, rrp, rcol, ru), off


def load_jobs(path):
    buf = open(path, "rb").read()
    off, jobs = 0, []
    while off < len(buf):
        magic = np.frombuffer(buf, np.uint32, 1, off)[0]
        assert magic == MAGIC, "bad job dump"
        off += 4
        g1, off = _read_graph(buf, off)
        g2, off = _read_graph(buf, off)
        d = g1.dim
        M = np.frombuffer(buf, np.float64, d * d, off); off += 8 * d * d
        pi = np.frombuffer(buf, np.float64, d, off); off += 8 * d
        sc = np.frombuffer(buf, np.float32, 10, off); off += 40
        jobs.append(Job(g1, g2, M, pi, sc))
    return jobs


class CJobs:
    """ctypes argument arrays for a list of jobs (keeps the numpy buffers alive)."""

    def __init__(self, jobs):
        self.jobs = jobs
        n = len(jobs)
        self.n = n
        self._g1 = [j.g1.c() for j in jobs]
        self._g2 = [j.g2.c() for j in jobs]
        self._m = []
        for j in jobs:
            m = pgm_model()
            m.M = j.M.ctypes.data_as(C.POINTER(C.c_double))
            m.pi = j.pi.ctypes.data_as(C.POINTER(C.c_double))
            self._m.append(m)
        self.g1 = (C.POINTER(pgm_graph) * n)(*[C.pointer(g) for g in self._g1])
        self.g2 = (C.POINTER(pgm_graph) * n)(*[C.pointer(g) for g in self._g2])
        self.m = (C.POINTER(pgm_model) * n)(*[C.pointer(m) for m in self._m])
        self.sc = (pgm_scores * n)()
        for i, j in enumerate(jobs):
            C.memmove(C.byref(self.sc[i]), j.scores.ctypes.data, 40)
        self.maps1 = [np.zeros(j.g1.n + j.g2.n, np.uint32) for j in jobs]
        self.maps2 = [np.zeros(j.g1.n + j.g2.n, np.uint32) for j in jobs]
        self.out = (pgm_align_out * n)()
        for i in range(n):
            self.out[i].map1 = self.maps1[i].ctypes.data_as(C.POINTER(C.c_uint32))
            self.out[i].map2 = self.maps2[i].ctypes.data_as(C.POINTER(C.c_uint32))

    def results(self):
        res = []
        for i in range(self.n):
            o = self.out[i]
            res.append(dict(score=o.score, n_tr_indels=o.n_tr_indels, status=o.status,
                            map1=self.maps1[i][:o.len].copy(), map2=self.maps2[i][:o.len].copy()))
        return res


def align_graphs_batch(ctx, jobs):
    """pgm_align_graphs_batch through the C ABI; returns a list of result dicts."""
    cj = CJobs(jobs)
    rc = lib.pgm_align_graphs_batch(ctx.handle, cj.n, cj.g1, cj.g2, cj.m, cj.sc, cj.out)
    if rc not in (0, 3):
        check(rc, "pgm_align_graphs_batch")
    return cj.results()


class Batch:
    """Staged form: create (upload) / run / fetch, inputs stay resident in HBM between runs."""

    def __init__(self, ctx, jobs, keep_matrices=False):
        """keep_matrices: PGM_BATCH_KEEP_MATRICES (every job's M, X, Y, W stay readable through read_matrices: test hook)."""
        self.ctx = ctx
        self.cj = CJobs(jobs)
        self.handle = C.c_void_p()
        check(lib.pgm_align_batch_create_ex(ctx.handle, self.cj.n, self.cj.g1, self.cj.g2, self.cj.m, self.cj.sc,
                                            PGM_BATCH_KEEP_MATRICES if keep_matrices else 0, C.byref(self.handle)), "pgm_align_batch_create_ex")

    @property
    def cells(self):
        return int(lib.pgm_align_batch_cells(self.handle))

    def run(self):
        check(lib.pgm_align_batch_run(self.ctx.handle, self.handle), "pgm_align_batch_run")

    def fetch_raw(self):
        """The C-ABI call alone: score / mappings land in the caller buffers of self.cj (no Python result objects)."""
        rc = lib.pgm_align_batch_fetch(self.ctx.handle, self.handle, self.cj.out)
        if rc not in (0, 3):
            check(rc, "pgm_align_batch_fetch")

    def fetch(self):
        self.fetch_raw()
        return self.cj.results()

    def time(self, reps):
        a, e, b, c = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        check(lib.pgm_align_batch_time(self.ctx.handle, self.handle, reps, C.byref(a), C.byref(e), C.byref(b), C.byref(c)))
        return a.value, e.value, b.value, c.value

    def stage_times(self, reset=True):
        """Mean (prep, emission, fill) device ms over the launches fetched since the last reset, and how many those were."""
        a, e, f, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint32()
        check(lib.pgm_align_batch_stage_times(self.handle, 1 if reset else 0, C.byref(a), C.byref(e), C.byref(f), C.byref(n)))
        return a.value, e.value, f.value, n.value

    def read_matrices(self, job):
        j = self.cj.jobs[job]
        N = j.g1.n * j.g2.n
        mats = [np.zeros(N, np.float32) for _ in range(5)]
        ptrs = [m.ctypes.data_as(C.POINTER(C.c_float)) for m in mats]
        check(lib.pgm_align_batch_read_matrices(self.ctx.handle, self.handle, job, *ptrs))
        return [m.reshape(j.g2.n, j.g1.n).T for m in mats]   # [y, x] views of the column-major n1 x n2 layout

    def close(self):
        if self.handle:
            lib.pgm_align_batch_destroy(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# -------------------------------------------------------------------------------------------------
# Synthetic jobs for parity tests: random DAGs with skip edges, unreachable-cost edges and repeat edges.
def random_graph(rng, n, dim, skip_frac=0.15, repeat_frac=0.0, onehot_frac=0.5, drop_chain_frac=0.03, skip_span=12, skip_max=3,
                 repeat_span=20, chain_cost_frac=0.0):
    sites = np.zeros((n, dim))
    for i in range(1, n - 1):
        if rng.random() < onehot_frac:
            sites[i, rng.integers(dim)] = 1.0
        else:
            v = rng.gamma(0.5, 1.0, dim) + 1e-3
            sites[i] = v / np.sqrt((v * v).sum())
    rp, col, val = [0], [], []
    rrp, rcol, ru = [0], [], []
    for v in range(n):
        preds = {}
        if v > 0:
            if rng.random() >= drop_chain_frac or v == 1:
                preds[v - 1] = 0.0 if rng.random() >= chain_cost_frac else float(rng.choice([0.80369, 1.60738, 2.5]))
            if v > 1 and rng.random() < skip_frac:
                for _ in range(int(rng.integers(1, skip_max + 1))):
                    p = int(rng.integers(max(0, v - skip_span), v))
                    c = float(rng.choice([0.0, 0.80369, 1.60738, 2.5, 7.25]))
                    preds[p] = min(preds.get(p, 1e9), c)
            if not preds:
                preds[v - 1] = 0.0
        for p in sorted(preds):
            col.append(p)
            val.append(np.float32(min(np.float32(preds[p]), np.float32(1e4))) - np.float32(1e4))
        rp.append(len(col))
        if v > 2 and rng.random() < repeat_frac:
            p = int(rng.integers(max(0, v - repeat_span), v - 1))
            rcol.append(p)
            ru.append(int(rng.integers(1, 4)))
        rrp.append(len(rcol))
    return Graph(n, dim, sites.reshape(-1), rp, col, val, rrp, rcol, ru)


def random_job(seed, n1, n2, dim=20, **kw):
    rng = np.random.default_rng(seed)
    g1 = random_graph(rng, n1, dim, **kw)
    g2 = random_graph(rng, n2, dim, **kw)
    # a plausible joint matrix M = diag(pi) P with P row-stochastic, pi uniform (the reference's quirk)
    P = rng.gamma(0.3, 1.0, (dim, dim)) + 0.02
    P += np.eye(dim) * dim * 0.3
    P /= P.sum(1, keepdims=True)
    pi = np.full(dim, 1.0 / dim)
    M = (pi[:, None] * P)
    scores = np.array([-6.9, -0.71, -0.03, -7.5, -1.2, -9.9, -4.4, -0.19, 3.3, 1.7], np.float32)
    scores += rng.normal(0, 0.01, 10).astype(np.float32)
    return Job(g1, g2, M.T.reshape(-1), pi, scores)   # column-major M
