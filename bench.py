#!/usr/bin/env python3
"""bench.py — DP cell-updates/s of the alignGraphs hot path on the BASELINE workload (256 seqs x 1000 aa, WAG).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (config.workload): the 255 graph-vs-graph alignGraphs jobs of one progressive pass over the synthetic
256 x 1000 aa family (tests/gen.py seed 3, guide tree tests/golden/c3.tree, --mldist), 2.71e8 DP cells, exactly the
jobs the product driver issues (captured with `pgmsa --dump_jobs` on this GPU during untimed set-up).
One "step" = one pass of the hot path over that batch with the inputs resident in HBM: prep (float casts,
T = M^T g2) + emission scores + the fill kernel (DP fill of every band and, after a job's last band, its traceback),
then the result/mapping copy back to the host.
Multi-GPU: one process per GPU, every rank runs the same workload on its own device (independent jobs, no collective
on the data path; weak scaling); torch.distributed is used only for the barrier and the max-over-ranks clock.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nseq", type=int, default=256, help="debug: smaller family (changes the workload; not the headline)")
    ap.add_argument("--len", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import numpy as np
    import gen
    import prographmsa_amd as pg
    from prographmsa_amd import jobs as J

    ctx = pg.Context(local_rank)
    devname, cus = ctx.device_info()

    # ---- untimed set-up: produce the jobs with the product driver on this GPU --------------------
    tmp = tempfile.mkdtemp(prefix="pgm_bench_r%d_" % rank)
    fa = os.path.join(tmp, "fam.fa")
    headline = (args.nseq == 256 and args.len == 1000)
    with open(fa, "w") as f:
        f.write(gen.fasta(gen.gen(args.nseq, args.len, 3)))
    env = dict(os.environ, PGM_DEVICE=str(local_rank))
    dump = os.path.join(tmp, "jobs.bin")
    if headline:
        tree_args = ["-t", os.path.join(ROOT, "tests", "golden", "c3.tree")]
    else:   # debug sizes: NW guide tree from the GPU all-pairs stage
        tr = subprocess.run([pg.PGMSA_PATH, "-a", "-m", "-T", "-i", "0", fa], capture_output=True, text=True, env=env, check=True).stdout
        with open(os.path.join(tmp, "t.tree"), "w") as f:
            f.write(tr)
        tree_args = ["-t", os.path.join(tmp, "t.tree")]
    t0 = time.time()
    r = subprocess.run([pg.PGMSA_PATH, "--fasta", "-m"] + tree_args + ["--dump_jobs", dump, "--stats", "-o", os.path.join(tmp, "out.fa"), fa],
                       capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise SystemExit("pgmsa failed: " + r.stderr)
    e2e_wall = time.time() - t0
    stats = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith('{"backend"')][-1])
    # the reference's default flow from sequences alone (`--fasta -a`: all-pairs guide tree, two rounds of progressive
    # alignment + guide-tree re-estimation, final alignment), rank 0 only; the FASTA's md5 is checked against the fixture of
    # the reference binary's output for the headline family
    default_flow = None
    if rank == 0:
        t0 = time.time()
        r2 = subprocess.run([pg.PGMSA_PATH, "--fasta", "-a", "--stats", fa], capture_output=True, text=True, env=env)
        dt2 = time.time() - t0
        if r2.returncode == 0:
            import hashlib
            st2 = json.loads([ln for ln in r2.stderr.splitlines() if ln.startswith('{"backend"')][-1])
            default_flow = {"cmd": "pgmsa --fasta -a", "wall_s": round(dt2, 3), "tree_s": st2["tree_s"], "progressive_s": st2["progressive_s"],
                            "align_cells": st2["align_cells"], "nw_cells": st2["nw_cells"]}
            if headline:
                want = json.load(open(os.path.join(ROOT, "tests", "golden", "md5.json"))).get("c3.a_iter.out.fa")
                default_flow["fasta_identical_to_reference"] = (hashlib.md5(r2.stdout.encode()).hexdigest() == want)
                default_flow["reference_wall_s"] = 919   # bin/ProGraphMSA_64 --fasta -a, one core of the build container
    jobs = J.load_jobs(dump)
    os.remove(dump)
    batch = J.Batch(ctx, jobs)
    cells = batch.cells

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
        batch.fetch_raw()
    if args.warmup > 0:
        check = batch.fetch()   # results of the last warm-up step as Python objects (sanity check)
        assert all(r["status"] == 0 and len(r["map1"]) > 0 for r in check)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()          # pgm_align_batch_run: prep + emission + fill/traceback kernels
        batch.fetch_raw()    # pgm_align_batch_fetch: wait, D2H of scores and mappings into the caller's buffers
    barrier()
    dt = time.perf_counter() - t0
    total_cells = float(cells) * args.steps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([total_cells], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dt, total_cells = float(t.item()), float(c.item())

    # ---- roofline of the dominant kernel (pgm_fill_kernel: DP fill + tracebacks), HIP events on the library's stream ----
    reps = sorted(batch.time(1) for _ in range(7))          # per-stage device times, median of 7 single runs
    ms_prep, ms_emis = sorted(r[0] for r in reps)[3], sorted(r[1] for r in reps)[3]
    ms_fill = sorted(r[2] for r in reps)[3]
    alg_bytes = 16.0 * cells            # one float4 {M,X,W,Y} store per cell; S is materialised by the emission
    achieved = alg_bytes / (ms_fill * 1e-3) / 1e9   # kernel, so this kernel also reads 4 B/cell that are not counted here
    # HBM traffic per launch from the PMC passes of tools/profile_bench.sh (FETCH_SIZE, WRITE_SIZE in KB; gfx950: reads doubled)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r1_v22_pmc.json")
    if headline and os.path.exists(pmc_path):
        pmc_all = json.load(open(pmc_path))
        pmc = next((v for k, v in pmc_all.items() if k.startswith("pgm_fill_kernel")), {})
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            traffic = round((2.0 * pmc["FETCH_SIZE"]["mean_kb"] + pmc["WRITE_SIZE"]["mean_kb"]) * 1024.0)
    roofline = {"bound": "hbm", "kernel": "pgm_fill_kernel", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic, "algorithmic_bytes": alg_bytes,
                "traffic_source": "profiles/r1_v22_pmc.json (rocprofv3 --pmc, separate passes; bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE)",
                "ms": {"prep": round(ms_prep, 4), "emission": round(ms_emis, 4), "fill_and_traceback": round(ms_fill, 4)},
                "fill_gcups": round(cells / (ms_fill * 1e-3) / 1e9, 3)}

    # ---- all-pairs stage (DistanceFactoryAlign, `-a`): the 32 640 alignPair jobs of the same family, sharded over the
    # ranks (strong scaling of this stage: no collective, rank 0 would gather counts/gaps) --------------------------------
    import ctypes as C
    from prographmsa_amd import workqueue as wq
    seqs_aa = gen.gen(args.nseq, args.len, 3)
    order = "ACDEFGHIKLMNPQRSTVWY"
    enc = [np.array([order.index(c) for c in s[1:] if True] if s.startswith("M") else [order.index(c) for c in s], np.int8) for s in seqs_aa]
    lens = [len(e) for e in enc]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    syms = np.concatenate(enc).astype(np.int8)
    pairs = wq.all_pairs(len(enc))
    mine = wq.shard_pairs(lens, rank, world)
    pi = np.array([pairs[p][0] for p in mine], np.uint32)
    pj = np.array([pairs[p][1] for p in mine], np.uint32)
    score = np.loadtxt(os.path.join(ROOT, "prographmsa_amd", "host", "data", "nw_aa.imat"), skiprows=1, dtype=np.int32).reshape(-1)
    counts = np.zeros(len(mine) * 400, np.int32)
    gaps = np.zeros(len(mine), np.uint32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    nw_cells = float(sum(lens[pairs[p][0]] * lens[pairs[p][1]] for p in mine))

    def nw_call():
        pg.check(pg.lib.pgm_nw_pairs_batch(ctx.handle, 20, P(score, C.c_int32), -10, -2, len(enc), P(syms, C.c_int8), P(offs, C.c_uint32),
                                           len(mine), P(pi, C.c_uint32), P(pj, C.c_uint32), P(counts, C.c_int32), P(gaps, C.c_uint32)))
    nw_call()
    barrier()
    t0 = time.perf_counter()
    nw_call()
    barrier()
    nw_dt = time.perf_counter() - t0
    nw_kernel_ms = float(pg.lib.pgm_nw_last_kernel_ms(ctx.handle))
    nw_total = nw_cells
    if world > 1:
        t = torch.tensor([nw_dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([nw_cells], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        nw_dt, nw_total = float(t.item()), float(c.item())

    # ---- context-specific leaf profiles (CSProfile::createProfile) at the scale of BASELINE config 5: a synthetic library
    # of K = 4000 context profiles (13 columns), the leaves of a 1024 x 600 aa family sharded over the ranks --------------
    cs = None
    if headline:
        rng = np.random.default_rng(5)
        K, ncols, nleaf, L = 4000, 13, 1024, 600
        p = rng.gamma(0.3, 1.0, (K, ncols, 20)) + 1e-4
        p /= p.sum(2, keepdims=True)
        w = 1.3 * 0.9 ** np.abs(np.arange(ncols) - ncols // 2)
        lp = np.zeros((K, ncols, 21))
        lp[:, :, :20] = np.log(p) * w[None, :, None]
        lpf = np.ascontiguousarray(lp, np.float64).reshape(-1)
        cf = np.ascontiguousarray(p[:, ncols // 2, :], np.float64).reshape(-1)
        prf = np.log(rng.dirichlet(np.ones(K)))
        lo, hi = wq.shard_range(nleaf, rank, world)
        nl = hi - lo
        syms_cs = rng.integers(0, 20, nl * L).astype(np.int8)
        offs_cs = (np.arange(nl + 1) * L).astype(np.uint32)
        out_offs = (np.arange(nl + 1) * 20 * (L + 2)).astype(np.uint64)
        tau = np.full(nl, 0.3)
        pi_cs = np.full(20, 0.05)
        pu_cs = np.full(nl * 20, 0.05)
        out_cs = np.zeros(int(out_offs[-1]))
        pg.check(pg.lib.pgm_csprofile_load(ctx.handle, K, ncols, P(lpf, C.c_double), P(cf, C.c_double), P(prf, C.c_double)))

        def cs_call():
            pg.check(pg.lib.pgm_csprofile_create_batch(ctx.handle, nl, P(syms_cs, C.c_int8), P(offs_cs, C.c_uint32), P(tau, C.c_double),
                                                       P(pi_cs, C.c_double), P(pu_cs, C.c_double), P(out_cs, C.c_double), P(out_offs, C.c_uint64)))
        cs_call()
        barrier()
        t0 = time.perf_counter()
        cs_call()
        barrier()
        cs_dt = time.perf_counter() - t0
        cs_ms = float(pg.lib.pgm_csprofile_last_kernel_ms(ctx.handle))
        if world > 1:
            t = torch.tensor([cs_dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cs_dt = float(t.item())
        flop = 34.0 * K * L * nleaf   # ~K*L*34 flop per sequence (fp64)
        cs = {"library": "synthetic K=%d x %d columns" % (K, ncols), "sequences_total": nleaf, "length": L, "wall_s": round(cs_dt, 4),
              "rank0_kernel_ms": round(cs_ms, 3), "gflops_wall": round(flop / cs_dt / 1e9, 1), "scaling": "strong",
              "note": "whole pgm_csprofile_create_batch call incl. H2D of the residues and D2H of the 20 x (L+2) fp64 profiles; "
                      "reference config 5 spends ~450 s of 510 s in createProfile"}

    out = None
    if rank == 0:
        out = {
            "metric": "DP cell-updates/sec (GCUPS) + wall-clock on 256 seqs × 1000 aa, WAG",
            "value": round(total_cells / dt / 1e9, 4), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d seqs x %d aa, WAG, --mldist: the %d alignGraphs jobs of one progressive pass "
                                   "(%.3e DP cells) per GPU, inputs resident in HBM" % (args.nseq, args.len, len(jobs), cells),
                       "headline": headline, "device": devname, "cus": cus, "jobs": len(jobs), "cells_per_step": cells},
            "roofline": roofline,
            "all_pairs_nw": {"pairs_total": len(pairs), "cells_total": nw_total, "wall_s": round(nw_dt, 4),
                             "gcups_wall": round(nw_total / nw_dt / 1e9, 2), "rank0_kernel_ms": round(nw_kernel_ms, 3),
                             "rank0_kernel_gcups": round(nw_cells / (nw_kernel_ms * 1e-3) / 1e9, 2), "scaling": "strong",
                             "note": "whole pgm_nw_pairs_batch call incl. H2D of sequences and D2H of the 400-int count matrices; "
                                     "2 direction bits/cell stored (reference formulation: 12 B/cell)"},
            "csprofile": cs,
            "end_to_end": {"pgmsa_wall_s": round(e2e_wall, 3), "progressive_s": stats["progressive_s"],
                           "align_call_s": stats["align_s"], "note": "untimed set-up run of the product driver incl. host merges, H2D/D2H and hipMalloc",
                           "default_flow": default_flow},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib   # test-only CPU restatement, used here solely as the reported CPU baseline
            tc = time.perf_counter()
            passes, ccells = 0, 0
            while passes == 0 or (time.perf_counter() - tc < 10.0 and passes < 8):
                for j in jobs:
                    oracle_lib.align_graphs(j)
                passes += 1
                ccells += sum(j.cells for j in jobs)
            cdt = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": round(ccells / cdt / 1e9, 5), "unit": "GCUPS", "cores": 1, "kind": "port",
                                   "sample": "%d pass(es) over the same %d-job batch (%.3e cells, %.1f s), oracle/pgm_oracle.c -O2, 1 thread"
                                             % (passes, len(jobs), ccells, cdt)}
        print(json.dumps(out))
    batch.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
