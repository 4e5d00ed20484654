"""world_size-2 gloo rehearsal of the multi-GPU path: the all-pairs stage is sharded over ranks with no data-path
collective; the only communication is the gather of results and the barrier / max-over-ranks clock bench.py uses.
The per-rank compute is done here by the CPU oracle (this is a test of the sharding logic, not of the kernels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from prographmsa_amd import workqueue as wq
    rng = np.random.default_rng(7)
    lens = [int(x) for x in rng.integers(20, 90, 9)]
    seqs = [rng.integers(0, 21, L).astype(np.int8) for L in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    syms = np.concatenate(seqs)
    score = rng.integers(-4, 6, 21 * 21).astype(np.int32)
    pairs = wq.all_pairs(len(lens))
    mine = wq.shard_pairs(lens, rank, world)
    pi = [pairs[p][0] for p in mine]
    pj = [pairs[p][1] for p in mine]
    counts, gaps = oracle_lib.nw_pairs(20, score, -10, -2, syms, offs, pi, pj)
    # gather on rank 0 (what the distance-matrix assembly does)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, counts, gaps))
    # barrier + max-over-ranks clock, as in bench.py
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(len(mine))], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    if rank == 0:
        full_c = np.zeros((len(pairs), 400), np.int32)
        full_g = np.zeros(len(pairs), np.uint32)
        for idx, cc, gg in gathered:
            full_c[idx] = cc
            full_g[idx] = gg
        ref_c, ref_g = oracle_lib.nw_pairs(20, score, -10, -2, syms, offs, [p[0] for p in pairs], [p[1] for p in pairs])
        ok = np.array_equal(full_c, ref_c) and np.array_equal(full_g, ref_g) and t.item() == float(world) and c.item() == len(pairs)
        open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_all_pairs_sharded_over_two_ranks(tmp_path, oracle_build):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"
