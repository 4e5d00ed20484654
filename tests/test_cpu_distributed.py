"""world_size-2 gloo rehearsals of the multi-GPU path (CPU only): the all-pairs stage is a farm of pair tiles that the
ranks pull from one atomic ticket counter (the rendezvous store of torch.distributed), with no data-path collective; the
only communication is the gather of results and the barrier / max-over-ranks clock bench.py uses.
The per-rank compute is done here by the CPU oracle (these are tests of the queue and of the plumbing, not of the kernels)."""
import json
import os
import subprocess
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
    from prographmsa_amd import farm
    rng = np.random.default_rng(7)
    lens = [int(x) for x in rng.integers(20, 90, 9)]
    seqs = [rng.integers(0, 21, L).astype(np.int8) for L in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    syms = np.concatenate(seqs)
    score = rng.integers(-4, 6, 21 * 21).astype(np.int32)
    pairs = farm.sorted_pairs(lens)
    tile = 5
    ntiles = (len(pairs) + tile - 1) // tile
    q = farm.TicketQueue("tiles", world)
    mine, res = [], []
    while True:
        t = q.next()
        if t >= ntiles:
            break
        sl = pairs[t * tile:(t + 1) * tile]
        mine.append(t)
        res.append(oracle_lib.nw_pairs(20, score, -10, -2, syms, offs, [p[0] for p in sl], [p[1] for p in sl]))
    # gather on rank 0 (what the distance-matrix assembly does)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, res))
    # barrier + max-over-ranks clock, as in bench.py
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(len(mine))], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    if rank == 0:
        full_c = np.zeros((len(pairs), 400), np.int32)
        full_g = np.zeros(len(pairs), np.uint32)
        seen = []
        for tiles, rr in gathered:
            for tt, (cc, gg) in zip(tiles, rr):
                seen.append(tt)
                full_c[tt * tile:tt * tile + len(gg)] = cc
                full_g[tt * tile:tt * tile + len(gg)] = gg
        ref_c, ref_g = oracle_lib.nw_pairs(20, score, -10, -2, syms, offs, [p[0] for p in pairs], [p[1] for p in pairs])
        ok = (sorted(seen) == list(range(ntiles)) and np.array_equal(full_c, ref_c) and np.array_equal(full_g, ref_g)
              and t.item() == float(world) and c.item() == ntiles)
        open(os.path.join(out_dir, "ok"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_all_pairs_tiles_pulled_by_two_ranks(tmp_path, oracle_build):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"


@pytest.mark.parametrize("gpus", [1, 2, 3])
def test_bench_honours_gpus_flag_dry_run(gpus):
    """`python bench.py --gpus N` without a launcher starts one process per rank itself (torch.distributed.run) and rank 0
    prints one JSON line with n_gpus = N.  --dry-run: rendezvous, ticket queue and aggregation only (gloo, no GPU, no compute)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == gpus and out["dry_run"] is True
    assert out["tiles_pulled"] == out["tiles"] and out["every_tile_exactly_once"] and out["max_clock"] == float(gpus)
