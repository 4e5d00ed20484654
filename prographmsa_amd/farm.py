"""Work sharing for the embarrassingly parallel stages across GPUs (bench / test plumbing; no data-path collective).

The product's farm lives in C++ (`host/distance.cpp`, computePwDistances): one host thread per device context pulls
pair tiles from an atomic counter.  `bench.py` runs one PROCESS per GPU (the launch contract), so the same queue is
rebuilt here on top of the rendezvous store of `torch.distributed` (`TCPStore.add` is an atomic fetch-and-add served
by rank 0): every rank pulls the next tile number until the tiles run out.  Same tiles, same order, same tile size
rule as the C++ farm; which rank computes a tile does not change any result.

Reference: the i < j double loop of DistanceFactoryAlign::computePwDistances (src/DistanceFactoryAlign.h:29-56) is the
set of independent jobs being farmed; the jobs of one guide-tree level (ProgressiveAlignment.cpp:50-51) are the other.
"""


def all_pairs(n):
    """(i, j) for i < j in the reference's loop order."""
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def sorted_pairs(lengths):
    """All pairs, longest (L1 * L2) first, ties in loop order: the order in which the farm hands them out."""
    pairs = all_pairs(len(lengths))
    order = sorted(range(len(pairs)), key=lambda p: (-lengths[pairs[p][0]] * lengths[pairs[p][1]], p))
    return [pairs[p] for p in order]


def tile_size(npairs, workers, override=None):
    """Pairs per tile: three tiles per worker, at least 256 pairs (host/distance.cpp uses the same rule, and says why)."""
    if override:
        return max(1, int(override))
    return max(256, (npairs + 3 * workers - 1) // (3 * workers))


class TicketQueue:
    """Atomic ticket counter shared by all ranks: `next()` returns 0, 1, 2, ... exactly once each across the job."""

    def __init__(self, name, world=1, store=None):
        self.name, self.world, self.store, self.local = name, world, store, 0
        if world > 1 and store is None:
            from torch.distributed.distributed_c10d import _get_default_store
            self.store = _get_default_store()

    def next(self):
        if self.world <= 1:
            self.local += 1
            return self.local - 1
        return int(self.store.add(self.name, 1)) - 1


def lpt_shards(costs, world):
    """Longest-processing-time-first assignment of independent jobs to `world` ranks: list of index lists."""
    shards, load = [[] for _ in range(world)], [0.0] * world
    for i in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += costs[i]
    return shards


def shard_range(n_units, rank, world):
    """Contiguous block partition of n_units independent units (leaves of a family, ...)."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
