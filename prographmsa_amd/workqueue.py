"""Static work sharing for the embarrassingly parallel stages (one process per GPU, no data-path collective).

The all-pairs stage (DistanceFactoryAlign::computePwDistances, reference src/DistanceFactoryAlign.h:29-56) is
N(N-1)/2 independent alignPair jobs.  Pairs are dealt round-robin over the ranks after sorting by cost
(L1*L2, longest first), so that every rank receives the same mix of long and short pairs; each rank pushes
its share through pgm_nw_pairs_batch on its own GPU (whose device-side queue does the fine-grained balancing)
and rank 0 gathers the (counts, gaps) back into the distance matrix.
"""


def all_pairs(n):
    """(i, j) for i < j in the reference's loop order."""
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def shard_pairs(lengths, rank, world):
    """Indices (into all_pairs(len(lengths))) owned by `rank`: cost-sorted round-robin."""
    pairs = all_pairs(len(lengths))
    order = sorted(range(len(pairs)), key=lambda p: (-lengths[pairs[p][0]] * lengths[pairs[p][1]], p))
    return order[rank::world]


def shard_range(n_units, rank, world):
    """Contiguous block partition of n_units independent units (jobs of one guide-tree level, leaves, ...)."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
