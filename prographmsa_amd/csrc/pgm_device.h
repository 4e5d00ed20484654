// pgm_device.h — device-side job descriptor shared by the HIP kernels and the C-ABI host code.
#ifndef PGM_DEVICE_H_
#define PGM_DEVICE_H_

#include <stdint.h>
#include "../../include/pgm_hip.h"

#define PGM_BAND 64        /* rows per band = lanes per wavefront */
#define PGM_BLOCK 16       /* steps between two synchronisation points of a band */

// One alignGraphs job, resident in HBM.  All pointers are device pointers.
//
// DP storage ("cells"): the reference keeps four n1 x n2 column-major float matrices M,X,Y,W
// (GraphAlign.h:206-209).  Here a cell is one float4 {M,X,Y,W} and the matrix is stored in the order
// the wavefronts produce it: rows are cut into bands of 64 (lane = row within band), a band sweeps the
// columns with lane l one column behind lane l-1, so "step" t of band b holds the cells
// (y = 64 b + l, x = t - l).  cells[((b * tsteps) + t) * 64 + l]  — every wave-level store is one
// contiguous 1 KiB line.  Rows 0..n1-2 and columns 0..n2-2 are stored (the END row/column are never
// written by the reference either); row 0 / column 0 are the border initialisation of GraphAlign.h:212-234.
struct PgmJob {
    uint32_t n1, n2;       // node counts incl. START/END
    uint32_t dim, dp;      // alphabet size and padded size (multiple of 4)
    uint32_t nb;           // number of row bands = ceil((n1-1)/64)
    uint32_t ncol;         // stored columns = n2-1
    uint32_t tsteps;       // steps per band = ncol + 63
    uint32_t maxn;         // max(n1,n2)
    pgm_scores sc;

    // inputs as uploaded
    const double *sites1, *sites2;   // dim x n column-major
    const double *M, *pi;            // dim x dim column-major, dim
    // per-node edge data prepared by the host from the CSR (costs are float32 as in Graph.h:223-239)
    const float *cc1, *cc2;          // cost of the chain edge (node-1 -> node), +inf if absent
    const int32_t *xp1, *xp2;        // CSR ptr of the remaining ("extra") predecessors, n+1 entries
    const uint32_t *xc1, *xc2;       // extra predecessor node
    const float *xv1, *xv2;          // extra predecessor cost (repeat edges already evaluated)
    const uint8_t *kill1, *kill2;    // 1 if an interior node has no predecessor at all
    // full predecessor lists in PredIterator order (regular ascending, then repeats) for the traceback
    const int32_t *pp1, *pp2;
    const uint32_t *pc1, *pc2;
    const float *pv1, *pv2;
    const uint32_t *pu1, *pu2;       // 0 for a regular edge, else 0x80000000 | repeat units

    // produced by the prep kernel
    float *g1f;            // [n1][dp]  float(sites1), node-major, zero padded
    float *a1;             // [n1]      g1^T pi
    float *t2;             // [n2][dp]  T = M^T g2
    float4 *aux2;          // [n2]      {pi^T g2, chain cost, extras begin (bits), extras count | kill<<31 (bits)}

    // DP storage
    float4 *cells;         // [nb][tsteps][64]
    float2 *brow;          // [nb][ncol] {W,Y} of the last row of each band (hand-off to the next band)

    // traceback output + scratch
    uint32_t *map1, *map2; // capacity n1+n2
    float *mark_score;     // [maxn]
    uint32_t *mark_prev;   // [maxn]
    struct Result { float score; uint32_t n_tr_indels; uint32_t len; int32_t status; } *result;
};

#endif
