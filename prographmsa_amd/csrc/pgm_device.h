// pgm_device.h — device-side job descriptor shared by the HIP kernels and the C-ABI host code.
#ifndef PGM_DEVICE_H_
#define PGM_DEVICE_H_

#include <stdint.h>
#include "../../include/pgm_hip.h"

#define PGM_HALO 0         /* every lane of a band owns a row (the rows above the band are "virtual lanes" in the LDS history) */
#define PGM_ROWS 64        /* rows per band = lanes of the sweeping wavefront */
#define PGM_VL 16          /* virtual lanes: the last 16 rows of the previous band, kept in the LDS history of the band below */
#define PGM_BLOCK 8        /* steps between two synchronisation points of a band */
#define PGM_NRING 128      /* columns of graph 2 whose predecessor summary is kept in LDS per sweeping wavefront */
#define PGM_KF 4           /* "far" predecessors per node served from the LDS history by a self-contained sweep (MODE 1) */
#define PGM_KF8 8          /* ... by the far helper of a MODE 2 sweep */
#define PGM_DCAP 28        /* largest predecessor distance served from the on-chip history (farther: cell storage) */
#define PGM_PF 4           /* MODE 2: the far helpers load the sources of long / remote entries this many steps ahead */
#define PGM_REMOTE_MAX 128 /* MODE 2: remote row entries per band served by the row helpers (passes 0 and 1) */
#define PGM_WAVES 8        /* wavefronts of a fill worker (one 512-thread workgroup per CU) */
#define PGM_WIDE_WAVES 4    /* pgm_band_kernel: sweeping wavefronts of a WIDE worker (sweeps whose history needs a quarter of the CU's LDS) */
#define PGM_POOL 163584    /* LDS bytes of a fill worker (one worker per CU: 163584 + 256 = 160 KiB) */
#define PGM_CPARTS 3       /* MODE 2: far helpers of the ROWS (row entry passes p = part + 3 q), 2 each for the near terms and the columns */
/* extra LDS of a MODE 2 sweep (PgmJob::mode2: seven helper wavefronts evaluate all but the chain terms), at PgmJob::aux_off: */
#define PGM_AUX_RES 0      /* float res[4][3][64]: maxima {M, X, Y} per lane of the steps t & 3 over all helpers (LDS float-max atomics; reset by the sweep) */
#define PGM_AUX_SBLK 3072  /* float sblk[3][8][64]: emission scores of the current block of each row helper (entry lanes read their owner's) */
#define PGM_AUX_EL 9216    /* uint2 elist[512]: far row entries of the band {owner lane | distance << 8 | remote << 31, cost} */
#define PGM_AUX_CNT 13312  /* int[2]: number of remote entries, number of local entries */
#define PGM_AUX_BYTES 13328

// Per-node predecessor summary prepared by the host from the CSR (80 bytes = 5 float4).
//   q0 = {cc, c2, c3, flags}   q1 = {fd0..fd3}   q2 = {fd4..fd7}   q3 = {fc0..fc3}   q4 = {fc4..fc7}
//   cc, c2, c3  cost of the edge from node-1 / node-2 / node-3 ("near" predecessors, served from registers of the
//               sweeping wavefront), +inf if absent: an absent edge then contributes -inf to every max without masking
//   flags  bits 0-3: number of "far" predecessors (any other edge), served from the on-chip history: at most PGM_KF8 in a
//                    MODE 2 job (far helper), at most PGM_KF otherwise
//          bits 8-15: largest distance among all on-chip predecessors of the node (>= 1)
//          bit 4   : generic — more far edges than that or a distance > PGM_DCAP: every non-chain predecessor of this
//                    node is read from the cell storage through the CSR lists (xp/xc/xv); then c2 = c3 = +inf, no far entries
//          bit 5   : kill — interior node without any predecessor
//          MODE 2, columns only:
//          bits 16-17: number of LONG entries (distance > PGM_DCAP; slots 7, 6, 5), served from the cell storage by the far
//                    helper (the count of bits 0-3 then covers the slots before them)
//          bits 20-24: number of OVERFLOW entries (on-chip entries beyond the slots of the summary, at most PGM_OV_ENT), kept in
//                    record (bits 25-31) of the job's overflow table (PgmJob::ov2, copied to LDS by the far helpers)
//   fd_k   (node - predecessor) of far edge k, 0 if absent;   fc_k its cost, +inf if absent
// In a MODE 2 job the far entries of the ROWS are not taken from here but from the CSR PgmJob::fp1 / fe1 (any number per
// row; an entry farther than PGM_DCAP or reaching above the band's virtual lanes is REMOTE: served from the cell storage).
// The column ring of a sweep keeps {q0, q1, q3} (MODE 0 / 1: 3 float4 per column) or all five (MODE 2).
#define PGM_NF_COUNT 15u
#define PGM_NF_GENERIC 16u
#define PGM_NF_KILL 32u
#define PGM_NF_NLONG(f) (((f) >> 16) & 3u)
#define PGM_NF_NOV(f) (((f) >> 20) & 31u)
#define PGM_NF_OVI(f) (((f) >> 25) & 127u)
#define PGM_OV_ENT 16      /* overflow entries per record */
#define PGM_OV_REC 128     /* records per job (16 KB of LDS) */
#define PGM_NLONG 3        /* long entries per column */
struct PgmNode2 {
    float cc, c2, c3;
    uint32_t flags;
    uint32_t fd[PGM_KF8];
    float fc[PGM_KF8];
};

// One alignGraphs job, resident in HBM.  All pointers are device pointers.
//
// DP storage ("cells"): the reference keeps four n1 x n2 column-major float matrices M,X,Y,W
// (GraphAlign.h:206-209).  Here a cell is one float4 {M,X,W,Y} and the matrix is stored in the order
// the wavefronts produce it: rows are cut into bands of 64; a band is swept by one wavefront whose
// lane l owns row y = 64 b + l, with lane l one column behind lane l-1, so "step" t of band b holds
// the cells (y = 64 b + l, x = t - l) at cells[((b * tsteps) + t) * 64 + l]  — every wave-level
// store is one contiguous 1 KB run.  With R = 1 << rshift rows per lane (lean jobs) a band is 64 R rows, lane l owns the
// rows 64 R b + R l + r (r < R) and step t holds R such runs: cells[(((b * tsteps) + t) * R + r) * 64 + l].
// Rows 0..n1-2 and columns 0..n2-2 are stored (the END row/column are never written by the reference
// either); row 0 / column 0 are the border initialisation of GraphAlign.h:212-234.
// Predecessor record of one node for the traceback's tile staging: the first PGM_TB_PK entries of the node's list in
// PredIterator order at a fixed address (no row-pointer indirection, i.e. one memory round trip less per tile).  Built
// by the prep kernel from the CSR.  Slots beyond the count repeat the last entry (every slot is a valid candidate read).
#define PGM_TB_PK 8
struct PgmTbNode {
    uint32_t cnt;                 // full predecessor count (may exceed PGM_TB_PK: the walker then uses the CSR lists)
    uint32_t c[PGM_TB_PK];        // predecessor node
    float v[PGM_TB_PK];           // edge cost
    uint32_t u[PGM_TB_PK];        // 0 for a regular edge, else 0x80000000 | repeat units
    uint32_t pad[7];              // 128 bytes
};

struct PgmJob {
    uint32_t n1, n2;       // node counts incl. START/END
    uint32_t dim, dp;      // alphabet size and padded size (multiple of 4)
    uint32_t nb;           // number of row bands = ceil((n1-1)/(64 R))
    uint32_t ncol;         // stored columns = n2-1
    uint32_t tsteps;       // steps per band = ncol + 63
    uint32_t maxn;         // max(n1,n2)
    uint32_t has_extras;   // some node of either graph has a predecessor other than its chain neighbour
    uint32_t has_far;      // some node has a predecessor served from the LDS history (the sweeps then record W, Y, X there)
    uint32_t mode2;        // one band per worker: its other seven wavefronts evaluate every term the sweep does not hold in registers (nslots = 1)
    uint32_t far_slack;    // min(4, smallest far distance): the far helpers may evaluate step t once step t - far_slack is recorded (4: the depth of the folded maxima)
    uint32_t aux_off;      // offset of the helper area (PGM_AUX_*) inside a sweep's LDS slot
    uint32_t hD, hDX;      // depth (steps, power of two) of the W / Y history and of the X history of a sweeping wavefront
    uint32_t slot_bytes;   // LDS bytes one sweeping wavefront needs for this job (history + column rings)
    uint32_t nslots;       // bands of this job one worker sweeps at a time = min(8, PGM_POOL / slot_bytes)
    uint32_t rshift;       // log2 of the rows a lane of a sweeping wavefront owns (R = 1 << rshift consecutive rows: a band is 64 R rows);
                           // cell and score storage are laid out per (band, step, r, lane), see `cells` / `S` below
    uint32_t lean;         // chain-only job (has_extras == 0): swept by pgm_sweep_chain — all bands of the job by ONE worker whose
                           // wavefronts hand the band's last row over through LDS rings (no hand-off through memory)
    uint32_t crit3;        // mode2 job swept by pgm_crit_kernel (no long / remote entries, overflow columns or generic nodes): sixteen wavefronts per band
    uint32_t c3_off;       // ... offset of its extra LDS (PGM_C3_*) inside the worker's slot
    uint32_t long1, long2; // mode2 only: graph 1 has REMOTE row entries / graph 2 has LONG column entries (served from the cell
                           // storage by the far helpers: farther than PGM_DCAP, or reaching above the band's virtual lanes)
    pgm_scores sc;

    // inputs as uploaded
    const double *sites1, *sites2;   // dim x n column-major
    const uint32_t *smap1, *smap2;   // profiles resident in HBM (pgm_site_ref): node i's column of sites is smap[i] (NULL: i)
    const double *M, *pi;            // dim x dim column-major, dim
    // per-node edge data prepared by the host from the CSR (costs are float32 as in Graph.h:223-239)
    const PgmNode2 *ni1, *ni2;       // near costs + far entries of every node
    const int32_t *xp1, *xp2;        // CSR ptr of the remaining ("extra") predecessors, n+1 entries
    const uint32_t *xc1, *xc2;       // extra predecessor node
    const float *xv1, *xv2;          // extra predecessor cost (repeat edges already evaluated)
    // mode2: far entries of the ROWS as a CSR (any number per row): .x = distance | remote << 31, .y = cost bits
    const int32_t *fp1;
    const uint2 *fe1;
    const uint2 *ov2;                // mode2: overflow table of the columns, nov2 records of PGM_OV_ENT {distance, cost bits}
    uint32_t nov2, ov_off;           // records in use; offset of the table's copy inside a sweep's LDS slot
    uint32_t rh_off;                 // LONG jobs: offset of the remote rows' W rings [3 row helpers][column & 31][lane]
    // full predecessor lists in PredIterator order (regular ascending, then repeats) for the traceback
    const int32_t *pp1, *pp2;
    const uint32_t *pc1, *pc2;
    const float *pv1, *pv2;
    const uint32_t *pu1, *pu2;       // 0 for a regular edge, else 0x80000000 | repeat units

    // produced by the prep kernel
    PgmTbNode *tb1, *tb2;  // [n1], [n2] predecessor records for the traceback
    float *g1f;            // [n1][dp]  float(sites1), node-major, zero padded
    float *a1;             // [n1]      g1^T pi
    float *t2;             // [n2][dp]  T = M^T g2
    float *b2;             // [n2]      pi^T g2

    // emission scores produced by the emission kernel, in the order the fill kernel consumes them:
    // S[((b * nblk + (t >> 3)) * 64 + lane) * 8 + (t & 7)] = S(y, x) of the cell lane `lane` of band b owns at step t
    // (R rows per lane: S[(((b * nblk + (t >> 3)) * R + r) * 64 + lane) * 8 + (t & 7)])
    float *S;
    uint32_t nblk;         // ceil(tsteps / 8)
    // Chain-only jobs of two SEQUENCE graphs — every profile column one-hot, uniform (1 / dim) or empty — need no score matrix: S(y, x)
    // depends on the classes of row y and column x alone, a (dim + 2)^2 table that pgm_lean_kernel builds per job from one node of
    // every class.  cls1 / cls2: class of every node (0 .. dim-1 the symbol, dim uniform, dim + 1 empty, 255 none of these; written
    // once per batch by pgm_classify_kernel); tabhdr[0] != 0: some node has no class (scores through S as for every other job),
    // tabhdr[1 + c] / tabhdr[1 + dim + 2 + c]: a node of class c of graph 1 / 2, plus one (0: the class is empty).  NULL: not a lean job.
    const uint8_t *cls1, *cls2;
    int *tabhdr;

    // lean jobs: the traceback's decisions, four bits per cell (see pgm_sweep_chain), [nb][nblk][R][64] words of eight
    // steps; the END node's predecessor cell (n1-2, n2-2) {M, X, W, Y}; whether the four matrices are written to `cells` at all
    uint32_t *codes;
    float4 *endcell;
    uint32_t keep_cells;

    // Pre-linked traceback tiles (pgm_prelink_tile): workers with nothing else to do build, for the 32 x 32 tiles of a fixed
    // grid that lie within PGM_LK_W / 2 tiles of the matrix diagonal, the successor of every (cell, state) — decided exactly as
    // the walker would — while the job's traceback worker is already walking.  ltab[tile k][state][32 ly + lx] is a 16-bit link
    // (pgm_lk_*); lready[r] = tiles of grid row r whose tables are complete and visible, lready[lrows] the next tile to take,
    // lready[lrows + 1] the grid row the walker is in (rows above it are still wanted).
    // Tile k = r * PGM_LK_W + j: grid row r (rows 32 r ..), grid column pgm_lk_first(r) + j.  lrows = 0: none (lean / small jobs).
    uint16_t *ltab;
    int *lready;
    uint32_t lrows, lcols; // grid rows, grid columns of the whole matrix

    // timeline of the last launch (100 MHz ticks of s_memrealtime): [0] the job's last band complete, [1] its traceback published
    unsigned long long *times;

    // DP storage
    float4 *cells;         // [nb][tsteps][R][64]
    int *prog;             // [nb] steps of band b that are complete and visible device-wide (zeroed before every launch)

    // traceback output + scratch
    uint32_t *map1, *map2; // capacity n1+n2
    float *mark_score;     // [maxn]
    uint32_t *mark_prev;   // [maxn]
    struct Result { float score; uint32_t n_tr_indels; uint32_t len; int32_t status; } *result;
    // The finished mappings (reversed into alignment order) and the result record are written straight into the batch's
    // pinned host block by the worker that walked the traceback: no device-to-host copy after the kernel.
    uint32_t *hmap1, *hmap2;
    Result *hresult;
};


// words of the batch's sync block the traceback kernel polls (a cache line of their own; [0] abort flag, [1] / [2] the tickets of the band and lean lists)
#define PGM_SY_BAND_TICKET 3  // ticket counter of pgm_band_kernel's list (with the other tickets in the first cache line)
#define PGM_SY_CRIT_TICKET 4  // ticket counter of the fill kernel's second launch (the jobs with the longest chains of sweeps)
#define PGM_SY_WIDE_TICKET 5  // ticket counter of pgm_band_kernel's wide bands
#define PGM_SY_LQ_N 32      // tracebacks that have started (pre-link announcements, ids in lq_ids)
#define PGM_SY_TB_DONE 33   // tracebacks finished
#define PGM_SY_TBQ_N 34     // ticket counter of the traceback kernel's job list
#define PGM_SY_TBQ_TAIL 35  // jobs in the ready queue of the traceback kernel that runs beside the sweeps
#define PGM_LK_W 4u        // grid tiles per grid row in the corridor around the diagonal (paths of the headline batch stay within 40 columns of it)
#define PGM_LK_T 32u       // tile edge
#define PGM_LK_NR 4u       // grid rows of tables a walker keeps in LDS (fetched together when known to be complete)
#define PGM_LK_MIN_ROWS 1216u   // jobs with fewer rows are not pre-linked (19 bands: the MODE 2 threshold)
#define PGM_LK_H 16u       // halo above / left of a tile staged with it: predecessors up to this far outside the tile still get a link
#define PGM_LK_TAB (3u * PGM_LK_T * PGM_LK_T)   // links per tile
// first grid column of the corridor in grid row r
__host__ __device__ inline uint32_t pgm_lk_first(uint32_t n1, uint32_t n2, uint32_t lcols, uint32_t r) {
    const uint32_t w = lcols < PGM_LK_W ? lcols : PGM_LK_W;
    const uint64_t xc = ((uint64_t)(PGM_LK_T * r + PGM_LK_T / 2u) * (n2 - 1u)) / (n1 - 1u);   // column of the diagonal at the tile row's middle
    const uint32_t g = (uint32_t)(xc / PGM_LK_T);
    const uint32_t lo = g >= w / 2u ? g - w / 2u : 0u;
    return lo + w > lcols ? lcols - w : lo;
}

// One unit of fill work: `count` consecutive bands (64 rows each) of job `job`, starting at `band`, one per wavefront of
// the worker that takes the item (count <= PgmJob::nslots <= 8; a MODE 2 job: one band, swept with seven helper wavefronts).  The list is ordered so that a job's bands come in
// ascending order (workers take the items in list order, see pgm_fill_kernel).
struct PgmItem {
    uint32_t job, band;
    uint32_t prio, count; // prio: wave priority 0..3 while the item (and a traceback that follows it) is processed
};

#endif
