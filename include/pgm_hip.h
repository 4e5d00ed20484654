/*
 * pgm_hip.h — C ABI of the MI355X (gfx950) accelerator for ProGraphMSA's hot path.
 *
 * The reference (acg-team/ProGraphMSA) has no plugin/FFI layer; its seams are three C++
 * functions.  Each entry point below replaces one of them (reference file:line cited per
 * function).  Plain C types only: pointers + sizes, caller-allocated outputs, `int`
 * status return (0 = PGM_OK).  The library is libpgm_hip.so (prographmsa_amd/csrc).
 *
 * The identical structs are consumed by the test-only CPU oracle (oracle/pgm_oracle.c),
 * which exports the same functions under the `pgmo_` prefix.
 */
#ifndef PGM_HIP_H_
#define PGM_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------ */
#define PGM_OK 0
#define PGM_ERR_INVALID 1    /* bad argument (null pointer, n < 2, dim mismatch ...)        */
#define PGM_ERR_DEVICE 2     /* HIP runtime error; see pgm_last_error()                     */
#define PGM_ERR_BACKTRACK 3  /* reference: error("backtracking failed") GraphAlign.h:410    */
#define PGM_ERR_NOMEM 4

#define PGM_GAP 0xFFFFFFFFu  /* (index_t)-1 in the reference's mappings                     */

/* ---- flattened reference types ------------------------------------------------------- */

/* Graph<ALPHABET> (reference src/Graph.h:31-33).  `sites` is the Eigen profile matrix
 * (dim x n, column-major, double; columns 0 and n-1 are the zero START/END columns).
 * `e_*` is the row-major CSR of `edges(to,from)`: row = node, col = predecessor (ascending),
 * value = min(cost,1e4f)-1e4f exactly as stored by Graph::setEdgesFromMap (Graph.h:85).
 * `r_*` is the CSR of `repeats(to,from)` (value = repeat-unit count); r_rowptr may be NULL
 * when the graph has no tandem-repeat edges.  Iteration order of PredIterator (Graph.h:196):
 * all e_ entries of the row, then all r_ entries. */
typedef struct pgm_graph {
    uint32_t n;
    uint32_t dim;
    const double *sites;
    const int32_t *e_rowptr; /* n+1 */
    const uint32_t *e_col;
    const float *e_val;
    const int32_t *r_rowptr; /* n+1 or NULL */
    const uint32_t *r_col;
    const uint32_t *r_units;
} pgm_graph;

/* The members of Model<ALPHABET> (reference src/Model.h:8-24) that reach alignGraphs:
 * M = diag(pi)*P(d), dim x dim column-major double; pi, dim doubles. */
typedef struct pgm_model {
    const double *M;
    const double *pi;
} pgm_model;

/* DynProgScores<ALPHABET> (reference src/GraphAlign.h:133-142), computed by the host caller
 * (it needs averageAlignmentLength and cmdlineopts) and passed by value. */
typedef struct pgm_scores {
    float gap_init, gap_extend, match_init, end_match, end_gap, end_skip;
    float start_gap, start_init, repeat_init, repeat_ext;
} pgm_scores;

/* AlignmentResult<ALPHABET> (reference src/GraphAlign.h:6-12).  map1/map2 are caller
 * allocated with capacity n1+n2 (the reference reserves the same, GraphAlign.h:297-298);
 * `len` entries are valid, PGM_GAP marks a gap, entry 0 is (0,0), the last (n1-1,n2-1). */
typedef struct pgm_align_out {
    float score;
    uint32_t n_tr_indels;
    uint32_t len;
    int32_t status; /* PGM_OK or PGM_ERR_BACKTRACK for this job */
    uint32_t *map1;
    uint32_t *map2;
} pgm_align_out;

typedef struct pgm_ctx pgm_ctx;
typedef struct pgm_align_batch pgm_align_batch;

/* ---- context ------------------------------------------------------------------------ */
int pgm_device_count(void);
/* One context per device (streams, buffer cache, resident arena).  Creation is where the start-up costs are paid: the device code
 * is loaded (an empty launch), the copy path is warmed (1 MB each way), the library's host threads are started, and the context's
 * buffer cache is filled with a first set of blocks — 128 MB + 8 MB of pinned staging memory and 3.4 GB of device memory (what the
 * levels of a 256 x 1000 or a 1024 x 600 progressive pass need; a batch that needs more replaces a block).  PGM_NO_STAGING_RESERVE=1 /
 * PGM_NO_DEVICE_RESERVE=1 in the environment: no such blocks, every batch allocates what it needs when it is created. */
int pgm_ctx_create(int device, pgm_ctx **out);
void pgm_ctx_destroy(pgm_ctx *ctx);
const char *pgm_last_error(void);
/* name of the device ("gfx950...") and CU count, for logs */
int pgm_ctx_device_info(pgm_ctx *ctx, char *name, size_t name_len, int *cu_count);

/* ---- (B1) alignGraphs  — replaces reference src/GraphAlign.h:200-534 ------------------
 * One call = njobs independent alignGraphs(g1[i], g2[i], model[i]) evaluations (the
 * reference calls it once per internal guide-tree node, ProgressiveAlignment.h:439, and
 * once per child in early refinement, :170).  Emission scores + ls_log (GraphAlign.h:146-163,
 * ls_log.h:22-59), border init (:205-234), fill (:238-260), end node (:264-280) and the
 * traceback incl. markAlternativePath (:166-198, :285-521) all run on the GPU. */
int pgm_align_graphs_batch(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1,
                           const pgm_graph *const *g2, const pgm_model *const *model,
                           const pgm_scores *scores, pgm_align_out *out);

/* Staged form of the same call (what pgm_align_graphs_batch does internally):
 *   create  : flatten + upload every job's inputs to HBM, allocate the DP storage
 *   run     : launch the prep, emission and fill(+traceback) kernels on the context's stream (asynchronous)
 *   fetch   : wait for the stream, then copy score / mappings from the batch's pinned result block (written by the kernel
 *             itself while it runs) into caller memory
 * bench.py times `run` with the inputs already resident.
 * A stage sizes its persistent grids for the WHOLE device: between `run` and the end of `fetch` no other batch may run on that
 * device (pgm_align_graphs_batch serialises the contexts of one device itself; callers of the staged form do it themselves). */
int pgm_align_batch_create(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1,
                           const pgm_graph *const *g2, const pgm_model *const *model,
                           const pgm_scores *scores, pgm_align_batch **out);
/* The same with flags.  PGM_BATCH_KEEP_MATRICES: every job's M, X, Y, W matrices are written to device memory so that
 * pgm_align_batch_read_matrices can return them (test hook).  Without it a job of two plain chains (sequence graph against
 * sequence graph) keeps one decision byte per cell instead of the 16 bytes of the four floats: its traceback walks the
 * decisions the fill took with the operands in registers (same tie rules, GraphAlign.h:382-411), the mappings are the same.
 * If, in addition, every profile column of both graphs is one-hot, uniform (1 / dim) or empty — the leaf level of a guide tree —
 * the emission scores (GraphAlign.h:146-163) are not stored either: S(y, x) depends on the classes of row y and column x alone,
 * a (dim + 2)^2 table per job (same operations on the same operands as for any other cell: bit-identical scores). */
#define PGM_BATCH_KEEP_MATRICES 1u
int pgm_align_batch_create_ex(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1,
                              const pgm_graph *const *g2, const pgm_model *const *model,
                              const pgm_scores *scores, uint32_t flags, pgm_align_batch **out);
/* Graphs whose node profiles already are in HBM (left there by pgm_merge_profiles_batch_ex with PGM_MERGE_RESIDENT): for job i,
 * res1[i].dev_sites != NULL replaces g1[i]->sites (which may then be NULL) by that device matrix, gathered through
 * res1[i].node_map — node v of the graph handed over (a CleanedGraph, src/CleanedGraph.h:39-146) has the profile in column
 * node_map[v] of the matrix (the merged graph it was cleaned from); node_map == NULL: column v.  res1 / res2 may be NULL. */
typedef struct pgm_site_ref {
    const double *dev_sites;
    const uint32_t *node_map; /* host array, n entries */
    uint32_t ncols;           /* columns of the device matrix: every node_map[v] (or n itself, without a map) must stay below it — PGM_ERR_INVALID otherwise */
} pgm_site_ref;
int pgm_align_batch_create_res(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1,
                               const pgm_graph *const *g2, const pgm_model *const *model,
                               const pgm_scores *scores, uint32_t flags, const pgm_site_ref *res1,
                               const pgm_site_ref *res2, pgm_align_batch **out);
int pgm_align_graphs_batch_res(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1,
                               const pgm_graph *const *g2, const pgm_model *const *model,
                               const pgm_scores *scores, const pgm_site_ref *res1,
                               const pgm_site_ref *res2, pgm_align_out *out);
int pgm_align_batch_run(pgm_ctx *ctx, pgm_align_batch *b);
int pgm_align_batch_fetch(pgm_ctx *ctx, pgm_align_batch *b, pgm_align_out *out);
void pgm_align_batch_destroy(pgm_ctx *ctx, pgm_align_batch *b);
/* Σ (n1-2)(n2-2) over the jobs of the batch (the GCUPS numerator). */
uint64_t pgm_align_batch_cells(const pgm_align_batch *b);
/* Run the batch `reps` times back to back and return the mean device time in milliseconds of
 * each kernel stage (prep, emission scores, fill), measured with HIP events on the context's
 * stream.  The tracebacks run inside the fill kernel (the worker that completes a job's last
 * band walks that job's path), so ms_fill includes them and ms_traceback is ~0. */
int pgm_align_batch_time(pgm_ctx *ctx, pgm_align_batch *b, int reps, float *ms_prep,
                         float *ms_emission, float *ms_fill, float *ms_traceback);
/* Mean device time (ms, HIP events on the library's stream) of the three stages — prep, emission scores, fill incl. the lean
 * jobs' kernel and all tracebacks — over the launches of pgm_align_batch_run that pgm_align_batch_fetch has completed since the
 * last reset, and their number: the stage times of exactly the steps a caller timed (bench.py), not of extra launches. */
int pgm_align_batch_stage_times(pgm_align_batch *b, int reset, float *ms_prep, float *ms_emission, float *ms_fill,
                                uint32_t *launches);
/* Timeline of the last completed launch: ticks[2 i] = the moment job i's last band was complete, ticks[2 i + 1] = the moment
 * its traceback was published, both in 10 ns ticks of the device's real-time counter (differences are meaningful, the origin
 * is not).  Tells which job's chain of sweeps, or which traceback, a batch waits for (tools/probe_jobtimes.py). */
int pgm_align_batch_job_times(pgm_ctx *ctx, pgm_align_batch *b, uint64_t *ticks);
/* Test hook for the hand-off time-out path: in the following launches band `band` of job `job` never publishes its progress
 * and a wavefront that waits for another gives up after `spin_limit` polls (0: the default); the band below then times out,
 * raises the batch's abort flag and every unfinished job reports PGM_ERR_DEVICE.  job = 0xFFFFFFFF switches it off. */
int pgm_align_batch_test_stall(pgm_align_batch *b, uint32_t job, uint32_t band, uint32_t spin_limit);
/* Test hook (host arithmetic only, no device call): how `cus` compute units are dealt to the launches of a batch's fill stage —
 * out5 = {lean queue, band queue, launch of the longest chains, main launch, traceback kernel beside the sweeps} — given each
 * queue's work in microseconds of one worker, its number of units (lean jobs, bands, items of the main launch, items of the
 * longest chains, tracebacks; ntb = 0: the tracebacks follow their launches) and the longest chain of sweeps in microseconds.
 * Every queue with work gets at least one CU and the shares never exceed `cus`. */
int pgm_test_cu_shares(uint32_t cus, double lean_cost, uint32_t nlean, double band_cost, uint32_t nbands, double rest_cost,
                       uint32_t nrest, uint32_t ncrit, double longest_chain, double tb_cost, uint32_t ntb, uint32_t *out5);
/* Test hook: copy one job's DP matrices back as the reference lays them out (n1 x n2,
 * column-major, element (y,x) at y + x*n1).  Only rows < n1-1 and columns < n2-1 are
 * defined (the END row/column are never written by the reference's fill either).
 * Any of M,X,Y,W may be NULL.  S is the emission matrix of GraphAlign.h:146-163 (PGM_ERR_INVALID for a job of two sequence graphs
 * in a batch without PGM_BATCH_KEEP_MATRICES: its scores were never stored). */
int pgm_align_batch_read_matrices(pgm_ctx *ctx, pgm_align_batch *b, uint32_t job, float *M,
                                  float *X, float *Y, float *W, float *S);

/* ---- (B2) DistanceFactoryAlign::alignPair — replaces reference
 * src/DistanceFactoryAlign.h:59-127 (called from computePwDistances, :29-56).
 * syms: concatenated sequences already mapped through ALPHABET::value() with negative
 * values replaced by 20 (the reference's quirk, :72,79); offs[nseq+1] delimits them.
 * score: (dim+1)x(dim+1) int32 column-major scoring matrix (DistanceFactoryAlign.cpp).
 * For pair p = (pi[p], pj[p]) with seq1 = sequence pi, seq2 = sequence pj the call returns
 * counts[p*dim*dim + s1 + dim*s2] (the reference's CountMatrix counts(s1,s2), column-major)
 * and gaps[p] (number of gap openings on the traceback path). */
int pgm_nw_pairs_batch(pgm_ctx *ctx, uint32_t dim, const int32_t *score, int32_t gap_open,
                       int32_t gap_extend, uint32_t nseq, const int8_t *syms,
                       const uint32_t *offs, uint32_t npairs, const uint32_t *pi,
                       const uint32_t *pj, int32_t *counts, uint32_t *gaps);
/* The same in two halves, so that a caller keeps TWO tiles of pairs in flight on one context: submit returns once the tile's
 * inputs are staged and its kernel and copies are enqueued (ticket 0 or 1); wait blocks until that tile's results are in
 * `counts` / `gaps` (the pointers given to submit, which must stay valid until then).  While the host waits for tile k and
 * copies its count matrices out, the kernel of tile k+1 runs.  A third submit without a wait is refused (PGM_ERR_INVALID).
 * flags: PGM_NW_REDUCED — `counts` receives two int32 per pair, (Σ_s counts(s,s), Σ counts), instead of the dim x dim matrix:
 * all the p-distance of DistanceFactoryML.h:143-146 reads (the default flow without --mldist).
 * Result buffers from pgm_host_alloc (pinned) are written by the D2H copy directly; any other memory goes through the
 * library's pinned staging block and one host copy.  One host thread per context. */
#define PGM_NW_REDUCED 1u
int pgm_nw_pairs_submit(pgm_ctx *ctx, uint32_t dim, const int32_t *score, int32_t gap_open,
                        int32_t gap_extend, uint32_t nseq, const int8_t *syms,
                        const uint32_t *offs, uint32_t npairs, const uint32_t *pi,
                        const uint32_t *pj, uint32_t flags, int32_t *counts, uint32_t *gaps, int *ticket);
int pgm_nw_pairs_wait(pgm_ctx *ctx, int ticket);
/* Device-time of the NW kernel of the tile the last pgm_nw_pairs_wait / pgm_nw_pairs_batch on this context completed (ms;
 * with two tiles in flight the second kernel's blocks wait for the first's to leave, so the times of overlapping tiles
 * overlap too). */
float pgm_nw_last_kernel_ms(pgm_ctx *ctx);
/* Pinned host memory for result buffers (NULL on failure). */
void *pgm_host_alloc(size_t bytes);
void pgm_host_free(void *p);

/* ---- (B3) CSProfile::createProfile — replaces reference src/CSProfile.cpp:175-225 ------
 * load: K context profiles with `ncols` window columns.  lprofiles: K x ncols x 21 doubles
 * (weights already applied, last entry of each row = 0, CSProfile.cpp:157-162), row-major
 * [k][col][symbol]; centre: K x 20 doubles = profiles_k.row(center); priors: K doubles
 * (= log(PRIOR)).  create: for each sequence (symbols 0..19, 20 = invalid) writes the
 * 20 x (L+2) column-major double profile of Model<AA>::Profile into out + out_offs[s].
 * tau[s] = model.divergence/0.8; pi = model.pi (20); p_uniform = model.P * (1/20), one vector of
 * 20 per sequence (nseq x 20, each leaf has its own model). */
int pgm_csprofile_load(pgm_ctx *ctx, uint32_t K, uint32_t ncols, const double *lprofiles,
                       const double *centre, const double *priors);
int pgm_csprofile_create_batch(pgm_ctx *ctx, uint32_t nseq, const int8_t *syms,
                               const uint32_t *offs, const double *tau, const double *pi,
                               const double *p_uniform, double *out, const uint64_t *out_offs);
/* The same with the profiles left in HBM (the leaf graphs of a resident pass, as pgm_resident_onehot for plain sequence graphs):
 * dev[s] = the 20 x (L_s + 2) matrix of sequence s in the context's resident arena (valid until pgm_resident_reset), to be named by
 * pgm_site_ref / pgm_merge_job; nothing is copied back. */
int pgm_csprofile_create_batch_res(pgm_ctx *ctx, uint32_t nseq, const int8_t *syms,
                                   const uint32_t *offs, const double *tau, const double *pi,
                                   const double *p_uniform, const double **dev);
float pgm_csprofile_last_kernel_ms(pgm_ctx *ctx);

/* ---- (f3) DistanceFactoryML::computeDistance / computeMLDist — replaces reference
 * src/DistanceFactoryML.h:66-190 for a batch of pairs (the tail of computePwDistances, src/DistanceFactoryAlign.h:49-51,
 * and of DistanceFactoryPrealigned, src/DistanceFactoryPrealigned.h:84-88).  The substitution model is handed over in the
 * eigen form the reference builds in ModelFactory (src/ModelFactory.h:48-67): P(d) = V diag(exp(sigma d)) V^-1, all
 * dim x dim matrices column-major double; dim <= 20: the kernel keeps P(d) and its two derivatives of one pair in the registers
 * of one wavefront (20 x 20 / 64 lanes); the 61-state codon models are refused (PGM_ERR_INVALID) and the host mirror keeps its
 * estimator for them (host/distance.cpp: 16 host threads; 8128 pairs of config 4 take 0.1 s).  min_dist / max_dist are the clamps of parseDistance
 * (src/ModelFactory.h:125), dist_max / var_max / var_min the constants of DistanceFactoryML.cpp:3-32.
 * counts[p * dim * dim + s1 + dim * s2], gaps[p], seqlen[p] = (L1 + L2) / 2 per pair -> dist[p], var[p]. */
typedef struct pgm_mldist_model {
    uint32_t dim;
    const double *Q, *V, *Vi, *sigma;
    double dist_max, var_max, var_min, cutoff_dist, min_dist, max_dist, indel_rate;
    int32_t mldist, mldist_gap; /* cmdlineopts.mldist_flag / mldist_gap_flag */
} pgm_mldist_model;
int pgm_mldist_batch(pgm_ctx *ctx, const pgm_mldist_model *model, uint32_t npairs, const int32_t *counts,
                     const uint32_t *gaps, const double *seqlen, double *dist, double *var);

/* ---- (f3) DistanceFactoryPrealigned pair counts — replaces the column loop of reference
 * src/DistanceFactoryPrealigned.h:34-90.  rows: nrows x ncols int8, row-major: ALPHABET::value() of a residue
 * (0..dim-1), -1 for a gap, -2 for a residue without a value.  Only values < 20 are counted (the reference's literal 20).
 * counts[p * dim * dim + s1 + dim * s2] and gaps[p] (gap openings) per pair (pi[p], pj[p]). */
int pgm_prealigned_counts_batch(pgm_ctx *ctx, uint32_t dim, uint32_t nrows, uint32_t ncols, const int8_t *rows,
                                uint32_t npairs, const uint32_t *pi, const uint32_t *pj, int32_t *counts,
                                uint32_t *gaps);
/* ---- (f4) DistanceFactoryAngle's cosine matrix — replaces the dense product of reference src/DistanceFactoryAngle.h:100
 * (`norms^-1 * counts2^T * counts2 * norms^-1`, the default initial distances without -a).  counts: nseq x ncols int32,
 * row-major (row i = the K-mer counts of sequence i; ncols = DIM^K).  cosine: nseq x nseq doubles, column-major,
 * cosine(i, j) = (sum_k (c_ik / |c_i|) c_jk) / |c_j| with k ascending (fp64, one multiply and one add per term), summed in the
 * depth blocks of the reference's GEMM — L1d / 128 terms each, L1d = 49152 bytes unless the environment names the size the
 * reference's host reports (PGM_EIGEN_L1D).  The matrix is NOT symmetric in its last bits; element (i, j) is at i + nseq j. */
int pgm_kmer_cosine(pgm_ctx *ctx, uint32_t nseq, uint32_t ncols, const int32_t *counts, double *cosine);
/* Device time of the kernel of the last pgm_mldist_batch / pgm_prealigned_counts_batch / pgm_kmer_cosine call on this context (ms). */
float pgm_dist_last_kernel_ms(pgm_ctx *ctx);

/* ---- (f1, numeric part) mergeGraphs' node profiles — replaces the P*g products and the L2 normalisation of reference
 * src/GraphAlign.h:569-620 for a batch of merges (one per internal guide-tree node of a level,
 * src/ProgressiveAlignment.h:449).  The caller walks the two mappings (the "unify" loops) and lists, per node of the
 * merged graph, the source node in g1 / g2 (PGM_GAP if none) and whether the g2 node is propagated with model1.P (the
 * reference does that for skipped g2 nodes, :591); the device returns the dim x nnodes profile matrix, bit-identical to
 * the reference's arithmetic (Eigen gemv association, sequential norm, multiplication by the reciprocal).  The edge lists
 * of the merged graph stay with the caller (host maps, O(edges)). */
typedef struct pgm_merge_job {
    uint32_t dim, n1, n2, nnodes;
    const double *sites1, *sites2; /* Graph::sites of the two (uncleaned) graphs, dim x n column-major */
    const double *P1, *P2;         /* model1.P, model2.P, dim x dim column-major */
    const uint32_t *k1, *k2;       /* nnodes each */
    const uint8_t *g2_with_P1;     /* nnodes */
    double *profiles;              /* out: dim x nnodes column-major */
} pgm_merge_job;
int pgm_merge_profiles_batch(pgm_ctx *ctx, uint32_t njobs, const pgm_merge_job *jobs);
/* The same with the results LEFT IN HBM (flags & PGM_MERGE_RESIDENT): dev_profiles[i] receives the device address of job i's
 * dim x nnodes matrix, jobs[i].profiles may be NULL and nothing is copied back; sites1 / sites2 of a job may themselves be such
 * addresses (the children's profiles never left the device).  The matrices stay valid until pgm_resident_reset(ctx), which
 * a caller issues at the start of a progressive pass (the memory is kept and handed out again); they are read by the
 * alignments (pgm_site_ref) and merges of the levels above. */
#define PGM_MERGE_RESIDENT 1u
int pgm_merge_profiles_batch_ex(pgm_ctx *ctx, uint32_t njobs, const pgm_merge_job *jobs, uint32_t flags,
                                const double **dev_profiles);
int pgm_resident_reset(pgm_ctx *ctx);
/* The leaf graphs of a pass built in HBM — replaces the profile matrix of reference src/SequenceGraph.h:101-109 for nseq
 * sequences: syms = ALPHABET::value() per residue, negative for a residue without a value (uniform 1 / dim column); dev[s]
 * receives the device address of sequence s's dim x (L + 2) matrix (START and END columns zero), valid until
 * pgm_resident_reset.  Used through pgm_site_ref and as sites1 / sites2 of pgm_merge_profiles_batch_ex. */
int pgm_resident_onehot(pgm_ctx *ctx, uint32_t dim, uint32_t nseq, const int8_t *syms, const uint32_t *offs,
                        const double **dev);
/* A resident matrix of another context copied into this one's resident memory (a pass sharded over several devices by subtree: the
 * parent of two subtrees needs both children's profiles on its device; hipMemcpyPeer between devices, a device-to-device copy within
 * one).  src must be an address pgm_merge_profiles_batch_ex / pgm_resident_onehot / pgm_resident_import returned for src_ctx and
 * still be valid there; *dst stays valid until pgm_resident_reset(ctx). */
int pgm_resident_import(pgm_ctx *ctx, pgm_ctx *src_ctx, const double *src, uint64_t count, const double **dst);
float pgm_merge_last_kernel_ms(pgm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PGM_HIP_H_ */
