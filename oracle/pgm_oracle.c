/*
 * pgm_oracle.c — TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C99, scalar, one thread)
 * of the three hot functions of acg-team/ProGraphMSA, over the flattened structs of
 * include/pgm_hip.h.  It is the checker for the HIP path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg); product code never links or calls it.
 *
 * Parity pin: the reference's own source cannot be compiled here (Eigen/TCLAP absent), so
 * this restatement is pinned end-to-end against outputs of the reference's prebuilt binary
 * (/root/reference/bin/ProGraphMSA_64) — see tests/golden/make_golden.py and
 * tests/test_oracle_golden.py.
 *
 * Compile with: gcc -O2 -std=c99 -ffp-contract=off (no -ffast-math): every float operation
 * below is meant to be a single IEEE-754 binary32 operation in the written order.
 *
 * Reference lines followed:
 *   pgmo_align_graphs   src/GraphAlign.h:145-163 (precomputeScores), :200-534 (alignGraphs),
 *                       :165-198 (markAlternativePath); src/Graph.h:180-248 (PredIterator);
 *                       src/ls_log.h:7-59 (ls_log / ls_log_add, SSE2 variant)
 *   pgmo_nw_pair        src/DistanceFactoryAlign.h:59-127 (alignPair)
 *   pgmo_csprofile_create  src/CSProfile.cpp:175-225 (createProfile)
 *   pgmo_merge_profiles    src/GraphAlign.h:569-620 (mergeGraphs: node profiles of the merged graph)
 *   pgmo_prealigned_counts src/DistanceFactoryPrealigned.h:34-90 (pair counts of an alignment)
 *   pgmo_kmer_cosine       src/DistanceFactoryAngle.h:100 (the cosine matrix of the k-mer count vectors)
 *   pgmo_mldist            src/DistanceFactoryML.h:66-190 (computeDistance / computeMLDist), src/ModelFactory.h:48-67, 104-127
 */
#include "../include/pgm_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_EDGE_COST 10000.0f /* Graph.h:18 */

/* ------------------------------------------------------------------------------------ */
/* PredIterator (Graph.h:180-248): regular predecessors in ascending order, then repeat   */
/* predecessors.                                                                          */
typedef struct {
    const pgm_graph *g;
    int32_t i, iend;   /* regular-edge cursor  */
    int32_t j, jend;   /* repeat-edge cursor   */
    float repeat_init, repeat_ext;
} pred_it;

static pred_it preds(const pgm_graph *g, uint32_t node, float repeat_init, float repeat_ext) {
    pred_it it;
    it.g = g;
    it.i = g->e_rowptr[node];
    it.iend = g->e_rowptr[node + 1];
    if (g->r_rowptr) {
        it.j = g->r_rowptr[node];
        it.jend = g->r_rowptr[node + 1];
    } else {
        it.j = it.jend = 0;
    }
    it.repeat_init = repeat_init;
    it.repeat_ext = repeat_ext;
    return it;
}
static int pred_ok(const pred_it *it) { return it->i < it->iend || it->j < it->jend; }
static void pred_next(pred_it *it) {
    if (it->i < it->iend) ++it->i; else ++it->j; /* Graph.h:196 */
}
static int pred_is_repeat(const pred_it *it) { return !(it->i < it->iend); }
static uint32_t pred_node(const pred_it *it) {
    return it->i < it->iend ? it->g->e_col[it->i] : it->g->r_col[it->j];
}
static float pred_value(const pred_it *it) { /* Graph.h:223-239 */
    if (it->i < it->iend) {
        float c = it->g->e_val[it->i];
        if (c == 0) return (float)INFINITY;
        return c + MAX_EDGE_COST;
    } else {
        uint32_t c = it->g->r_units[it->j];
        if (c == 0) return (float)INFINITY;
        return it->repeat_init + it->repeat_ext * (c - 1);
    }
}

/* ------------------------------------------------------------------------------------ */
/* ls_log_add, SSE2 float variant (ls_log.h:22-59).  Elements [0, len&~3) take the vector */
/* path (bit manipulation, op order t=b/(d-c); u=e+a; t=t+add; t=t+u), the last len%4     */
/* elements take the scalar frexp path (ls_log.h:7-19,57).                                 */
static float ls_log_vec1(float x, float add) {
    const float a = 2.44247459618085927548717403238913328776812604856113966238617812902399112761292613763080658235564;
    const float b = -4.2040783745848554315883301529007786406310628696382695994938550046831869207082846248658671;
    const float c = -0.72123729809042963774358701619456664388406302428056983119308906451199556380646306;
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int32_t ei = (int32_t)(bits >> 23) - 126;            /* _mm_srli_epi32 / _mm_sub_epi32   */
    float e = (float)ei;                                   /* _mm_cvtepi32_ps                  */
    uint32_t dbits = ((bits << 9) >> 9) ^ 0x3f000000u;
    float d;
    memcpy(&d, &dbits, 4);
    float t = b / (d - c);
    float u = e + a;
    t = t + add;
    t = t + u;
    return t;
}
static float ls_log_scalar(float x) {
    const float a = 2.44247459618085927548717403238913328776812604856113966238617812902399112761292613763080658235564;
    const float b = -4.2040783745848554315883301529007786406310628696382695994938550046831869207082846248658671;
    const float c = -0.72123729809042963774358701619456664388406302428056983119308906451199556380646306;
    int e;
    float d = frexpf(x, &e);
    return a + b / (d - c) + e;
}
static void ls_log_add(float *data, float add, size_t len) {
    size_t alen = len & ~(size_t)3;
    for (size_t i = 0; i < alen; ++i) data[i] = ls_log_vec1(data[i], add);
    for (size_t i = alen; i < len; ++i) data[i] = ls_log_scalar(data[i]) + add;
}

/* ------------------------------------------------------------------------------------ */
/* precomputeScores (GraphAlign.h:145-163).                                               */
/* S = (g1s^T (M^T g2s)) ./ ((g1s^T pi)(pi^T g2s)), all operands cast to float first.      */
/* Summation order: Eigen's order is not visible in the source; SURVEY §7 established       */
/* empirically against the binary that each output element is accumulated over k ascending */
/* from zero, multiply then add, with T = M^T g2s formed first.                             */
/* The two denominators g1s^T*pi and pi^T*g2s (GraphAlign.h:153) are matrix-vector products: Eigen's SSE kernel
 * (GeneralMatrixVector.h, row-major lhs) accumulates four lane sums over the packets of four floats, adds them
 * horizontally as (a0+a2)+(a1+a3) (SSE predux) and then adds the scalar tail (DIM = 61: element 60).  The numerators are
 * matrix-matrix products, whose kernel accumulates sequentially over k (precompute_scores below).  Pinned by the
 * fixture c2.m.out.fa (tests/golden/md5.json): with sequential denominators one cell of one alignment of that family
 * is 1 ulp off and a gap moves by one column; cd1..cd4 pin the 61-state tail handling. */
static float packet_dot(const float *u, const float *v, uint32_t D) {
    float l[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t aligned = D / 4 * 4;
    for (uint32_t k = 0; k < aligned; k += 4)
        for (int q = 0; q < 4; ++q) { float p = u[k + q] * v[k + q]; l[q] = l[q] + p; }
    float t02 = l[0] + l[2], t13 = l[1] + l[3];
    float acc = 0.0f + (t02 + t13);
    for (uint32_t k = aligned; k < D; ++k) { float p = u[k] * v[k]; acc = acc + p; }
    return acc;
}

static void precompute_scores(const pgm_graph *g1, const pgm_graph *g2, const pgm_model *model,
                              float match_init, float *S /* n1 x n2 col-major */) {
    const uint32_t D = g1->dim, n1 = g1->n, n2 = g2->n;
    float *g1s = (float *)malloc(sizeof(float) * D * n1);
    float *g2s = (float *)malloc(sizeof(float) * D * n2);
    float *Mf = (float *)malloc(sizeof(float) * D * D);
    float *pif = (float *)malloc(sizeof(float) * D);
    float *T = (float *)malloc(sizeof(float) * D * n2);
    float *a = (float *)malloc(sizeof(float) * n1);
    float *b = (float *)malloc(sizeof(float) * n2);
    for (size_t i = 0; i < (size_t)D * n1; ++i) g1s[i] = (float)g1->sites[i];
    for (size_t i = 0; i < (size_t)D * n2; ++i) g2s[i] = (float)g2->sites[i];
    for (size_t i = 0; i < (size_t)D * D; ++i) Mf[i] = (float)model->M[i];
    for (uint32_t i = 0; i < D; ++i) pif[i] = (float)model->pi[i];

    for (uint32_t x = 0; x < n2; ++x) {
        for (uint32_t k = 0; k < D; ++k) { /* T(k,x) = sum_j M(j,k) g2s(j,x) */
            float acc = 0.0f;
            for (uint32_t j = 0; j < D; ++j) {
                float p = Mf[j + D * k] * g2s[j + D * x];
                acc = acc + p;
            }
            T[k + D * x] = acc;
        }
        b[x] = packet_dot(pif, g2s + (size_t)D * x, D);
    }
    for (uint32_t y = 0; y < n1; ++y) {
        a[y] = packet_dot(g1s + (size_t)D * y, pif, D);
    }
    for (uint32_t x = 0; x < n2; ++x) {
        for (uint32_t y = 0; y < n1; ++y) {
            float acc = 0.0f;
            for (uint32_t k = 0; k < D; ++k) {
                float p = g1s[k + D * y] * T[k + D * x];
                acc = acc + p;
            }
            float den = a[y] * b[x];
            S[y + (size_t)n1 * x] = acc / den;
        }
    }
    ls_log_add(S, match_init, (size_t)n1 * n2);
    free(g1s); free(g2s); free(Mf); free(pif); free(T); free(a); free(b);
}

/* ------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t *m1, *m2;
    uint32_t len, cap;
} mapbuf;
static void push(mapbuf *mb, uint32_t y, uint32_t x) {
    if (mb->len == mb->cap) { /* cannot happen for capacity n1+n2, kept as a guard */
        mb->cap = mb->cap * 2 + 16;
        mb->m1 = (uint32_t *)realloc(mb->m1, sizeof(uint32_t) * mb->cap);
        mb->m2 = (uint32_t *)realloc(mb->m2, sizeof(uint32_t) * mb->cap);
    }
    mb->m1[mb->len] = y;
    mb->m2[mb->len] = x;
    mb->len++;
}

/* markAlternativePath (GraphAlign.h:165-198).  `first` selects which of the two mapping   */
/* vectors receives the node indices (the other receives -1).                               */
static void mark_alternative_path(uint32_t start, uint32_t end, const pgm_graph *g, mapbuf *mb,
                                  int first) {
    uint32_t len = end - start + 1;
    float *score = (float *)malloc(sizeof(float) * len);
    uint32_t *prev = (uint32_t *)malloc(sizeof(uint32_t) * len);
    for (uint32_t i = 0; i < len; ++i) { score[i] = -INFINITY; prev[i] = (uint32_t)-1; }
    score[0] = 0;
    for (uint32_t i = 1; i < len; ++i) {
        uint32_t real_ix = i + start;
        for (pred_it it = preds(g, real_ix, INFINITY, INFINITY); pred_ok(&it); pred_next(&it)) {
            uint32_t p = pred_node(&it);
            if (p >= start && p <= end) {
                uint32_t i2 = p - start;
                float v = pred_value(&it);
                if (score[i] <= score[i2] - v) {
                    score[i] = score[i2] - v;
                    prev[i] = i2;
                }
            }
        }
    }
    if (score[len - 1] > -INFINITY) {
        uint32_t i = prev[len - 1];
        while (i != 0) {
            if (first) push(mb, i + start, (uint32_t)-1);
            else push(mb, (uint32_t)-1, i + start);
            i = prev[i];
        }
    }
    free(score);
    free(prev);
}

static float fmax_std(float a, float b) { return (a < b) ? b : a; } /* std::max(a,b) */

/* alignGraphs (GraphAlign.h:200-534).  dbg[0..4] (optional) receive copies of M,X,Y,W,S. */
int pgmo_align_graphs(const pgm_graph *g1, const pgm_graph *g2, const pgm_model *model,
                      const pgm_scores *sc, pgm_align_out *out, float **dbg) {
    if (!g1 || !g2 || !model || !sc || !out || g1->n < 2 || g2->n < 2 || g1->dim != g2->dim)
        return PGM_ERR_INVALID;
    const pgm_scores s = *sc;
    const uint32_t n1 = g1->n, n2 = g2->n;
    const size_t N = (size_t)n1 * n2;
    const float minfty = -INFINITY;
    int status = PGM_OK;

    float *S = (float *)malloc(sizeof(float) * N);
    float *M = (float *)malloc(sizeof(float) * N);
    float *X = (float *)malloc(sizeof(float) * N);
    float *Y = (float *)malloc(sizeof(float) * N);
    float *W = (float *)malloc(sizeof(float) * N);
    if (!S || !M || !X || !Y || !W) return PGM_ERR_NOMEM;
    precompute_scores(g1, g2, model, s.match_init, S);
    for (size_t i = 0; i < N; ++i) M[i] = X[i] = Y[i] = W[i] = minfty;
#define AT(A, y, x) A[(size_t)(y) + (size_t)n1 * (x)]

    /* borders (GraphAlign.h:212-234) */
    AT(W, 0, 0) = s.start_init;
    for (uint32_t y = 1; y < n1 - 1; ++y) {
        float Sy = minfty;
        for (pred_it yit = preds(g1, y, s.repeat_init, s.repeat_ext); pred_ok(&yit); pred_next(&yit)) {
            uint32_t yp = pred_node(&yit);
            Sy = fmax_std(Sy, fmax_std(AT(Y, yp, 0) + s.gap_extend, AT(W, yp, 0) + s.start_gap) - pred_value(&yit));
        }
        AT(Y, y, 0) = Sy;
        AT(W, y, 0) = AT(Y, y, 0);
    }
    for (uint32_t x = 1; x < n2 - 1; ++x) {
        float Sx = minfty;
        for (pred_it xit = preds(g2, x, s.repeat_init, s.repeat_ext); pred_ok(&xit); pred_next(&xit)) {
            uint32_t xp = pred_node(&xit);
            Sx = fmax_std(Sx, fmax_std(AT(X, 0, xp) + s.gap_extend, AT(W, 0, xp) + s.start_gap) - pred_value(&xit));
        }
        AT(X, 0, x) = Sx;
        AT(W, 0, x) = AT(X, 0, x);
    }

    /* fill (GraphAlign.h:238-260) */
    for (uint32_t y = 1; y < n1 - 1; ++y) {
        for (uint32_t x = 1; x < n2 - 1; ++x) {
            for (pred_it yit = preds(g1, y, s.repeat_init, s.repeat_ext); pred_ok(&yit); pred_next(&yit)) {
                for (pred_it xit = preds(g2, x, s.repeat_init, s.repeat_ext); pred_ok(&xit); pred_next(&xit)) {
                    uint32_t yp = pred_node(&yit);
                    uint32_t xp = pred_node(&xit);
                    float yv = pred_value(&yit), xv = pred_value(&xit);
                    float Sm = AT(W, yp, xp) + AT(S, y, x) - yv - xv;
                    float Sx = fmax_std(AT(X, y, xp) + s.gap_extend, AT(W, y, xp) + s.gap_init) - xv;
                    float Sy = fmax_std(AT(Y, yp, x) + s.gap_extend, AT(W, yp, x) + s.gap_init) - yv;
                    float Sw = fmax_std(Sm, fmax_std(Sx, Sy));
                    AT(M, y, x) = fmax_std(AT(M, y, x), Sm);
                    AT(X, y, x) = fmax_std(AT(X, y, x), Sx);
                    AT(Y, y, x) = fmax_std(AT(Y, y, x), Sy);
                    AT(W, y, x) = fmax_std(AT(W, y, x), Sw);
                }
            }
        }
    }

    /* end node (GraphAlign.h:264-280) */
    float Wend = minfty;
    for (pred_it yit = preds(g1, n1 - 1, s.repeat_init, s.repeat_ext); pred_ok(&yit); pred_next(&yit)) {
        for (pred_it xit = preds(g2, n2 - 1, s.repeat_init, s.repeat_ext); pred_ok(&xit); pred_next(&xit)) {
            uint32_t yp = pred_node(&yit), xp = pred_node(&xit);
            float yv = pred_value(&yit), xv = pred_value(&xit);
            if (xp == 0 && yp == 0) {
                Wend = fmax_std(s.end_skip - yv - xv, Wend);
            } else {
                Wend = fmax_std(AT(X, yp, xp) + s.end_gap - yv - xv, Wend);
                Wend = fmax_std(AT(Y, yp, xp) + s.end_gap - yv - xv, Wend);
                Wend = fmax_std(AT(M, yp, xp) + s.end_match - yv - xv, Wend);
            }
        }
    }

    /* backtracking (GraphAlign.h:285-521) */
    out->score = Wend;
    out->n_tr_indels = 0;
    enum { State_m, State_x, State_y } current_state = State_m, next_state = State_m;
    float current_score = minfty;
    uint32_t y = n1 - 1, x = n2 - 1;
    mapbuf mb;
    mb.cap = n1 + n2;
    mb.len = 0;
    mb.m1 = (uint32_t *)malloc(sizeof(uint32_t) * mb.cap);
    mb.m2 = (uint32_t *)malloc(sizeof(uint32_t) * mb.cap);
    push(&mb, n1 - 1, n2 - 1);

    int tr_indel_x = 0, tr_indel_y = 0;
    float best_match = INFINITY;
    for (pred_it yit = preds(g1, n1 - 1, s.repeat_init, s.repeat_ext); pred_ok(&yit); pred_next(&yit)) {
        for (pred_it xit = preds(g2, n2 - 1, s.repeat_init, s.repeat_ext); pred_ok(&xit); pred_next(&xit)) {
            uint32_t yp = pred_node(&yit), xp = pred_node(&xit);
            float yv = pred_value(&yit), xv = pred_value(&xit);
            float d;
            d = fabsf(Wend - (AT(M, yp, xp) + s.end_match - yv - xv));
            if (best_match > d) {
                best_match = d;
                tr_indel_x = pred_is_repeat(&xit); tr_indel_y = pred_is_repeat(&yit);
                current_score = AT(M, yp, xp); current_state = State_m; y = yp; x = xp;
            }
            d = fabsf(Wend - (AT(Y, yp, xp) + s.end_gap - yv - xv));
            if (best_match > d) {
                best_match = d;
                tr_indel_x = pred_is_repeat(&xit); tr_indel_y = pred_is_repeat(&yit);
                current_score = AT(Y, yp, xp); current_state = State_y; y = yp; x = xp;
            }
            d = fabsf(Wend - (AT(X, yp, xp) + s.end_gap - yv - xv));
            if (best_match > d) {
                best_match = d;
                tr_indel_x = pred_is_repeat(&xit); tr_indel_y = pred_is_repeat(&yit);
                current_score = AT(X, yp, xp); current_state = State_x; y = yp; x = xp;
            }
            d = fabsf(Wend - (s.end_skip - yv - xv));
            if (xp == 0 && yp == 0 && best_match > d) {
                best_match = d;
                tr_indel_x = pred_is_repeat(&xit); tr_indel_y = pred_is_repeat(&yit);
                y = yp; x = xp;
            }
        }
    }
    out->n_tr_indels += tr_indel_x + tr_indel_y;
    if (tr_indel_y) mark_alternative_path(y, n1 - 1, g1, &mb, 1);
    if (tr_indel_x) mark_alternative_path(x, n2 - 1, g2, &mb, 0);

    if (x != 0 || y != 0) {
        if (current_state == State_m) push(&mb, y, x);
        else if (current_state == State_x) push(&mb, (uint32_t)-1, x);
        else push(&mb, y, (uint32_t)-1);
    }

    float next_score = INFINITY;
    uint32_t next_x = (uint32_t)-1, next_y = (uint32_t)-1;
    /* the walk terminates after at most n1+n2 steps on a DAG; guard against corrupt input */
    uint32_t guard = 0;
    while ((x != 0 || y != 0) && status == PGM_OK) {
        if (++guard > n1 + n2 + 4) { status = PGM_ERR_BACKTRACK; break; }
        best_match = INFINITY;

#define PICK_STATE_FROM_W()                                                          \
    if (next_x != 0 || next_y != 0) {                                                \
        if (AT(W, next_y, next_x) == AT(M, next_y, next_x)) {                        \
            next_score = AT(M, next_y, next_x); next_state = State_m;                \
        } else if (AT(W, next_y, next_x) == AT(Y, next_y, next_x)) {                 \
            next_score = AT(Y, next_y, next_x); next_state = State_y;                \
        } else if (AT(W, next_y, next_x) == AT(X, next_y, next_x)) {                 \
            next_score = AT(X, next_y, next_x); next_state = State_x;                \
        } else {                                                                     \
            status = PGM_ERR_BACKTRACK; /* error("backtracking failed") */           \
        }                                                                            \
    }

        if (current_state == State_y) {
            for (pred_it yit = preds(g1, y, s.repeat_init, s.repeat_ext); pred_ok(&yit); pred_next(&yit)) {
                uint32_t yp = pred_node(&yit);
                float yv = pred_value(&yit);
                float d = fabsf(current_score - (AT(Y, yp, x) + s.gap_extend - yv));
                if (best_match > d) {
                    best_match = d;
                    tr_indel_x = 0; tr_indel_y = pred_is_repeat(&yit);
                    next_x = x; next_y = yp;
                    next_score = AT(Y, next_y, next_x); next_state = State_y;
                }
                d = fabsf(current_score - (AT(W, yp, x) + s.gap_init - yv));
                if (best_match > d) {
                    best_match = d;
                    tr_indel_x = 0; tr_indel_y = pred_is_repeat(&yit);
                    next_x = x; next_y = yp;
                    PICK_STATE_FROM_W();
                }
            }
        }
        if (current_state == State_x) {
            for (pred_it xit = preds(g2, x, s.repeat_init, s.repeat_ext); pred_ok(&xit); pred_next(&xit)) {
                uint32_t xp = pred_node(&xit);
                float xv = pred_value(&xit);
                float d = fabsf(current_score - (AT(X, y, xp) + s.gap_extend - xv));
                if (best_match > d) {
                    best_match = d;
                    tr_indel_x = pred_is_repeat(&xit); tr_indel_y = 0;
                    next_x = xp; next_y = y;
                    next_score = AT(X, next_y, next_x); next_state = State_x;
                }
                d = fabsf(current_score - (AT(W, y, xp) + s.gap_init - xv));
                if (best_match > d) {
                    best_match = d;
                    tr_indel_x = pred_is_repeat(&xit); tr_indel_y = 0;
                    next_x = xp; next_y = y;
                    PICK_STATE_FROM_W();
                }
            }
        }
        if (current_state == State_m) {
            for (pred_it yit = preds(g1, y, s.repeat_init, s.repeat_ext); pred_ok(&yit); pred_next(&yit)) {
                for (pred_it xit = preds(g2, x, s.repeat_init, s.repeat_ext); pred_ok(&xit); pred_next(&xit)) {
                    uint32_t yp = pred_node(&yit), xp = pred_node(&xit);
                    float yv = pred_value(&yit), xv = pred_value(&xit);
                    float d = fabsf(current_score - (AT(W, yp, xp) + AT(S, y, x) - yv - xv));
                    if (best_match > d) {
                        best_match = d;
                        tr_indel_x = pred_is_repeat(&xit); tr_indel_y = pred_is_repeat(&yit);
                        next_y = yp; next_x = xp;
                        PICK_STATE_FROM_W();
                    }
                }
            }
        }
        if (status != PGM_OK) break;
        out->n_tr_indels += tr_indel_x + tr_indel_y;
        if (tr_indel_y) mark_alternative_path(next_y, y, g1, &mb, 1);
        if (tr_indel_x) mark_alternative_path(next_x, x, g2, &mb, 0);

        x = next_x; y = next_y;
        current_state = next_state;
        current_score = next_score;
        if (x != 0 || y != 0) {
            if (current_state == State_m) push(&mb, y, x);
            else if (current_state == State_x) push(&mb, (uint32_t)-1, x);
            else push(&mb, y, (uint32_t)-1);
        }
    }
    push(&mb, 0, 0);

    /* reverse into the caller's buffers */
    out->len = mb.len;
    out->status = status;
    for (uint32_t i = 0; i < mb.len; ++i) {
        out->map1[i] = mb.m1[mb.len - 1 - i];
        out->map2[i] = mb.m2[mb.len - 1 - i];
    }
    free(mb.m1);
    free(mb.m2);
#undef AT
#undef PICK_STATE_FROM_W
    if (dbg) {
        float *src[5] = {M, X, Y, W, S};
        for (int k = 0; k < 5; ++k)
            if (dbg[k]) memcpy(dbg[k], src[k], sizeof(float) * N);
    }
    free(S); free(M); free(X); free(Y); free(W);
    return status;
}

int pgmo_align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                            const pgm_model *const *model, const pgm_scores *scores,
                            pgm_align_out *out) {
    int rc = PGM_OK;
    for (uint32_t i = 0; i < njobs; ++i) {
        int r = pgmo_align_graphs(g1[i], g2[i], model[i], &scores[i], &out[i], NULL);
        if (r != PGM_OK) rc = r;
    }
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* alignPair (DistanceFactoryAlign.h:59-127): integer Gotoh, traceback counts.             */
/* s1/s2: symbols already mapped (negative -> 20).  counts: dim x dim col-major int32,     */
/* counts(s1,s2) at s1 + dim*s2.                                                           */
int pgmo_nw_pair(uint32_t dim, const int32_t *score, int32_t gap_open, int32_t gap_extend,
                 const int8_t *seq1, uint32_t L1, const int8_t *seq2, uint32_t L2,
                 int32_t *counts, uint32_t *gaps_out) {
    const int minfty = -10000;
    const size_t R = (size_t)L2 + 1, C = (size_t)L1 + 1; /* rows = seq2, cols = seq1 */
    int *W = (int *)malloc(sizeof(int) * R * C);
    int *X = (int *)malloc(sizeof(int) * R * C);
    int *Y = (int *)malloc(sizeof(int) * R * C);
    if (!W || !X || !Y) return PGM_ERR_NOMEM;
    const uint32_t sd = dim + 1;
#define AT(A, y, x) A[(size_t)(y) + R * (size_t)(x)]
#define SC(a, b) score[(a) + sd * (b)] /* scoring_matrix(a,b), column-major */
    AT(W, 0, 0) = 0;
    AT(X, 0, 0) = 0; /* uninitialised in the reference, never read */
    AT(Y, 0, 0) = 0;
    for (uint32_t x = 1; x < L1 + 1; ++x) {
        AT(X, 0, x) = AT(W, 0, x) = gap_open + ((int)x - 1) * gap_extend;
        AT(Y, 0, x) = minfty;
    }
    for (uint32_t y = 1; y < L2 + 1; ++y) {
        AT(Y, y, 0) = AT(W, y, 0) = gap_open + ((int)y - 1) * gap_extend;
        AT(X, y, 0) = minfty;
    }
    for (uint32_t y = 1; y < L2 + 1; ++y) {
        for (uint32_t x = 1; x < L1 + 1; ++x) {
            int w = AT(W, y - 1, x - 1) + SC(seq2[y - 1], seq1[x - 1]);
            int xx = AT(X, y, x - 1) + gap_extend;
            int t = AT(W, y, x - 1) + gap_open;
            if (t > xx) xx = t;
            int yy = AT(Y, y - 1, x) + gap_extend;
            t = AT(W, y - 1, x) + gap_open;
            if (t > yy) yy = t;
            AT(X, y, x) = xx;
            AT(Y, y, x) = yy;
            t = xx > yy ? xx : yy;
            AT(W, y, x) = t > w ? t : w;
        }
    }
    memset(counts, 0, sizeof(int32_t) * dim * dim);
    uint32_t gaps = 0;
    int gap_opened1 = 0, gap_opened2 = 0, status = PGM_OK;
    for (uint32_t y = L2, x = L1; y != 0 && x != 0;) {
        int a = seq1[x - 1], b = seq2[y - 1];
        if (AT(W, y, x) == AT(W, y - 1, x - 1) + SC(b, a)) {
            if (a < (int)dim && b < (int)dim) ++counts[a + dim * b];
            gap_opened1 = 0; gap_opened2 = 0;
            --x; --y;
        } else if (AT(W, y, x) == AT(X, y, x)) {
            if (!gap_opened1) ++gaps;
            gap_opened1 = 1; gap_opened2 = 0;
            --x;
        } else if (AT(W, y, x) == AT(Y, y, x)) {
            if (!gap_opened2) ++gaps;
            gap_opened1 = 0; gap_opened2 = 1;
            --y;
        } else {
            status = PGM_ERR_BACKTRACK; /* error("error while backtracking") */
            break;
        }
    }
#undef AT
#undef SC
    *gaps_out = gaps;
    free(W); free(X); free(Y);
    return status;
}

int pgmo_nw_pairs_batch(uint32_t dim, const int32_t *score, int32_t gap_open, int32_t gap_extend,
                        uint32_t nseq, const int8_t *syms, const uint32_t *offs, uint32_t npairs,
                        const uint32_t *pi, const uint32_t *pj, int32_t *counts, uint32_t *gaps) {
    (void)nseq;
    int rc = PGM_OK;
    for (uint32_t p = 0; p < npairs; ++p) {
        uint32_t i = pi[p], j = pj[p];
        int r = pgmo_nw_pair(dim, score, gap_open, gap_extend, syms + offs[i], offs[i + 1] - offs[i],
                             syms + offs[j], offs[j + 1] - offs[j], counts + (size_t)p * dim * dim,
                             &gaps[p]);
        if (r != PGM_OK) rc = r;
    }
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* createProfile (CSProfile.cpp:175-225).  lprofiles [k][col][21], centre [k][20],         */
/* priors[k]; seq symbols 0..19 or 20 (invalid).  out: 20 x (L+2) column-major doubles.    */
int pgmo_csprofile_create(uint32_t K, uint32_t ncols, const double *lprofiles, const double *centre,
                          const double *priors, const int8_t *seq, uint32_t L, double tau,
                          const double *pi, const double *p_uniform, double *out) {
    const int center = (int)ncols / 2;
    double *profile = (double *)calloc((size_t)(L + 2) * 20, sizeof(double)); /* [row][20] */
    if (!profile) return PGM_ERR_NOMEM;
    for (uint32_t k = 0; k < K; ++k) {
        const double *lp = lprofiles + (size_t)k * ncols * 21;
        const double *ce = centre + (size_t)k * 20;
        for (uint32_t i = 0; i < L; ++i) {
            double pk = priors[k];
            for (int j = -center; j <= center; ++j) {
                if ((int)i + j >= 0 && i + j < L) {
                    int cj = seq[i + j];
                    pk += lp[(size_t)(j + center) * 21 + cj];
                }
            }
            double e = exp(pk);
            double *row = profile + (size_t)(i + 1) * 20;
            for (int a = 0; a < 20; ++a) row[a] += ce[a] * e;
        }
    }
    for (uint32_t i = 0; i < L; ++i) {
        int c = seq[i];
        double *row = profile + (size_t)(i + 1) * 20;
        double sum = 0;
        for (int a = 0; a < 20; ++a) sum += row[a];
        if (sum <= 0) {
            for (int a = 0; a < 20; ++a) row[a] = p_uniform[a];
        } else if (c < 0 || c > 19) {
            double f = 1.0 / sum;
            for (int a = 0; a < 20; ++a) row[a] *= f;
            for (int a = 0; a < 20; ++a) row[a] *= (1.0 / 20.0) * (1.0 / pi[a]);
        } else {
            double f = tau / sum;
            for (int a = 0; a < 20; ++a) row[a] *= f;
            row[c] += 1.0 - tau;
            if (row[c] <= 0.0) row[c] = 1e-3;
            for (int a = 0; a < 20; ++a) row[a] *= (1.0 / 20.0) * (1.0 / pi[a]);
        }
    }
    /* transpose to 20 x (L+2) column-major == [node][20] contiguous: same memory image */
    memcpy(out, profile, sizeof(double) * (size_t)(L + 2) * 20);
    free(profile);
    return PGM_OK;
}

/* ------------------------------------------------------------------------------------ */
/* mergeGraphs, node profiles (GraphAlign.h:569-620).  The caller has walked the two mappings (the "unify" loops) and lists
 * per node of the merged graph its source nodes k1 / k2 (PGM_GAP = none) and whether a g2 node is propagated with model1.P
 * (the reference does that for SKIPPED g2 nodes, :591).  Per node:
 *     p = P1 g1[k1]                      (k2 none)
 *     p = Pb g2[k2]                      (k1 none; Pb = P1 for a skipped node, P2 for a mapped one: :591, :612)
 *     p = (P1 g1[k1]) .* (P2 g2[k2])     (both: :601)
 *     node = p.norm() == 0 ? p : p.normalized()
 * in double.  Association of the matrix-vector product: Eigen 3.0-3.2's column-major gemv as the reference binary shows it
 * (four columns at a time, res += (c0 v0 + c1 v1) + (c2 v2 + c3 v3), leftover columns one by one); the norm is the
 * sequential sum of squares, sqrt; normalized() multiplies by the reciprocal (Eigen's scalar quotient for floating
 * point).  Pinned through the whole-FASTA fixtures (every merged profile enters the next alignGraphs' emission scores). */
static void pgmo_gemv(const double *P, const double *v, uint32_t n, double *out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = 0.0;
    uint32_t j = 0;
    for (; j + 4 <= n; j += 4) {
        const double *c0 = P + (size_t)n * j, *c1 = c0 + n, *c2 = c1 + n, *c3 = c2 + n;
        const double v0 = v[j], v1 = v[j + 1], v2 = v[j + 2], v3 = v[j + 3];
        for (uint32_t i = 0; i < n; ++i) out[i] += (c0[i] * v0 + c1[i] * v1) + (c2[i] * v2 + c3[i] * v3);
    }
    for (; j < n; ++j) {
        const double *pc = P + (size_t)n * j;
        for (uint32_t i = 0; i < n; ++i) out[i] += pc[i] * v[j];
    }
}

int pgmo_merge_profiles(const pgm_merge_job *job) {
    if (!job || !job->sites1 || !job->sites2 || !job->P1 || !job->P2 || !job->k1 || !job->k2 || !job->g2_with_P1 || !job->profiles ||
        job->dim == 0 || job->dim > 64)
        return PGM_ERR_INVALID;
    const uint32_t D = job->dim;
    double a[64], b[64];
    for (uint32_t v = 0; v < job->nnodes; ++v) {
        const uint32_t k1 = job->k1[v], k2 = job->k2[v];
        if ((k1 == PGM_GAP && k2 == PGM_GAP) || (k1 != PGM_GAP && k1 >= job->n1) || (k2 != PGM_GAP && k2 >= job->n2)) return PGM_ERR_INVALID;
        double *p = job->profiles + (size_t)D * v;
        if (k1 != PGM_GAP) pgmo_gemv(job->P1, job->sites1 + (size_t)D * k1, D, a);
        if (k2 != PGM_GAP) pgmo_gemv(job->g2_with_P1[v] ? job->P1 : job->P2, job->sites2 + (size_t)D * k2, D, b);
        for (uint32_t i = 0; i < D; ++i) p[i] = (k1 != PGM_GAP && k2 != PGM_GAP) ? a[i] * b[i] : (k1 != PGM_GAP ? a[i] : b[i]);
        double ss = 0.0;
        for (uint32_t i = 0; i < D; ++i) ss += p[i] * p[i];
        const double nrm = sqrt(ss);
        if (nrm != 0.0) {
            const double inv = 1.0 / nrm;
            for (uint32_t i = 0; i < D; ++i) p[i] *= inv;
        }
    }
    return PGM_OK;
}

/* ------------------------------------------------------------------------------------ */
/* DistanceFactoryPrealigned::computePwDistances, the scan of two aligned rows (DistanceFactoryPrealigned.h:49-79).
 * rows: nrows x ncols, value() of a residue (0..dim-1), -1 gap, -2 residue without a value.  Only values < 20 are
 * counted — the reference's literal 20, for every alphabet (:62).  counts(c1, c2) column-major: element c1 + dim c2. */
int pgmo_prealigned_counts(uint32_t dim, uint32_t nrows, uint32_t ncols, const int8_t *rows, uint32_t npairs, const uint32_t *pi,
                           const uint32_t *pj, int32_t *counts, uint32_t *gaps) {
    if (!rows || !pi || !pj || !counts || !gaps || dim == 0) return PGM_ERR_INVALID;
    for (uint32_t p = 0; p < npairs; ++p) {
        if (pi[p] >= nrows || pj[p] >= nrows) return PGM_ERR_INVALID;
        const int8_t *s1 = rows + (size_t)pi[p] * ncols, *s2 = rows + (size_t)pj[p] * ncols;
        int32_t *c = counts + (size_t)p * dim * dim;
        memset(c, 0, sizeof(int32_t) * (size_t)dim * dim);
        uint32_t g = 0;
        int gap_opened1 = 0, gap_opened2 = 0;
        for (uint32_t k = 0; k < ncols; ++k) {
            const int g1 = s1[k] == -1, g2 = s2[k] == -1;
            if (!g1 && !g2) {
                const int c1 = s1[k], c2 = s2[k];
                if (c1 >= 0 && c1 < 20 && c2 >= 0 && c2 < 20) ++c[(uint32_t)c1 + dim * (uint32_t)c2];
                gap_opened1 = 0; gap_opened2 = 0;
            } else if (g1 && g2) {
                /* skip */
            } else if (!g1 && !gap_opened1) {
                ++g; gap_opened1 = 1; gap_opened2 = 0;
            } else if (!g2 && !gap_opened2) {
                ++g; gap_opened1 = 0; gap_opened2 = 1;
            }
        }
        gaps[p] = g;
    }
    return PGM_OK;
}

/* ------------------------------------------------------------------------------------ */
/* DistanceFactoryML::computeDistance + computeMLDist (DistanceFactoryML.h:66-190) for the pairs of a batch.  The model is
 * handed over in the eigen form ModelFactory builds (ModelFactory.h:48-67): getModel(d) = parseDistance (with -m / -M: the
 * distance itself, NaN -> 5.2, clamped to [min_dist, max_dist], :104-127), P = V diag(exp(sigma d)) V^-1.  Every matrix
 * product accumulates k = 0..n-1 from zero (one multiply, one add per term), f and f' add the n^2 entries in storage
 * order: the order of the host mirror (host/distance.cpp, host/model_factory.cpp), whose distances agree with the
 * reference binary's to the 6 significant digits its newick output prints (tests/golden: nw_pairs.json, *.nw_ml.tree). */
static void pgmo_matmul(const double *A, const double *B, uint32_t n, double *C) {
    for (size_t i = 0; i < (size_t)n * n; ++i) C[i] = 0.0;
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t k = 0; k < n; ++k) {
            const double bk = B[k + n * j];
            for (uint32_t i = 0; i < n; ++i) C[i + n * j] += A[i + n * k] * bk;
        }
}
static void pgmo_model_P(const pgm_mldist_model *m, double distance, double *P, double *tmp, double *e) {
    const uint32_t n = m->dim;
    distance = distance > 0.0 ? distance : 0.0;                 /* std::max(0.0, distance): NaN stays NaN */
    if (distance != distance) distance = 5.2;
    double d = distance < m->max_dist ? distance : m->max_dist;  /* std::min(model.distance, max_dist) */
    d = d > m->min_dist ? d : m->min_dist;                       /* std::max(.., min_dist) */
    for (uint32_t k = 0; k < n; ++k) e[k] = exp(m->sigma[k] * d);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t k = 0; k < n; ++k) tmp[i + n * k] = m->V[i + n * k] * e[k];
    pgmo_matmul(tmp, m->Vi, n, P);
}

int pgmo_mldist(const pgm_mldist_model *m, uint32_t npairs, const int32_t *counts, const uint32_t *gaps, const double *seqlen,
                double *dist_out, double *var_out) {
    if (!m || !m->Q || !m->V || !m->Vi || !m->sigma || !counts || !gaps || !seqlen || !dist_out || !var_out || m->dim == 0 || m->dim > 64)
        return PGM_ERR_INVALID;
    const uint32_t n = m->dim, nn = n * n;
    double *P = (double *)malloc(sizeof(double) * nn * 4 + sizeof(double) * n);
    if (!P) return PGM_ERR_NOMEM;
    double *pp = P + nn, *ppp = pp + nn, *tmp = ppp + nn, *e = tmp + nn;
    const double EPSILON = 1e-5;
    const uint32_t MAXITER = 20;
    for (uint32_t pr = 0; pr < npairs; ++pr) {
        const int32_t *c = counts + (size_t)pr * nn;
        double ident = 0, total = 0;
        for (uint32_t i = 0; i < n; ++i) ident += c[i + n * i];
        for (uint32_t i = 0; i < nn; ++i) total += c[i];
        double dist0 = 1.0 - ident / total, dist, var;
        if (m->mldist || m->mldist_gap) {
            if (total == 0 || dist0 > 0.85) { dist = dist0 = m->dist_max; var = m->var_max; }
            else { dist = dist0 = -log(1.0 - dist0 - 0.2 * dist0 * dist0); var = dist / total; }
            if (total > 0 && ident != total) {
                /* computeMLDist (:66-135) */
                const double d00 = dist, v00 = var;
                double dist_min = 0, dist_max = INFINITY, delta = 1;
                uint32_t iteration = 0;
                while (fabs(delta) > EPSILON) {
                    if (iteration > MAXITER) {
                        if (dist_max == INFINITY) { dist = m->dist_max; var = m->var_max; }
                        else { dist = d00; var = v00; }
                        break;
                    }
                    pgmo_model_P(m, dist, P, tmp, e);
                    pgmo_matmul(m->Q, P, n, pp);
                    pgmo_matmul(m->Q, pp, n, ppp);
                    double f = 0, ff = 0;
                    for (uint32_t i = 0; i < nn; ++i) {
                        const double ci = c[i];
                        f += ci * pp[i] / P[i];
                        ff += (ci * (ppp[i] * P[i] - pp[i] * pp[i])) / (P[i] * P[i]);
                    }
                    if (m->mldist_gap) {
                        const double grate = m->indel_rate * seqlen[pr] * dist;
                        f += (-grate + gaps[pr]) / dist;
                        ff += -(double)gaps[pr] / (dist * dist);
                    }
                    var = -1.0 / ff;
                    if (f > 0) dist_min = dist_min > dist ? dist_min : dist;
                    else dist_max = dist_max < dist ? dist_max : dist;
                    double new_dist = dist - f / ff;
                    if (!(new_dist < dist_max && new_dist > dist_min)) {
                        const double upper = (dist_max == INFINITY) ? dist * 3 : dist_max;
                        new_dist = (upper + dist_min) / 2.0;
                    }
                    delta = 1.0 - new_dist / dist;
                    dist = new_dist;
                    ++iteration;
                }
            }
        } else {
            if (total == 0) { dist = dist0 = 1.0; var = m->var_max; }
            else { dist = dist0; var = dist0 / total; }
        }
        if (!(dist < m->dist_max)) { dist = m->dist_max; var = m->var_max; }
        if (dist > m->cutoff_dist) dist = m->cutoff_dist;
        if (var < m->var_min) var = m->var_min;
        if (!(var < m->var_max)) var = m->var_max;
        dist_out[pr] = dist;
        var_out[pr] = var;
    }
    free(P);
    return PGM_OK;
}

/* ------------------------------------------------------------------------------------ */
/* DistanceFactoryAngle, the dense product (DistanceFactoryAngle.h:100): norms^-1 * counts2^T * counts2 * norms^-1, evaluated
 * left to right.  counts: nseq x ncols, row-major.  out(i, j), column-major = (sum_k (c_ik * inv_i) * c_jk) * inv_j with k
 * ascending.  PARITY NOTE: the reference's sum runs in the order of Eigen's GEMM micro-kernel, which the sources do not
 * show; with this order the BioNJ tree of the binary is reproduced for about two families in three (every BioNJ run ends in an
 * exact tie that the last bits of the distances decide) — tests pin the fixtures where it is, DESIGN.md states the rate. */
int pgmo_kmer_cosine(uint32_t nseq, uint32_t ncols, const int32_t *counts, double *out) {
    if (!counts || !out || ncols == 0) return PGM_ERR_INVALID;
    double *inv = (double *)malloc(sizeof(double) * (nseq ? nseq : 1));
    if (!inv) return PGM_ERR_NOMEM;
    for (uint32_t i = 0; i < nseq; ++i) {
        unsigned long long ss = 0;
        for (uint32_t k = 0; k < ncols; ++k) { const long long c = counts[(size_t)i * ncols + k]; ss += (unsigned long long)(c * c); }
        inv[i] = 1.0 / sqrt((double)ss);
    }
    /* Eigen's GEMM (3.1, SSE2, x86-64) sums k in ascending order inside a depth block of kc = L1d / 128 terms — 2 nr RhsProgress
     * sizeof(double) = 2 * 4 * 2 * 8 bytes of the packed right-hand panel per k, computeProductBlockingSizes — and adds the blocks'
     * sums to the result one after the other; L1d is read from cpuid at run time.  48 KB (kc = 384: the golden files' host; the
     * 400 amino-acid 2-mers are two blocks) unless PGM_EIGEN_L1D names another size in bytes.  Pinned by tests/golden/
     * angle_trees.json (12 of 12 trees; 9 with one block) and a campaign of 60 random families against the reference binary. */
    const char *l1e = getenv("PGM_EIGEN_L1D");
    const long l1d = l1e ? atol(l1e) : 49152;
    const uint32_t kc = (uint32_t)(l1d / 128 > 0 ? l1d / 128 : 1);
    for (uint32_t j = 0; j < nseq; ++j)
        for (uint32_t i = 0; i < nseq; ++i) {
            double res = 0.0;
            for (uint32_t k0 = 0; k0 < ncols; k0 += kc) {
                double part = 0.0;
                for (uint32_t k = k0; k < ncols && k - k0 < kc; ++k) part += ((double)counts[(size_t)i * ncols + k] * inv[i]) * (double)counts[(size_t)j * ncols + k];
                res = res + part;
            }
            out[(size_t)i + (size_t)nseq * j] = res * inv[j];
        }
    free(inv);
    return PGM_OK;
}
