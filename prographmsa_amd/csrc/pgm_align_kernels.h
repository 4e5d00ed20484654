// pgm_align_kernels.h — HIP kernels (gfx950 / CDNA4, wave64) for the graph-vs-graph pair-HMM DP of
// ProGraphMSA (reference src/GraphAlign.h:145-163 precomputeScores, :200-534 alignGraphs,
// :165-198 markAlternativePath; src/ls_log.h:22-59).
//
// Arithmetic contract (parity with the reference CPU path, checked against oracle/pgm_oracle.c):
//   * every float operation is a single IEEE binary32 op in the reference's order: no FMA contraction
//     (explicit __fmul_rn/__fadd_rn/__fsub_rn), correctly rounded division (__fdiv_rn), denormals on;
//   * emission dot products accumulate k = 0..D-1 in order from zero, multiply then add;
//   * ls_log is the bit-manipulating SSE2 form (op order t=b/(d-c); u=e+a; t=t+add; t=t+u).
#ifndef PGM_ALIGN_KERNELS_H_
#define PGM_ALIGN_KERNELS_H_

#include <hip/hip_runtime.h>
#include <math.h>
#include "pgm_device.h"

#define PGM_NEG_INF (-INFINITY)

// ---------------------------------------------------------------------------------------------
// ls_log_add, SSE2 float variant (ls_log.h:22-59), one element.
__device__ __forceinline__ float pgm_ls_log_add(float x, float add) {
    const float a = 2.44247459618085927548717403238913328776812604856113966238617812902399112761292613763080658235564f;
    const float b = -4.2040783745848554315883301529007786406310628696382695994938550046831869207082846248658671f;
    const float c = -0.72123729809042963774358701619456664388406302428056983119308906451199556380646306f;
    const uint32_t bits = __float_as_uint(x);
    const float e = (float)((int32_t)(bits >> 23) - 126);
    const float d = __uint_as_float(((bits << 9) >> 9) ^ 0x3f000000u);
    float t = __fdiv_rn(b, __fsub_rn(d, c));
    const float u = __fadd_rn(e, a);
    t = __fadd_rn(t, add);
    t = __fadd_rn(t, u);
    return t;
}

// S(y,x) from the dot product, the two per-node denominators and match_init (GraphAlign.h:154-159).
// x86 SSE yields the default NaN 0xFFC00000 for 0/0 (border rows/columns); canonicalise so that the
// bit-twiddling log sees the same input bits.
__device__ __forceinline__ float pgm_emission_finish(float num, float ay, float bx, float match_init) {
    float q = __fdiv_rn(num, __fmul_rn(ay, bx));
    if (q != q) q = __uint_as_float(0xFFC00000u);
    return pgm_ls_log_add(q, match_init);
}

// ---------------------------------------------------------------------------------------------
// Prep kernel: float casts, T = M^T g2, per-node denominators.  grid = (njobs, 2), block = 256.
//   side 0: g1f[y][k] = float(sites1(k,y));  a1[y] = sum_k g1f[y][k] * pi_f[k]
//   side 1: t2[x][k]  = sum_j M_f(j,k) * g2f(j,x);  aux2[x] = {sum_k pi_f[k] g2f(k,x), cc2[x], xp2[x], count|kill}
__global__ void __launch_bounds__(256) pgm_prep_kernel(const PgmJob *__restrict__ jobs) {
    extern __shared__ float prep_lds[];  // Mf (dim*dim) then pif (dim)
    const PgmJob &J = jobs[blockIdx.x];
    const uint32_t D = J.dim, DP = J.dp;
    float *Mf = prep_lds;
    float *pif = prep_lds + D * D;
    for (uint32_t i = threadIdx.x; i < D * D; i += blockDim.x) Mf[i] = (float)J.M[i];
    for (uint32_t i = threadIdx.x; i < D; i += blockDim.x) pif[i] = (float)J.pi[i];
    __syncthreads();
    if (blockIdx.y == 0) {
        for (uint32_t y = threadIdx.x; y < J.n1; y += blockDim.x) {
            const double *col = J.sites1 + (size_t)D * y;
            float *dst = J.g1f + (size_t)DP * y;
            float acc = 0.0f;
            for (uint32_t k = 0; k < D; ++k) {
                const float g = (float)col[k];
                dst[k] = g;
                acc = __fadd_rn(acc, __fmul_rn(g, pif[k]));
            }
            for (uint32_t k = D; k < DP; ++k) dst[k] = 0.0f;
            J.a1[y] = acc;
        }
    } else {
        for (uint32_t x = threadIdx.x; x < J.n2; x += blockDim.x) {
            const double *col = J.sites2 + (size_t)D * x;
            float *dst = J.t2 + (size_t)DP * x;
            for (uint32_t k = 0; k < D; ++k) {
                float acc = 0.0f;
                for (uint32_t j = 0; j < D; ++j) acc = __fadd_rn(acc, __fmul_rn(Mf[j + D * k], (float)col[j]));
                dst[k] = acc;
            }
            for (uint32_t k = D; k < DP; ++k) dst[k] = 0.0f;
            float b = 0.0f;
            for (uint32_t k = 0; k < D; ++k) b = __fadd_rn(b, __fmul_rn(pif[k], (float)col[k]));
            const uint32_t xb = (uint32_t)J.xp2[x], xe = (uint32_t)J.xp2[x + 1];
            float4 a;
            a.x = b;
            a.y = J.cc2[x];
            a.z = __uint_as_float(xb);
            a.w = __uint_as_float((xe - xb) | ((uint32_t)J.kill2[x] << 31));
            J.aux2[x] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cell addressing (see PgmJob)
__device__ __forceinline__ size_t pgm_cell_index(const PgmJob &J, uint32_t y, uint32_t x) {
    const uint32_t b = y >> 6, l = y & 63u;
    return ((size_t)b * J.tsteps + (x + l)) * 64u + l;
}
__device__ __forceinline__ float4 pgm_load_cell(const PgmJob &J, uint32_t y, uint32_t x) {
    return J.cells[pgm_cell_index(J, y, x)];
}

__device__ __forceinline__ float pgm_dpp_wave_shr1(float src, float lane0_value) {
    // lane l receives src of lane l-1; lane 0 keeps `lane0_value` (DPP wave_shr:1, bound_ctrl off)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0_value), __float_as_int(src), 0x138, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------------------------
// Fill kernel (GraphAlign.h:212-260 incl. the border initialisation as row/column 0).
// One workgroup per job, NW wavefronts; wavefront w sweeps the row bands w, w+NW, ...  Lane l of a
// band owns row y = 64 b + l and at step t computes column x = t - l, so the three chain neighbours
// (y-1,x-1), (y-1,x), (y,x-1) are in registers of lane l-1 (one DPP shift) or of the lane itself.
// Per-wave LDS ring: the last R columns of T = M^T g2 plus the per-column scalars, refilled 16 columns
// at a time.  The last row of a band is handed to the next band through `brow` in global memory,
// guarded by a per-band progress counter in LDS (workgroup-scope release/acquire).
// Predecessors other than the chain edge ("extras": skip edges of merged graphs, repeat edges) are
// read back from the cell storage in HBM/L2.
template <int DP, int NW, int R>
__global__ void __launch_bounds__(NW * 64) pgm_fill_kernel(const PgmJob *__restrict__ jobs, const uint32_t *__restrict__ order) {
    constexpr int NQ = DP / 4 + 1;  // float4 per ring column: DP/4 of T + 1 aux
    extern __shared__ __attribute__((aligned(16))) float4 fill_lds[];
    const PgmJob &J = jobs[order[blockIdx.x]];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *ring = fill_lds + (size_t)wave * R * NQ;
    int *prog = (int *)(fill_lds + (size_t)NW * R * NQ);

    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nb = J.nb;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap, mi = J.sc.match_init;

    for (uint32_t i = threadIdx.x; i < nb; i += NW * 64) prog[i] = 0;
    __syncthreads();

    for (uint32_t b = wave; b < nb; b += NW) {
        const uint32_t y = 64u * b + lane;
        const bool rowvalid = y + 1 < n1;  // rows 0..n1-2
        const uint32_t yc = rowvalid ? y : 0u;
        float gy[DP];
        {
            const float4 *src = (const float4 *)(J.g1f + (size_t)DP * yc);
#pragma unroll
            for (int q = 0; q < DP / 4; ++q) {
                const float4 v = src[q];
                gy[4 * q] = v.x; gy[4 * q + 1] = v.y; gy[4 * q + 2] = v.z; gy[4 * q + 3] = v.w;
            }
        }
        const float ay = J.a1[yc];
        const float ccy = J.cc1[yc];
        const uint32_t xby = (uint32_t)J.xp1[yc], xey = (uint32_t)J.xp1[yc + 1];
        const bool hasy = xey > xby;
        const bool ykill = J.kill1[yc] != 0;
        const float gopen_x = (y == 0) ? sg : gi;  // row 0 opens gaps with start_gap (GraphAlign.h:229)
        const bool has_next = (b + 1 < nb);
        const float2 *brow_prev = J.brow + (size_t)(b - 1) * ncol;  // only dereferenced for b > 0
        float2 *brow_mine = J.brow + (size_t)b * ncol;
        float4 *cells_band = J.cells + (size_t)b * tsteps * 64u;

        float W_left = PGM_NEG_INF, X_left = PGM_NEG_INF, W_diag = PGM_NEG_INF;
        float W_o = PGM_NEG_INF, Y_o = PGM_NEG_INF;

        for (uint32_t t0 = 0; t0 < tsteps; t0 += PGM_BLOCK) {
            // (a) ring refill: columns [t0, t0+16) (T and aux exist for all n2 = ncol+1 nodes)
            for (int idx = lane; idx < PGM_BLOCK * NQ; idx += 64) {
                const uint32_t col = t0 + (uint32_t)(idx / NQ);
                const int q = idx % NQ;
                if (col <= ncol) {
                    float4 v;
                    if (q < DP / 4) v = ((const float4 *)(J.t2 + (size_t)DP * col))[q];
                    else v = J.aux2[col];
                    ring[(size_t)(col & (R - 1)) * NQ + q] = v;
                }
            }
            // (b) boundary row from the previous band
            float bndW = PGM_NEG_INF, bndY = PGM_NEG_INF;
            if (b > 0) {
                const int need = (int)min(t0 + (uint32_t)PGM_BLOCK, ncol);
                if ((int)t0 < need) {
                    while (__hip_atomic_load(&prog[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                        __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane < PGM_BLOCK && t0 + lane < ncol) {
                        const float2 v = brow_prev[t0 + lane];
                        bndW = v.x;
                        bndY = v.y;
                    }
                }
            }
            // (c) 16 anti-diagonal steps
#pragma unroll 1
            for (int i = 0; i < PGM_BLOCK; ++i) {
                const uint32_t t = t0 + i;
                const int xs = (int)t - lane;
                const bool active = rowvalid && xs >= 0 && xs < (int)ncol;
                const uint32_t x = (uint32_t)xs;
                const float4 *rc = ring + (size_t)(x & (R - 1)) * NQ;
                const float4 aux = rc[DP / 4];
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < DP / 4; ++q) {
                    const float4 tv = rc[q];
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q], tv.x));
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q + 1], tv.y));
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q + 2], tv.z));
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q + 3], tv.w));
                }
                const float S = pgm_emission_finish(acc, ay, aux.x, mi);
                const float ccx = aux.y;
                const uint32_t xbx = __float_as_uint(aux.z);
                const uint32_t xw = __float_as_uint(aux.w);
                const uint32_t xnx = xw & 0x7fffffffu;
                const bool xkill = (xw >> 31) != 0;
                const float gopen_y = (xs == 0) ? sg : gi;  // column 0 opens gaps with start_gap (:218)

                const float bw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bndW), i));
                const float by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bndY), i));
                const float W_up = pgm_dpp_wave_shr1(W_o, bw);
                const float Y_up = pgm_dpp_wave_shr1(Y_o, by);

                // chain x chain pair (GraphAlign.h:245-250)
                float Mv = __fsub_rn(__fsub_rn(__fadd_rn(W_diag, S), ccy), ccx);
                float Xv = __fsub_rn(fmaxf(__fadd_rn(X_left, ge), __fadd_rn(W_left, gopen_x)), ccx);
                float Yv = __fsub_rn(fmaxf(__fadd_rn(Y_up, ge), __fadd_rn(W_up, gopen_y)), ccy);

                const bool slow = active && (hasy || xnx != 0);
                if (__builtin_amdgcn_ballot_w64(slow) != 0) {
                    if (slow) {
                        for (uint32_t e = xby; e < xey; ++e) {
                            const uint32_t yp = J.xc1[e];
                            const float cy = J.xv1[e];
                            const float4 c = pgm_load_cell(J, yp, x);
                            Yv = fmaxf(Yv, __fsub_rn(fmaxf(__fadd_rn(c.z, ge), __fadd_rn(c.w, gopen_y)), cy));
                            if (x > 0) {
                                const float4 c2 = pgm_load_cell(J, yp, x - 1);
                                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c2.w, S), cy), ccx));
                            }
                            for (uint32_t f = xbx; f < xbx + xnx; ++f) {
                                const uint32_t xp = J.xc2[f];
                                const float cx = J.xv2[f];
                                const float4 c3 = pgm_load_cell(J, yp, xp);
                                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c3.w, S), cy), cx));
                            }
                        }
                        for (uint32_t f = xbx; f < xbx + xnx; ++f) {
                            const uint32_t xp = J.xc2[f];
                            const float cx = J.xv2[f];
                            const float4 c = pgm_load_cell(J, y, xp);
                            Xv = fmaxf(Xv, __fsub_rn(fmaxf(__fadd_rn(c.y, ge), __fadd_rn(c.w, gopen_x)), cx));
                            if (y > 0) {
                                const float4 c2 = pgm_load_cell(J, y - 1, xp);
                                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c2.w, S), ccy), cx));
                            }
                        }
                    }
                }
                if (ykill) Xv = PGM_NEG_INF;  // interior row without predecessors: the pair loop never runs
                if (xkill) Yv = PGM_NEG_INF;
                float Wv = fmaxf(Mv, fmaxf(Xv, Yv));
                if (y == 0 && xs == 0) Wv = J.sc.start_init;  // GraphAlign.h:212
                if (active) {
                    float4 cell;
                    cell.x = Mv; cell.y = Xv; cell.z = Yv; cell.w = Wv;
                    cells_band[(size_t)t * 64u + lane] = cell;
                    if (lane == 63 && has_next) brow_mine[x] = make_float2(Wv, Yv);
                    W_left = Wv;
                    X_left = Xv;
                }
                W_diag = W_up;
                W_o = Wv;
                Y_o = Yv;
            }
            // (d) publish progress of this band's last row
            if (has_next) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                int done = (int)t0 + PGM_BLOCK - 63;
                done = done < 0 ? 0 : (done > (int)ncol ? (int)ncol : done);
                if (lane == 0) __hip_atomic_store(&prog[b], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense emission matrix (test hook only): S[y + n1*x] for all nodes, same arithmetic as the fill.
__global__ void __launch_bounds__(256) pgm_emission_kernel(const PgmJob *__restrict__ jobs, uint32_t job, float *__restrict__ S) {
    const PgmJob &J = jobs[job];
    const size_t N = (size_t)J.n1 * J.n2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(i % J.n1), x = (uint32_t)(i / J.n1);
        const float *g = J.g1f + (size_t)J.dp * y;
        const float *tt = J.t2 + (size_t)J.dp * x;
        float acc = 0.0f;
        for (uint32_t k = 0; k < J.dim; ++k) acc = __fadd_rn(acc, __fmul_rn(g[k], tt[k]));
        S[i] = pgm_emission_finish(acc, J.a1[y], J.aux2[x].x, J.sc.match_init);
    }
}

// ---------------------------------------------------------------------------------------------
// Traceback (GraphAlign.h:264-521) — one wavefront per job, lane 0 walks END -> START with the
// reference's tie rules (smallest |current - recomputed|, first candidate in PredIterator order wins,
// extension tested before opening, state after a W source chosen by equality in the order M, Y, X).
struct PgmPred {
    const int32_t *pp; const uint32_t *pc; const float *pv; const uint32_t *pu;
};
__device__ __forceinline__ float pgm_mark_value(const PgmPred &P, int32_t e) {
    // value() with repeat_init = repeat_ext = +inf (markAlternativePath, GraphAlign.h:174)
    const uint32_t u = P.pu[e];
    if (u == 0) return P.pv[e];
    const uint32_t units = u & 0x7fffffffu;
    if (units == 0) return INFINITY;
    return __fadd_rn(INFINITY, __fmul_rn(INFINITY, (float)(units - 1)));
}

struct PgmMapOut {
    uint32_t *m1, *m2;
    uint32_t len, cap;
    __device__ __forceinline__ void push(uint32_t y, uint32_t x) {
        if (len < cap) { m1[len] = y; m2[len] = x; }
        ++len;
    }
};

__device__ static void pgm_mark_alternative_path(const PgmJob &J, uint32_t start, uint32_t end, const PgmPred &P, PgmMapOut &mo, bool first) {
    const uint32_t len = end - start + 1;
    float *score = J.mark_score;
    uint32_t *prev = J.mark_prev;
    for (uint32_t i = 0; i < len; ++i) { score[i] = PGM_NEG_INF; prev[i] = 0xFFFFFFFFu; }
    score[0] = 0.0f;
    for (uint32_t i = 1; i < len; ++i) {
        const uint32_t real_ix = i + start;
        for (int32_t e = P.pp[real_ix]; e < P.pp[real_ix + 1]; ++e) {
            const uint32_t p = P.pc[e];
            if (p >= start && p <= end) {
                const uint32_t i2 = p - start;
                const float cand = __fsub_rn(score[i2], pgm_mark_value(P, e));
                if (score[i] <= cand) { score[i] = cand; prev[i] = i2; }
            }
        }
    }
    if (score[len - 1] > PGM_NEG_INF) {
        uint32_t i = prev[len - 1];
        while (i != 0) {
            if (first) mo.push(i + start, 0xFFFFFFFFu); else mo.push(0xFFFFFFFFu, i + start);
            i = prev[i];
        }
    }
}

__device__ __forceinline__ float pgm_emission_at(const PgmJob &J, uint32_t y, uint32_t x) {
    const float *g = J.g1f + (size_t)J.dp * y;
    const float *tt = J.t2 + (size_t)J.dp * x;
    float acc = 0.0f;
    for (uint32_t k = 0; k < J.dim; ++k) acc = __fadd_rn(acc, __fmul_rn(g[k], tt[k]));
    return pgm_emission_finish(acc, J.a1[y], J.aux2[x].x, J.sc.match_init);
}

__global__ void __launch_bounds__(64) pgm_traceback_kernel(const PgmJob *__restrict__ jobs) {
    const PgmJob &J = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    __shared__ uint32_t s_len;
    if (lane == 0) {
        const pgm_scores s = J.sc;
        const uint32_t n1 = J.n1, n2 = J.n2;
        const PgmPred P1 = {J.pp1, J.pc1, J.pv1, J.pu1};
        const PgmPred P2 = {J.pp2, J.pc2, J.pv2, J.pu2};
        int status = PGM_OK;
        uint32_t n_tr = 0;

        // end node (GraphAlign.h:264-280)
        float Wend = PGM_NEG_INF;
        for (int32_t ey = P1.pp[n1 - 1]; ey < P1.pp[n1]; ++ey) {
            for (int32_t ex = P2.pp[n2 - 1]; ex < P2.pp[n2]; ++ex) {
                const uint32_t yp = P1.pc[ey], xp = P2.pc[ex];
                const float yv = P1.pv[ey], xv = P2.pv[ex];
                if (xp == 0 && yp == 0) {
                    Wend = fmaxf(__fsub_rn(__fsub_rn(s.end_skip, yv), xv), Wend);
                } else {
                    const float4 c = pgm_load_cell(J, yp, xp);
                    Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.y, s.end_gap), yv), xv), Wend);
                    Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.z, s.end_gap), yv), xv), Wend);
                    Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.x, s.end_match), yv), xv), Wend);
                }
            }
        }

        enum { State_m = 0, State_x = 1, State_y = 2 };
        int current_state = State_m, next_state = State_m;
        float current_score = PGM_NEG_INF;
        uint32_t y = n1 - 1, x = n2 - 1;
        PgmMapOut mo = {J.map1, J.map2, 0u, n1 + n2};
        mo.push(n1 - 1, n2 - 1);

        bool tr_x = false, tr_y = false;
        float best = INFINITY;
        for (int32_t ey = P1.pp[n1 - 1]; ey < P1.pp[n1]; ++ey) {
            for (int32_t ex = P2.pp[n2 - 1]; ex < P2.pp[n2]; ++ex) {
                const uint32_t yp = P1.pc[ey], xp = P2.pc[ex];
                const float yv = P1.pv[ey], xv = P2.pv[ex];
                const bool ry = P1.pu[ey] != 0, rx = P2.pu[ex] != 0;
                const float4 c = pgm_load_cell(J, yp, xp);
                float d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.x, s.end_match), yv), xv)));
                if (best > d) { best = d; tr_x = rx; tr_y = ry; current_score = c.x; current_state = State_m; y = yp; x = xp; }
                d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.z, s.end_gap), yv), xv)));
                if (best > d) { best = d; tr_x = rx; tr_y = ry; current_score = c.z; current_state = State_y; y = yp; x = xp; }
                d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.y, s.end_gap), yv), xv)));
                if (best > d) { best = d; tr_x = rx; tr_y = ry; current_score = c.y; current_state = State_x; y = yp; x = xp; }
                d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(s.end_skip, yv), xv)));
                if (xp == 0 && yp == 0 && best > d) { best = d; tr_x = rx; tr_y = ry; y = yp; x = xp; }
            }
        }
        n_tr += (uint32_t)tr_x + (uint32_t)tr_y;
        if (tr_y) pgm_mark_alternative_path(J, y, n1 - 1, P1, mo, true);
        if (tr_x) pgm_mark_alternative_path(J, x, n2 - 1, P2, mo, false);
        if (x != 0 || y != 0) {
            if (current_state == State_m) mo.push(y, x);
            else if (current_state == State_x) mo.push(0xFFFFFFFFu, x);
            else mo.push(y, 0xFFFFFFFFu);
        }

        float next_score = INFINITY;
        uint32_t next_x = 0xFFFFFFFFu, next_y = 0xFFFFFFFFu;
        uint32_t guard = 0;
        while ((x != 0 || y != 0) && status == PGM_OK) {
            if (++guard > n1 + n2 + 4) { status = PGM_ERR_BACKTRACK; break; }
            best = INFINITY;
            // choose the state of a W source by equality, order M, Y, X (GraphAlign.h:400-411)
#define PGM_PICK(cell)                                                                        \
    if (next_x != 0 || next_y != 0) {                                                          \
        if ((cell).w == (cell).x) { next_score = (cell).x; next_state = State_m; }              \
        else if ((cell).w == (cell).z) { next_score = (cell).z; next_state = State_y; }         \
        else if ((cell).w == (cell).y) { next_score = (cell).y; next_state = State_x; }         \
        else status = PGM_ERR_BACKTRACK;                                                       \
    }
            if (current_state == State_y) {
                for (int32_t e = P1.pp[y]; e < P1.pp[y + 1]; ++e) {
                    const uint32_t yp = P1.pc[e];
                    const float yv = P1.pv[e];
                    const float4 c = pgm_load_cell(J, yp, x);
                    float d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.z, s.gap_extend), yv)));
                    if (best > d) { best = d; tr_x = false; tr_y = P1.pu[e] != 0; next_x = x; next_y = yp; next_score = c.z; next_state = State_y; }
                    d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.w, s.gap_init), yv)));
                    if (best > d) { best = d; tr_x = false; tr_y = P1.pu[e] != 0; next_x = x; next_y = yp; PGM_PICK(c) }
                }
            } else if (current_state == State_x) {
                for (int32_t e = P2.pp[x]; e < P2.pp[x + 1]; ++e) {
                    const uint32_t xp = P2.pc[e];
                    const float xv = P2.pv[e];
                    const float4 c = pgm_load_cell(J, y, xp);
                    float d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.y, s.gap_extend), xv)));
                    if (best > d) { best = d; tr_x = P2.pu[e] != 0; tr_y = false; next_x = xp; next_y = y; next_score = c.y; next_state = State_x; }
                    d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.w, s.gap_init), xv)));
                    if (best > d) { best = d; tr_x = P2.pu[e] != 0; tr_y = false; next_x = xp; next_y = y; PGM_PICK(c) }
                }
            } else {
                const float S = pgm_emission_at(J, y, x);
                for (int32_t ey = P1.pp[y]; ey < P1.pp[y + 1]; ++ey) {
                    for (int32_t ex = P2.pp[x]; ex < P2.pp[x + 1]; ++ex) {
                        const uint32_t yp = P1.pc[ey], xp = P2.pc[ex];
                        const float yv = P1.pv[ey], xv = P2.pv[ex];
                        const float4 c = pgm_load_cell(J, yp, xp);
                        const float d = fabsf(__fsub_rn(current_score, __fsub_rn(__fsub_rn(__fadd_rn(c.w, S), yv), xv)));
                        if (best > d) { best = d; tr_x = P2.pu[ex] != 0; tr_y = P1.pu[ey] != 0; next_y = yp; next_x = xp; PGM_PICK(c) }
                    }
                }
            }
#undef PGM_PICK
            if (status != PGM_OK) break;
            n_tr += (uint32_t)tr_x + (uint32_t)tr_y;
            if (tr_y) pgm_mark_alternative_path(J, next_y, y, P1, mo, true);
            if (tr_x) pgm_mark_alternative_path(J, next_x, x, P2, mo, false);
            x = next_x; y = next_y;
            current_state = next_state;
            current_score = next_score;
            if (x != 0 || y != 0) {
                if (current_state == State_m) mo.push(y, x);
                else if (current_state == State_x) mo.push(0xFFFFFFFFu, x);
                else mo.push(y, 0xFFFFFFFFu);
            }
        }
        mo.push(0, 0);
        if (mo.len > mo.cap) { status = PGM_ERR_BACKTRACK; mo.len = mo.cap; }
        J.result->score = Wend;
        J.result->n_tr_indels = n_tr;
        J.result->len = mo.len;
        J.result->status = status;
        s_len = mo.len;
    }
    __syncthreads();
    // reverse the two mappings in place (GraphAlign.h:520-521), all 64 lanes
    const uint32_t len = s_len;
    __threadfence_block();
    for (uint32_t i = lane; i < len / 2; i += 64) {
        const uint32_t j = len - 1 - i;
        uint32_t a = J.map1[i], b2 = J.map1[j];
        J.map1[i] = b2; J.map1[j] = a;
        a = J.map2[i]; b2 = J.map2[j];
        J.map2[i] = b2; J.map2[j] = a;
    }
}

#endif
