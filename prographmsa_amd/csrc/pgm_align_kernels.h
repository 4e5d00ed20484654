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
//   side 1: t2[x][k]  = sum_j M_f(j,k) * g2f(j,x);  aux2[x] = {sum_k pi_f[k] g2f(k,x), PgmNodeInfo of column x}
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
            const PgmNodeInfo ni = J.ni2[x];
            float4 a;
            a.x = b;
            a.y = ni.cc;
            a.z = __uint_as_float(ni.flags);
            a.w = __uint_as_float(ni.dpack);
            J.aux2[2 * x] = a;
            J.aux2[2 * x + 1] = make_float4(ni.c1, ni.c2, ni.c3, 0.0f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cell addressing (see PgmJob): cell = float4 {M, X, W, Y}
__device__ __forceinline__ size_t pgm_cell_index(const PgmJob &J, uint32_t y, uint32_t x) {
    const uint32_t b = y / PGM_ROWS, l = PGM_HALO + (y - b * PGM_ROWS);
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
//
// One workgroup per job, NW wavefronts; wavefront w sweeps the row bands w, w+NW, ...  A band is 48
// rows: lanes 16..63 own rows y = 48 b + l - 16 and at step t compute column x = t - l, so the three
// chain neighbours (y-1,x-1), (y-1,x), (y,x-1) are in registers of lane l-1 (one DPP shift) or of the
// lane itself.  Lanes 0..15 do not compute: they REPLAY the last 16 rows of band b-1 (read back from
// the cell storage, one contiguous 256 B run per step, staged through LDS a block ahead), so that
// every predecessor row within 16 rows of a lane lives in a lower lane of the same wavefront.
//
// Per-wavefront LDS (no LDS is shared between wavefronts except the progress counters):
//   ring  : the last 80 columns of T = M^T g2 plus the per-column predecessor summary (refilled 16
//           columns at a time, loaded one block ahead)
//   hW/hY/hX : W, Y, X of all 64 lanes for the last 16 steps.  A skip-edge predecessor pair
//           (y-dy, x-dx) was produced by lane l-dy exactly dy+dx steps ago, so merged-graph skip edges
//           (distances of a few nodes) are served from this history without touching HBM.
//   rep   : replay values {W,Y} of the next 16 steps
// Predecessor pairs that reach further than the history (dy+dx > 15), nodes with more than 3 extra
// predecessors and tandem-repeat edges far back fall back to reading the cell storage (HBM/L2).
// Bands hand over through the cell storage itself; prog[b] (LDS) = number of steps of band b that
// are complete AND visible (the producer publishes behind a counted s_waitcnt, never vmcnt(0)).
template <int DP, int NW>
__global__ void __launch_bounds__(NW * 64) pgm_fill_kernel(const PgmJob *__restrict__ jobs, const uint32_t *__restrict__ order) {
    constexpr int NT = DP / 4;       // float4 of T per column
    constexpr int NQ = NT + 2;       // + {b, cc, flags, dpack} + {c1, c2, c3, -}
    constexpr int R = PGM_RING, H = PGM_HIST, HR = PGM_HALO, RC = PGM_ROWS, BL = PGM_BLOCK;
    constexpr int WAVE_LDS = R * NQ * 16 + 3 * H * 64 * 4 + BL * HR * 8;  // bytes
    static_assert(WAVE_LDS % 16 == 0, "LDS carve must stay 16-byte aligned");
    constexpr int PFQ = (BL * NQ + 63) / 64;   // ring prefetch quads per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char fill_lds[];
    const PgmJob &J = jobs[order[blockIdx.x]];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char *wl = fill_lds + (size_t)wave * WAVE_LDS;
    float4 *ring = (float4 *)wl;
    float *hW = (float *)(wl + R * NQ * 16);
    float *hY = hW + H * 64;
    float *hX = hY + H * 64;
    float2 *rep = (float2 *)(hX + H * 64);
    int *prog = (int *)(fill_lds + (size_t)NW * WAVE_LDS);

    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nb = J.nb;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap, mi = J.sc.match_init, s_init = J.sc.start_init;
    const float4 *t2q = (const float4 *)J.t2;

    for (uint32_t i = threadIdx.x; i < nb; i += NW * 64) prog[i] = 0;
    __syncthreads();

    for (uint32_t b = wave; b < nb; b += NW) {
        const bool comp = lane >= HR;                       // compute lane (else replay lane)
        const uint32_t y = RC * b + (uint32_t)(lane - HR);  // only meaningful for compute lanes
        const bool rowvalid = comp && y + 1 < n1;           // rows 0..n1-2
        const uint32_t yc = rowvalid ? y : 0u;
        float gy[DP];
        {
            const float4 *src = (const float4 *)(J.g1f + (size_t)DP * yc);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float4 v = src[q];
                gy[4 * q] = v.x; gy[4 * q + 1] = v.y; gy[4 * q + 2] = v.z; gy[4 * q + 3] = v.w;
            }
        }
        const float ay = J.a1[yc];
        const PgmNodeInfo niy = J.ni1[yc];
        const float ccy = niy.cc;
        const uint32_t fy = rowvalid ? niy.flags : 0u;
        const uint32_t nyl = fy & 3u;
        const bool geny = (fy & 4u) != 0;
        const bool ykill = (fy & 8u) != 0;
        const uint32_t dymax = (fy >> 8) & 255u;
        const uint32_t dpy = niy.dpack;
        const float cy1 = niy.c1, cy2 = niy.c2, cy3 = niy.c3;
        const uint32_t xby = (uint32_t)J.xp1[yc], xey = (uint32_t)J.xp1[yc + 1];
        const float gopen_x = (rowvalid && y == 0) ? sg : gi;  // row 0 opens gaps with start_gap (GraphAlign.h:229)
        const bool has_next = (b + 1 < nb);
        const bool has_prev = (b > 0);
        float4 *cells_band = J.cells + (size_t)b * tsteps * 64u;
        const float4 *cells_prev = J.cells + (size_t)(b - 1) * tsteps * 64u;  // only dereferenced if has_prev

        // history planes start at -inf (no NaN may ever be read from them)
        for (int i = lane; i < 3 * H * 64; i += 64) hW[i] = PGM_NEG_INF;

        float W_left = PGM_NEG_INF, X_left = PGM_NEG_INF, W_diag = PGM_NEG_INF;
        float W_o = PGM_NEG_INF, Y_o = PGM_NEG_INF;
        int xr = (R - lane) % R;   // ring slot of this lane's column, advanced every step

        // ---- prefetch state -------------------------------------------------------------------
        float4 pfq[PFQ];           // ring columns [t0+16, t0+32) loaded during block t0
        float2 pfr[4];             // replay tile of block t0+16 loaded during block t0
        auto load_ring_block = [&](uint32_t c0) {
#pragma unroll
            for (int u = 0; u < PFQ; ++u) {
                const int idx = lane + 64 * u;
                const uint32_t col = c0 + (uint32_t)(idx / NQ);
                const int q = idx % NQ;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < BL * NQ && col <= ncol) {
                    if (q < NT) v = t2q[(size_t)NT * col + q];
                    else v = J.aux2[2 * col + (q - NT)];
                }
                pfq[u] = v;
            }
        };
        auto store_ring_block = [&](uint32_t c0) {
#pragma unroll
            for (int u = 0; u < PFQ; ++u) {
                const int idx = lane + 64 * u;
                const uint32_t col = c0 + (uint32_t)(idx / NQ);
                const int q = idx % NQ;
                if (idx < BL * NQ) ring[(size_t)(col % R) * NQ + q] = pfq[u];
            }
        };
        // replay tile of the 16 steps starting at s0: element e = i*16 + l -> {W,Y} of band b-1, step s0+i+48, lane 48+l
        auto load_rep_block = [&](uint32_t s0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane * 4 + u;
                const uint32_t st = s0 + (uint32_t)(e >> 4) + (uint32_t)RC;
                const int l = e & 15;
                float2 v = make_float2(PGM_NEG_INF, PGM_NEG_INF);
                if (has_prev && st < tsteps) {
                    const float4 c = cells_prev[(size_t)st * 64u + (uint32_t)(RC + l)];
                    v = make_float2(c.z, c.w);
                }
                pfr[u] = v;
            }
        };
        auto store_rep_block = [&]() {
#pragma unroll
            for (int u = 0; u < 4; ++u) rep[lane * 4 + u] = pfr[u];
        };
        auto wait_prev = [&](uint32_t steps_needed) {   // band b-1 has completed (and made visible) that many steps
            if (has_prev) {
                const int need = (int)min(steps_needed, tsteps);
                while (__hip_atomic_load(&prog[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                    __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        };

        // prologue: data of block 0 straight in, data of block 1 in flight
        load_ring_block(0);
        store_ring_block(0);
        wait_prev(BL + RC + BL);
        load_rep_block(0);
        store_rep_block();
        load_ring_block(BL);
        load_rep_block(BL);

        for (uint32_t t0 = 0; t0 < tsteps; t0 += BL) {
            if (t0 > 0) {
                // data loaded during the previous block becomes current; start loading the next block
                store_ring_block(t0);
                store_rep_block();
                load_ring_block(t0 + BL);
                wait_prev(t0 + BL + BL + RC);   // replay tile of block t0+16 reads steps < t0+16+16+48 of band b-1
                load_rep_block(t0 + BL);
            }
#pragma unroll 1
            for (int i = 0; i < BL; ++i) {
                const uint32_t t = t0 + i;
                const int xs = (int)t - lane;
                const bool incol = xs >= 0 && xs < (int)ncol;
                const bool active = rowvalid && incol;
                const uint32_t x = (uint32_t)xs;
                const float4 *rc = ring + (size_t)xr * NQ;
                xr = (xr + 1 == R) ? 0 : xr + 1;
                const float4 aux = rc[NT];
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const float4 tv = rc[q];
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q], tv.x));
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q + 1], tv.y));
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q + 2], tv.z));
                    acc = __fadd_rn(acc, __fmul_rn(gy[4 * q + 3], tv.w));
                }
                const float S = pgm_emission_finish(acc, ay, aux.x, mi);
                const float ccx = aux.y;
                const uint32_t fx = active ? __float_as_uint(aux.z) : 0u;
                const uint32_t dpx = __float_as_uint(aux.w);
                const uint32_t nxl = fx & 3u;
                const bool genx = (fx & 4u) != 0;
                const bool xkill = (fx & 8u) != 0;
                const uint32_t dxmax = (fx >> 8) & 255u;
                const float gopen_y = (xs == 0) ? sg : gi;  // column 0 opens gaps with start_gap (:218)

                const float W_up = pgm_dpp_wave_shr1(W_o, PGM_NEG_INF);
                const float Y_up = pgm_dpp_wave_shr1(Y_o, PGM_NEG_INF);

                // chain x chain pair (GraphAlign.h:245-250)
                float Mv = __fsub_rn(__fsub_rn(__fadd_rn(W_diag, S), ccy), ccx);
                float Xv = __fsub_rn(fmaxf(__fadd_rn(X_left, ge), __fadd_rn(W_left, gopen_x)), ccx);
                float Yv = __fsub_rn(fmaxf(__fadd_rn(Y_up, ge), __fadd_rn(W_up, gopen_y)), ccy);

                const bool anyex = active && ((nyl | nxl) != 0 || geny || genx);
                if (__builtin_amdgcn_ballot_w64(anyex) != 0) {
                    const uint32_t reach = max(dymax, 1u) + max(dxmax, 1u);
                    const bool gen = anyex && (geny || genx || reach > (uint32_t)(H - 1));
                    const bool near = anyex && !gen;
                    if (__builtin_amdgcn_ballot_w64(near) != 0) {
                        const float4 cxs = rc[NT + 1];
                        // row extras k = 0..2 (distance dyk, cost cyk)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const bool mk = near && (uint32_t)k < nyl;
                            if (__builtin_amdgcn_ballot_w64(mk) != 0) {
                                const uint32_t dyk = (dpy >> (8 * k)) & 255u;
                                const float cyk = k == 0 ? cy1 : (k == 1 ? cy2 : cy3);
                                if (mk) {
                                    const int sl = lane - (int)dyk;
                                    const int o0 = (int)(((t - dyk) & (H - 1)) << 6) + sl;
                                    const float Wk = hW[o0], Yk = hY[o0];
                                    Yv = fmaxf(Yv, __fsub_rn(fmaxf(__fadd_rn(Yk, ge), __fadd_rn(Wk, gopen_y)), cyk));
                                    const float W1 = hW[(int)(((t - dyk - 1u) & (H - 1)) << 6) + sl];
                                    Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(W1, S), cyk), ccx));
#pragma unroll
                                    for (int j = 0; j < 3; ++j) {
                                        if ((uint32_t)j < nxl) {
                                            const uint32_t dxj = (dpx >> (8 * j)) & 255u;
                                            const float cxj = j == 0 ? cxs.x : (j == 1 ? cxs.y : cxs.z);
                                            const float W2 = hW[(int)(((t - dyk - dxj) & (H - 1)) << 6) + sl];
                                            Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(W2, S), cyk), cxj));
                                        }
                                    }
                                }
                            }
                        }
                        // column extras j = 0..2 (distance dxj, cost cxj)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const bool mj = near && (uint32_t)j < nxl;
                            if (__builtin_amdgcn_ballot_w64(mj) != 0) {
                                const uint32_t dxj = (dpx >> (8 * j)) & 255u;
                                const float cxj = j == 0 ? cxs.x : (j == 1 ? cxs.y : cxs.z);
                                if (mj) {
                                    const int o0 = (int)(((t - dxj) & (H - 1)) << 6) + lane;
                                    const float Wj = hW[o0], Xj = hX[o0];
                                    Xv = fmaxf(Xv, __fsub_rn(fmaxf(__fadd_rn(Xj, ge), __fadd_rn(Wj, gopen_x)), cxj));
                                    const float W1 = hW[(int)(((t - 1u - dxj) & (H - 1)) << 6) + lane - 1];
                                    Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(W1, S), ccy), cxj));
                                }
                            }
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(gen) != 0) {
                        // far / many predecessors: every non-chain pair from the cell storage (HBM/L2)
                        if (gen) {
                            const uint32_t xbx = (uint32_t)J.xp2[x], xex = (uint32_t)J.xp2[x + 1];
                            for (uint32_t e = xby; e < xey; ++e) {
                                const uint32_t yp = J.xc1[e];
                                const float cy = J.xv1[e];
                                const float4 c = pgm_load_cell(J, yp, x);
                                Yv = fmaxf(Yv, __fsub_rn(fmaxf(__fadd_rn(c.w, ge), __fadd_rn(c.z, gopen_y)), cy));
                                if (x > 0) {
                                    const float4 c2 = pgm_load_cell(J, yp, x - 1);
                                    Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c2.z, S), cy), ccx));
                                }
                                for (uint32_t f = xbx; f < xex; ++f) {
                                    const uint32_t xp = J.xc2[f];
                                    const float cx = J.xv2[f];
                                    const float4 c3 = pgm_load_cell(J, yp, xp);
                                    Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c3.z, S), cy), cx));
                                }
                            }
                            for (uint32_t f = xbx; f < xex; ++f) {
                                const uint32_t xp = J.xc2[f];
                                const float cx = J.xv2[f];
                                const float4 c = pgm_load_cell(J, y, xp);
                                Xv = fmaxf(Xv, __fsub_rn(fmaxf(__fadd_rn(c.y, ge), __fadd_rn(c.z, gopen_x)), cx));
                                if (y > 0) {
                                    const float4 c2 = pgm_load_cell(J, y - 1, xp);
                                    Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c2.z, S), ccy), cx));
                                }
                            }
                        }
                    }
                }
                if (ykill) Xv = PGM_NEG_INF;  // interior row without predecessors: the pair loop never runs
                if (xkill) Yv = PGM_NEG_INF;
                float Wv = fmaxf(Mv, fmaxf(Xv, Yv));
                if (rowvalid && y == 0 && xs == 0) Wv = s_init;  // GraphAlign.h:212
                if (!active) { Mv = PGM_NEG_INF; Xv = PGM_NEG_INF; Yv = PGM_NEG_INF; Wv = PGM_NEG_INF; }
                if (!comp) {
                    // replay lane: the values band b-1 computed for this cell
                    const float2 rv = rep[i * HR + (lane & (HR - 1))];
                    Wv = incol ? rv.x : PGM_NEG_INF;
                    Yv = incol ? rv.y : PGM_NEG_INF;
                }
                if (active) {
                    float4 cell;
                    cell.x = Mv; cell.y = Xv; cell.z = Wv; cell.w = Yv;
                    cells_band[(size_t)t * 64u + lane] = cell;
                    W_left = Wv;
                    X_left = Xv;
                }
                {
                    const int ho = (int)((t & (H - 1)) << 6) + lane;
                    hW[ho] = Wv; hY[ho] = Yv; hX[ho] = Xv;
                }
                W_diag = W_up;
                W_o = Wv;
                Y_o = Yv;
            }
            // publish: everything but the youngest few vector-memory operations of this wavefront has
            // completed, hence every cell store of the blocks before this one is visible to the CU.
            if (has_next) {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                if (lane == 0) __hip_atomic_store(&prog[b], (int)t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (has_next) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&prog[b], (int)0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        S[i] = pgm_emission_finish(acc, J.a1[y], J.aux2[2 * x].x, J.sc.match_init);
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
    return pgm_emission_finish(acc, J.a1[y], J.aux2[2 * x].x, J.sc.match_init);
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
                    Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.w, s.end_gap), yv), xv), Wend);
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
                d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.w, s.end_gap), yv), xv)));
                if (best > d) { best = d; tr_x = rx; tr_y = ry; current_score = c.w; current_state = State_y; y = yp; x = xp; }
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
        if ((cell).z == (cell).x) { next_score = (cell).x; next_state = State_m; }              \
        else if ((cell).z == (cell).w) { next_score = (cell).w; next_state = State_y; }         \
        else if ((cell).z == (cell).y) { next_score = (cell).y; next_state = State_x; }         \
        else status = PGM_ERR_BACKTRACK;                                                       \
    }
            if (current_state == State_y) {
                for (int32_t e = P1.pp[y]; e < P1.pp[y + 1]; ++e) {
                    const uint32_t yp = P1.pc[e];
                    const float yv = P1.pv[e];
                    const float4 c = pgm_load_cell(J, yp, x);
                    float d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.w, s.gap_extend), yv)));
                    if (best > d) { best = d; tr_x = false; tr_y = P1.pu[e] != 0; next_x = x; next_y = yp; next_score = c.w; next_state = State_y; }
                    d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.z, s.gap_init), yv)));
                    if (best > d) { best = d; tr_x = false; tr_y = P1.pu[e] != 0; next_x = x; next_y = yp; PGM_PICK(c) }
                }
            } else if (current_state == State_x) {
                for (int32_t e = P2.pp[x]; e < P2.pp[x + 1]; ++e) {
                    const uint32_t xp = P2.pc[e];
                    const float xv = P2.pv[e];
                    const float4 c = pgm_load_cell(J, y, xp);
                    float d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.y, s.gap_extend), xv)));
                    if (best > d) { best = d; tr_x = P2.pu[e] != 0; tr_y = false; next_x = xp; next_y = y; next_score = c.y; next_state = State_x; }
                    d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(c.z, s.gap_init), xv)));
                    if (best > d) { best = d; tr_x = P2.pu[e] != 0; tr_y = false; next_x = xp; next_y = y; PGM_PICK(c) }
                }
            } else {
                const float S = pgm_emission_at(J, y, x);
                for (int32_t ey = P1.pp[y]; ey < P1.pp[y + 1]; ++ey) {
                    for (int32_t ex = P2.pp[x]; ex < P2.pp[x + 1]; ++ex) {
                        const uint32_t yp = P1.pc[ey], xp = P2.pc[ex];
                        const float yv = P1.pv[ey], xv = P2.pv[ex];
                        const float4 c = pgm_load_cell(J, yp, xp);
                        const float d = fabsf(__fsub_rn(current_score, __fsub_rn(__fsub_rn(__fadd_rn(c.z, S), yv), xv)));
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
