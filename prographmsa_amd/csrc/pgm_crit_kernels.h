// pgm_crit_kernels.h — the sweep of a band on the batch's critical path, split over the sixteen wavefronts of a 1024-thread
// worker (reference src/GraphAlign.h:237-260, the fill loop; the same cells, the same float operations as pgm_sweep_band<2>).
//
// Why.  The chain of sweeps of the largest job bounds a batch: nb * lag + n2 steps of ONE wavefront.  In
// pgm_sweep_band<2> that wavefront issued ~115 instructions per step, although the recurrence only forces five of them to wait
// for the step before: of all terms of a cell (y, x) at step t = x + lane only TWO read step t - 1 —
//     X from column x-1 (the lane's own previous cell) and Y from row y-1 (the lane above, one DPP shift) —
// every M term reads step t - 2 or older ((y-1, x-1) is two anti-diagonals back), and so does every other X / Y term.
//
// Roles of a worker (one band of one job at a time; wavefront w of the workgroup):
//   0      CHAIN   pgm_crit_sweep: the two chain terms, the merge with the step's pre-folded maxima {M, X, Y}, W, the cell store,
//                  the record of {W, Y, X} in the LDS history, the hand-off with the bands above and below (as pgm_sweep_band)
//   1, 2   FOLD    pgm_crit_fold, the even / the odd steps: once step t - 2 is recorded, the three terms that read it (M from
//                  (y-1, x-1), X from column x-2, Y from row y-2), merged with the maxima the helpers below have folded for step t
//                  (read and reset here); the result is the step's pre-folded {M, X, Y} (plain LDS stores, counter last)
//   3, 4   NEAR    pgm_crit_near, even / odd steps: the ten near terms that read step t - 3 and older (eight M pairs, X from column
//                  x-3, Y from row y-3)
//   5 - 8  COLUMNS pgm_terms_helper<1>: far edges of the columns, parts 0 / 1 x even / odd steps
//   9 - 14 ROWS    pgm_terms_helper<2>: far edges of the rows, parts 0 / 1 / 2 x even / odd steps
// All of 3-14 fold into res[t & 3] with LDS float-max atomics as before (exact, order free); the FOLD wavefront of a step reads
// and resets those words, so the CHAIN wavefront reads three words per step and resets nothing.
//
// Jobs: PgmJob::crit3 — MODE 2 jobs without long / remote entries, overflow columns or generic nodes (everything else stays with
// pgm_fill_kernel).  Hand-shakes (LDS words, fsync[]): [0] last step the chain has recorded + 2 (1: prologue done), [1] / [2] steps
// published by the FOLD wavefront of the even / odd steps, [3] row entry list built, [4..11] / [12..19] steps published by the
// helper wavefronts of the even / odd steps ({NEAR, COLUMNS 0, 1, ROWS 0, 1, 2, -, -}; wavefronts that take no part: "far ahead").
#ifndef PGM_CRIT_KERNELS_H_
#define PGM_CRIT_KERNELS_H_

#include "pgm_align_kernels.h"

#define PGM_C3_WAVES 16
#define PGM_C3_PRE 0         /* float pre[4][3][64]: pre-folded maxima {M, X, Y} of the steps t & 3 */
#define PGM_C3_COLA 3072     /* float2 colA[PGM_NRING]: what the chain wavefront needs of a column {cost of the chain edge, gap opening score of Y (start_gap in column 0)} */
#define PGM_C3_SBLK 4096     /* float sblk[6][8][64]: score blocks of the six row helpers */
#define PGM_C3_BYTES (4096 + 6 * 2048)
#define PGM_C3_FLAG 3
#define PGM_C3_H0 4          /* first helper counter of the even steps; + 8: of the odd steps */

// Maxima without the compiler's canonicalisation of operands that come straight from memory (v_max_f32 x, x, x in front of every
// fmaxf of a loaded value: four extra instructions per step of the chain wavefront).  No operand here is ever a NaN.
__device__ __forceinline__ float pgm_max2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float pgm_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---------------------------------------------------------------------------------------------
// CHAIN wavefront.
// (DBG, tools build: dbg[0..3] = ticks of 10 ns in all / waiting for the fold wavefronts / waiting for the band above, waits entered)
template <bool DBG>
__device__ __forceinline__ void pgm_crit_sweep(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *abort_flag, bool &aborted,
                                               const uint32_t spin_limit, const bool stall, int *sw_generic, unsigned long long *dbg) {
    unsigned long long d_fold = 0, d_prev = 0, d_n = 0;
    const unsigned long long d_t0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull, d_c0 = DBG ? __builtin_readcyclecounter() : 0ull;
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING, RS = 5;
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef float pgm_v2f __attribute__((ext_vector_type(2)));
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;
    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nb = J.nb;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap, s_init = J.sc.start_init;
    const uint32_t D = J.hD, Dm = D - 1u, DX = J.hDX, DXm = DX - 1u;
    const bool has_far = J.has_far != 0;
    float *hW = (float *)slot, *hY = hW + D * HS, *hX = hY + D * HS;
    float4 *ring3 = (float4 *)(hX + DX * 64u);
    float *res = (float *)(slot + J.aux_off + PGM_AUX_RES);
    float *pre = (float *)(slot + J.c3_off + PGM_C3_PRE);
    pgm_v2f *colA = (pgm_v2f *)(slot + J.c3_off + PGM_C3_COLA);
    const uint32_t y = 64u * b + (uint32_t)lane;
    const bool rowvalid = y + 1 < n1;
    const uint32_t yc = rowvalid ? y : 0u;
    const float4 r0 = pgm_gload4((const float4 *)(J.ni1 + yc));
    const float ccy = r0.x;
    const float gopen_x = (rowvalid && y == 0) ? sg : gi;
    const uint32_t ncol_row = rowvalid ? ncol : 0u;
    const int x_init = (rowvalid && y == 0) ? 0 : -0x40000000;
#ifdef PGM_X_NOHAND   /* timing experiment (results are wrong): every band sweeps as if it were the first, nothing handed over */
    const bool has_next = false, has_prev = false;
#else
    const bool has_next = (b + 1 < nb), has_prev = (b > 0);
#endif
    float4 *cells_band = J.cells + (size_t)b * tsteps * 64u;
    const __amdgpu_buffer_rsrc_t cells_rsrc = pgm_band_rsrc(cells_band, tsteps * 1024u);
    const float4 *cells_prev = J.cells + (size_t)(b - 1) * tsteps * 64u;
    const float4 *ni2q = (const float4 *)J.ni2;
    const uint32_t lb = (uint32_t)(VL + lane);

    for (uint32_t i = (uint32_t)lane; i < D * HS; i += 64u) { hW[i] = PGM_NEG_INF; hY[i] = PGM_NEG_INF; }
    for (uint32_t i = (uint32_t)lane; i < DX * 64u; i += 64u) hX[i] = PGM_NEG_INF;
    for (int i = lane; i < NR * RS; i += 64) ring3[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // "column < 0" slots
    for (int i = lane; i < NR; i += 64) colA[i] = pgm_v2f{0.f, gi};
    for (int i = lane; i < 4 * 192; i += 64) { res[i] = PGM_NEG_INF; pre[i] = PGM_NEG_INF; }

    // ---- block prefetch: column summaries (for every wavefront of the worker) and the virtual lanes, as in pgm_sweep_band<2> ----
    float4 pfq;
    float2 pfr;
    const int rq_col = lane / RS, rq_part = lane % RS;
    auto load_ring_block = [&](uint32_t c0) {
        const uint32_t col = c0 + (uint32_t)rq_col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < RS * BL && col <= ncol) v = pgm_gload4(ni2q + 5u * col + (uint32_t)rq_part);
        pfq = v;
    };
    auto store_ring_block = [&](uint32_t c0) {
        if (lane < RS * BL) {
            const uint32_t cs = (c0 + (uint32_t)rq_col) & (uint32_t)(NR - 1);
            ring3[(uint32_t)rq_part * (uint32_t)NR + cs] = pfq;
            if (rq_part == 0) colA[cs] = pgm_v2f{pfq.x, (c0 + (uint32_t)rq_col) == 0u ? sg : gi};
        }
    };
    auto load_rep_half = [&](int s0) {
        const int s = s0 + lane / VL, v = lane % VL;
        const int col = s + (VL - v);
        float2 val = make_float2(PGM_NEG_INF, PGM_NEG_INF);
        if (has_prev && col >= 0 && col < (int)ncol) val = pgm_gload_cell_wy(cells_prev + (size_t)(s + 64) * 64u + (uint32_t)(64 - VL + v));
        pfr = val;
    };
    auto store_rep_half = [&](int s0) {
        const int s = s0 + lane / VL, v = lane % VL;
        const uint32_t idx = ((uint32_t)s & Dm) * HS + (uint32_t)v;
        hW[idx] = pfr.x;
        hY[idx] = pfr.y;
    };
    int seen = has_prev ? 0 : 0x7fffffff, pend = 0;
    auto wait_prev = [&](uint32_t steps_needed) {
        if (seen != 0x7fffffff && !aborted) {
            const int need = (int)min(steps_needed, tsteps);
            uint32_t spins = 0;
            const unsigned long long w0 = (DBG && seen < need) ? __builtin_amdgcn_s_memrealtime() : 0ull;
            while (seen < need) {
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)&J.prog[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= need) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > spin_limit || __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    aborted = true;
                    break;
                }
            }
            if (DBG && w0) d_prev += __builtin_amdgcn_s_memrealtime() - w0;
        }
    };
    auto poll_issue = [&]() { if (seen != 0x7fffffff) pend = __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)&J.prog[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto poll_collect = [&]() { if (seen != 0x7fffffff) seen = max(seen, __builtin_amdgcn_readfirstlane(pend)); };

    load_ring_block(0); store_ring_block(0);
    load_ring_block(BL); store_ring_block(BL);
    wait_prev(BL + 64);
    if (has_prev) {
        if (D >= (uint32_t)(VL + BL)) {
            for (int h = -4; h < 0; ++h) { load_rep_half(h * (BL / 2)); store_rep_half(h * (BL / 2)); }
        } else {
            for (int h = -2; h < 0; ++h) { load_rep_half(h * (BL / 2)); store_rep_half(h * (BL / 2)); }
        }
        load_rep_half(0); store_rep_half(0);
    }
    load_ring_block(2 * BL);
    load_rep_half(BL / 2);
    poll_issue();
    {   // history initialised, first blocks staged: the other wavefronts may start
        if (lane >= PGM_C3_H0 && lane < PGM_C3_H0 + 16) {
            // helper wavefronts that take no part in this band publish nothing: {NEAR, COLUMNS 0, 1, ROWS 0, 1, 2, -, -} per parity
            const int k = (lane - PGM_C3_H0) & 7;
#ifdef PGM_X_NOFAR   /* timing experiment (results are wrong): no far helpers */
            const bool runs = k == 0;
#else
            const bool runs = k == 0 || (has_far && k <= 5);
#endif
            if (!runs) sw[lane] = 0x7fffffff;
        }
        asm volatile("" ::: "memory");
        __hip_atomic_store(sw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    float W_o = PGM_NEG_INF, Y_o = PGM_NEG_INF, X_o = PGM_NEG_INF;
    pgm_v2f ca_n = colA[((uint32_t)(-lane)) & (uint32_t)(NR - 1)];
    float inW1 = hW[(0xFFFFFFFFu & Dm) * HS + VL - 1], inY1 = hY[(0xFFFFFFFFu & Dm) * HS + VL - 1];
    for (uint32_t t0 = 0; !aborted; t0 += BL) {
        // (a block of eight steps never wraps in the history rings: their depths are multiples of eight — the step's slots are the
        // block's base plus a constant)
        const uint32_t hb = (t0 & Dm) * HS, hxb = (t0 & DXm) * 64u;
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const uint32_t t = t0 + i;
            const int xs = (int)t - lane;
            const bool active = (uint32_t)xs < ncol_row;
            const uint32_t x = (uint32_t)xs;
            const pgm_v2f ca = ca_n;
            const float iW1 = inW1, iY1 = inY1;
            // the step's pre-folded maxima, read speculatively behind the counter of the FOLD wavefront that writes them (LDS
            // operations execute in order: if the counter read below shows the step published, these reads saw its values)
            const uint32_t po = (uint32_t)(i & 3) * 192u + (uint32_t)lane;
            int cB = __hip_atomic_load(sw + 1 + (i & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            float pM = pre[po], pX = pre[po + 64], pY = pre[po + 128];
            // operands of step t + 1: the column's chain cost and flags, lane 0's upper neighbour (virtual lane 15 of the history)
            ca_n = colA[(x + 1u) & (uint32_t)(NR - 1)];
            inW1 = hW[hb + (uint32_t)(i * HS + VL - 1)]; inY1 = hY[hb + (uint32_t)(i * HS + VL - 1)];
            const float ccx = ca.x, gopen_y = ca.y;
            const float u1W = pgm_dpp_wave_shr1(W_o, iW1), u1Y = pgm_dpp_wave_shr1(Y_o, iY1);
            float Xv = __fsub_rn(pgm_max2(__fadd_rn(X_o, ge), __fadd_rn(W_o, gopen_x)), ccx);
            float Yv = __fsub_rn(pgm_max2(__fadd_rn(u1Y, ge), __fadd_rn(u1W, gopen_y)), ccy);
            const int want = (int)t + 1;
            asm volatile("" : "+v"(Xv), "+v"(Yv), "+v"(cB));   // (the chain terms are issued before the wavefront waits for the counter)
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(cB) < want, 0)) {
                uint32_t spins = 0;
                const unsigned long long w0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
                for (;;) {
                    cB = __hip_atomic_load(sw + 1 + (i & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                    pM = pre[po]; pX = pre[po + 64]; pY = pre[po + 128];
                    asm volatile("" ::: "memory");
                    if (__builtin_amdgcn_readfirstlane(cB) == 0x7fffffff) { aborted = __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
                    if (__builtin_amdgcn_readfirstlane(cB) >= want) break;
                    if (++spins > (1u << 22)) { __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
                }
                if (DBG) { d_fold += __builtin_amdgcn_s_memrealtime() - w0; ++d_n; }
            }
            float Mv = pM;
            Xv = pgm_max2(Xv, pX);
            Yv = pgm_max2(Yv, pY);
            float Wv = pgm_max3(Mv, Xv, Yv);
            if (xs == x_init) Wv = s_init;
            Mv = active ? Mv : PGM_NEG_INF; Xv = active ? Xv : PGM_NEG_INF; Yv = active ? Yv : PGM_NEG_INF; Wv = active ? Wv : PGM_NEG_INF;
#ifndef PGM_X_NOSTORE   /* timing experiment: no cell stores */
            pgm_store_cell_masked(cells_rsrc, t, lane, active, Mv, Xv, Wv, Yv);
#endif
            {
                const uint32_t ho = hb + lb + (uint32_t)(i * HS);
                hW[ho] = Wv;
                hY[ho] = Yv;
                hX[hxb + (uint32_t)lane + (uint32_t)(i * 64)] = Xv;
                asm volatile("" ::: "memory");
                __hip_atomic_store(sw, (int)t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            W_o = Wv; Y_o = Yv; X_o = Xv;
            if (i == BL / 2 - 1 && has_next && !stall) {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BL / 2) : "memory");
                if (lane == 0) __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)&J.prog[b], (int)t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (i == BL / 2 - 1) {
                store_rep_half((int)(t0 + BL / 2));
                poll_collect();
                wait_prev(t0 + BL + 64 + BL / 2);
                load_rep_half((int)(t0 + BL));
                poll_issue();
            }
        }
        if (has_next && !stall) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BL / 2 + 1) : "memory");
            if (lane == 0) __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)&J.prog[b], (int)t0 + BL / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t t1 = t0 + BL;
        if (t1 >= tsteps) break;
        store_ring_block(t1 + BL);
        store_rep_half((int)t1);
        load_ring_block(t1 + 2 * BL);
        poll_collect();
        wait_prev(t1 + BL + 64);
        load_rep_half((int)(t1 + BL / 2));
        poll_issue();
    }
    __hip_atomic_store(sw, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // releases the other wavefronts (also after an abort)
    if (DBG && dbg && lane == 0) { dbg[0] = __builtin_amdgcn_s_memrealtime() - d_t0; dbg[1] = d_fold; dbg[2] = d_prev; dbg[3] = d_n; dbg[60] = __builtin_readcyclecounter() - d_c0; }   // ([60]: shader cycles of the band)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && !stall) __hip_atomic_store(&J.prog[b], aborted ? (int)0 : (int)0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Shared set-up of the FOLD and NEAR wavefronts: the row's edge costs, the band's emission scores a block ahead.
struct PgmCritRow {
    float ccy, c2y, c3y, gopen_x;
};
__device__ __forceinline__ PgmCritRow pgm_crit_row(const PgmJob &J, const uint32_t b, const int lane) {
    const uint32_t y = 64u * b + (uint32_t)lane;
    const bool rowvalid = y + 1 < J.n1;
    const float4 r0 = pgm_gload4((const float4 *)(J.ni1 + (rowvalid ? y : 0u)));
    PgmCritRow r;
    r.ccy = r0.x; r.c2y = r0.y; r.c3y = r0.z;     // (a crit3 job has no generic rows)
    r.gopen_x = (rowvalid && y == 0) ? J.sc.start_gap : J.sc.gap_init;
    return r;
}

// ---------------------------------------------------------------------------------------------
// FOLD wavefront of the steps t % 2 == Q: the three terms that read step t - 2 — M from (y-1, x-1), X from column x-2, Y from row
// y-2 — merged with everything the NEAR and far helper wavefronts have folded for the step (res[t & 3], read and reset here) into
// the step's pre-folded maxima.  This is the only work between "step t - 2 recorded" and "step t may be merged": kept short.
// (DBG: dbg[0..3] = ticks in all / waiting for the record of step t - 2 / waiting for the helpers, steps whose maxima were taken late)
template <int Q, bool DBG>
__device__ __forceinline__ void pgm_crit_fold(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *sw_generic, int *abort_flag, unsigned long long *dbg) {
    unsigned long long d_rec = 0, d_help = 0, d_late = 0;
    const unsigned long long d_t0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING;
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef int pgm_v4i __attribute__((ext_vector_type(4)));
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;
    const uint32_t tsteps = J.tsteps, nblk = J.nblk;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap;
    const uint32_t D = J.hD, Dm = D - 1u, DX = J.hDX, DXm = DX - 1u;
    const float *hW = (const float *)slot, *hY = hW + D * HS, *hX = hY + D * HS;
    const float4 *ring3 = (const float4 *)(hX + DX * 64u);
    float *res = (float *)(slot + J.aux_off + PGM_AUX_RES);
    float *pre = (float *)(slot + J.c3_off + PGM_C3_PRE);
    const PgmCritRow R = pgm_crit_row(J, b, lane);
    const float ccy = R.ccy, c2y = R.c2y, gopen_x = R.gopen_x;
    const float4 *S_band = (const float4 *)(J.S + (size_t)b * nblk * 64u * BL);
    const uint32_t lb = (uint32_t)(VL + lane);
    float4 pfs[BL / 4];
    auto load_s_block = [&](uint32_t s0) {
        const uint32_t tb = min(s0 / BL, nblk - 1u);
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) pfs[k] = pgm_gload4(S_band + ((size_t)tb * 64u + (uint32_t)lane) * (BL / 4) + k);
    };
    load_s_block(0);
    int seen = 0, seen_h = 0;
    pgm_lds_int *hcnt = sw + PGM_C3_H0 + 8 * Q;
    // nothing of the worker's LDS is read before the chain wavefront's prologue has initialised it (column ring, history, maxima)
    while (seen < 1) seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
    bool gone = false;   // the chain wavefront is through (its word says "far ahead"), or a wait ran into its limit (abort raised)
    for (uint32_t t0 = 0; t0 < tsteps && !gone; t0 += BL) {
        float Sc[BL];
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) { Sc[4 * k] = pfs[k].x; Sc[4 * k + 1] = pfs[k].y; Sc[4 * k + 2] = pfs[k].z; Sc[4 * k + 3] = pfs[k].w; }
        load_s_block(t0 + BL);
#pragma unroll
        for (int i = Q; i < BL; i += 2) {
            const uint32_t t = t0 + (uint32_t)i;
            const int xs = (int)t - lane;
            const uint32_t rslot = ((uint32_t)xs) & (uint32_t)(NR - 1);
            const float S = Sc[i];
            const float gopen_y = (xs == 0) ? sg : gi;
            const float4 cn = ring3[rslot];
            const uint32_t ro = (t & 3u) * 192u + (uint32_t)lane;
            // the helpers' maxima of this step: taken now if they are all through with it (they run up to four steps ahead) ...
            float rM = PGM_NEG_INF, rX = PGM_NEG_INF, rY = PGM_NEG_INF;
            bool merged = false;
            auto helpers_done = [&]() {
                const pgm_v4i wa = *(const __attribute__((address_space(3))) pgm_v4i *)hcnt, wb = *(const __attribute__((address_space(3))) pgm_v4i *)(hcnt + 4);
                asm volatile("" ::: "memory");
                int m = min(min(wa.x, wa.y), min(wa.z, wa.w));
                m = min(m, min(min(wb.x, wb.y), min(wb.z, wb.w)));
                seen_h = __builtin_amdgcn_readfirstlane(m);
                return seen_h >= (int)t + 1;
            };
            auto take = [&]() {
                rM = res[ro]; rX = res[ro + 64]; rY = res[ro + 128];
                res[ro] = PGM_NEG_INF; res[ro + 64] = PGM_NEG_INF; res[ro + 128] = PGM_NEG_INF;
                merged = true;
            };
            if (seen_h >= (int)t + 1 || helpers_done()) take();
            // ---- step t - 2 recorded (word >= t) ----
            {
                const int need = max(1, (int)t);
                uint32_t spins = 0;
                const unsigned long long w0 = (DBG && seen < need) ? __builtin_amdgcn_s_memrealtime() : 0ull;
                while (seen < need) {
                    seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (seen < need && ++spins > (1u << 24)) { __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); gone = true; break; }
                }
                asm volatile("" ::: "memory");
                if (DBG && w0) d_rec += __builtin_amdgcn_s_memrealtime() - w0;
                if (seen == 0x7fffffff) gone = true;
            }
            const float w11 = hW[((t - 2u) & Dm) * HS + lb - 1u];
            const float x2 = hX[((t - 2u) & DXm) * 64u + (uint32_t)lane], wx2 = hW[((t - 2u) & Dm) * HS + lb];
            const float y2 = hY[((t - 2u) & Dm) * HS + lb - 2u], wy2 = hW[((t - 2u) & Dm) * HS + lb - 2u];
            float Mv = __fsub_rn(__fsub_rn(__fadd_rn(w11, S), ccy), cn.x);
            float Xv = __fsub_rn(fmaxf(__fadd_rn(x2, ge), __fadd_rn(wx2, gopen_x)), cn.y);
            float Yv = __fsub_rn(fmaxf(__fadd_rn(y2, ge), __fadd_rn(wy2, gopen_y)), c2y);
            if (!merged) {   // ... or as soon as they are
                uint32_t spins = 0;
                const unsigned long long w0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
                while (!gone && !helpers_done()) {
                    if (++spins > (1u << 22)) { __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); gone = true; }
                    else if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0x7fffffff) gone = true;
                }
                if (DBG) { d_help += __builtin_amdgcn_s_memrealtime() - w0; ++d_late; }
                take();
            }
            Mv = fmaxf(Mv, rM); Xv = fmaxf(Xv, rX); Yv = fmaxf(Yv, rY);
            pre[ro] = Mv; pre[ro + 64] = Xv; pre[ro + 128] = Yv;
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + 1 + Q, (int)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __hip_atomic_store(sw + 1 + Q, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (DBG && dbg && lane == 0) { dbg[0] = __builtin_amdgcn_s_memrealtime() - d_t0; dbg[1] = d_rec; dbg[2] = d_help; dbg[3] = d_late; }
}

// ---------------------------------------------------------------------------------------------
// NEAR wavefront of the steps t % 2 == Q: the ten near terms that read step t - 3 and older — the M pairs (y-1, x-2), (y-2, x-1)
// [t - 3], (y-1, x-3), (y-2, x-2), (y-3, x-1) [t - 4], (y-2, x-3), (y-3, x-2) [t - 5], (y-3, x-3) [t - 6], X from column x-3 and
// Y from row y-3 [t - 3] — folded into res[t & 3] like the far helpers' terms (three steps of lead).
template <int Q, bool DBG>
__device__ __forceinline__ void pgm_crit_near(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *sw_generic, const int hidx, unsigned long long *dbg) {
    unsigned long long d_wait = 0;
    const unsigned long long d_t0 = DBG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING;
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef __attribute__((address_space(3))) float pgm_lds_float;
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;
    const uint32_t tsteps = J.tsteps, nblk = J.nblk;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap;
    const uint32_t D = J.hD, Dm = D - 1u, DX = J.hDX, DXm = DX - 1u;
    const float *hW = (const float *)slot, *hY = hW + D * HS, *hX = hY + D * HS;
    const float4 *ring3 = (const float4 *)(hX + DX * 64u);
    float *res = (float *)(slot + J.aux_off + PGM_AUX_RES);
    const PgmCritRow R = pgm_crit_row(J, b, lane);
    const float ccy = R.ccy, c2y = R.c2y, c3y = R.c3y, gopen_x = R.gopen_x;
    const float4 *S_band = (const float4 *)(J.S + (size_t)b * nblk * 64u * BL);
    const uint32_t lb = (uint32_t)(VL + lane);
    float4 pfs[BL / 4];
    auto load_s_block = [&](uint32_t s0) {
        const uint32_t tb = min(s0 / BL, nblk - 1u);
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) pfs[k] = pgm_gload4(S_band + ((size_t)tb * 64u + (uint32_t)lane) * (BL / 4) + k);
    };
    load_s_block(0);
    int seen = 0;
    for (uint32_t t0 = 0; t0 < tsteps && seen != 0x7fffffff; t0 += BL) {
        float Sc[BL];
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) { Sc[4 * k] = pfs[k].x; Sc[4 * k + 1] = pfs[k].y; Sc[4 * k + 2] = pfs[k].z; Sc[4 * k + 3] = pfs[k].w; }
        load_s_block(t0 + BL);
#pragma unroll
        for (int i = Q; i < BL; i += 2) {
            const uint32_t t = t0 + (uint32_t)i;
            const int need = max(1, (int)t - 1);      // step t - 3 recorded
            const unsigned long long w0 = (DBG && seen < need) ? __builtin_amdgcn_s_memrealtime() : 0ull;
            while (seen < need) seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            asm volatile("" ::: "memory");
            if (DBG && w0) d_wait += __builtin_amdgcn_s_memrealtime() - w0;
            const int xs = (int)t - lane;
            const float4 cn = ring3[((uint32_t)xs) & (uint32_t)(NR - 1)];
            const float S = Sc[i];
            const float gopen_y = (xs == 0) ? sg : gi;
            auto w_at = [&](uint32_t dy, uint32_t dx) { return hW[((t - (dy + dx)) & Dm) * HS + lb - dy]; };
            auto mt = [&](float w, float cy, float cx) { return __fsub_rn(__fsub_rn(__fadd_rn(w, S), cy), cx); };
            const float w12 = w_at(1, 2), w21 = w_at(2, 1), w13 = w_at(1, 3), w22 = w_at(2, 2), w31 = w_at(3, 1), w23 = w_at(2, 3), w32 = w_at(3, 2), w33 = w_at(3, 3);
            const float x3 = hX[((t - 3u) & DXm) * 64u + (uint32_t)lane], wx3 = w_at(0, 3);
            const float y3 = hY[((t - 3u) & Dm) * HS + lb - 3u], wy3 = w_at(3, 0);
            const float ccx = cn.x, c2x = cn.y, c3x = cn.z;
            float Mv = fmaxf(fmaxf(mt(w12, ccy, c2x), mt(w21, c2y, ccx)), fmaxf(fmaxf(mt(w13, ccy, c3x), mt(w22, c2y, c2x)), mt(w31, c3y, ccx)));
            Mv = fmaxf(Mv, fmaxf(fmaxf(mt(w23, c2y, c3x), mt(w32, c3y, c2x)), mt(w33, c3y, c3x)));
            const float Xv = __fsub_rn(fmaxf(__fadd_rn(x3, ge), __fadd_rn(wx3, gopen_x)), c3x);
            const float Yv = __fsub_rn(fmaxf(__fadd_rn(y3, ge), __fadd_rn(wy3, gopen_y)), c3y);
            float *rs = res + (t & 3u) * 192u + (uint32_t)lane;
            __builtin_amdgcn_ds_fmaxf((pgm_lds_float *)rs, Mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            __builtin_amdgcn_ds_fmaxf((pgm_lds_float *)(rs + 64), Xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            __builtin_amdgcn_ds_fmaxf((pgm_lds_float *)(rs + 128), Yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + hidx, (int)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (DBG && dbg && lane == 0) { dbg[0] = __builtin_amdgcn_s_memrealtime() - d_t0; dbg[1] = d_wait; }
}

// ---------------------------------------------------------------------------------------------
// Far helpers of pgm_crit_kernel.  Same terms, same operations as pgm_terms_helper<1> / <2> without the LONG / overflow branches
// (a crit3 job has neither); what differs is the schedule.  A helper's step was a chain of five or six dependent LDS round trips
// (poll, column flags, entry lists, history, fold) of 150-200 cycles each beside fifteen other wavefronts: ~2300 cycles, all
// latency.  Only the history reads depend on what the chain wavefront has recorded; the column summaries are graph data staged
// blocks ahead.  So every step fetches the flags and entry lists of the wavefront's NEXT step while its own history reads are in
// flight, and the history is addressed with 24-bit multiplies on LDS offsets instead of 64-bit pointer arithmetic.
struct PgmCritHist {
    const __attribute__((address_space(3))) float *hW, *hY, *hX;
    uint32_t Dm, DXm;
    __device__ __forceinline__ uint32_t row(uint32_t s) const { return __umul24(s & Dm, 64u + PGM_VL); }
    __device__ __forceinline__ float W(uint32_t s, uint32_t l) const { return hW[row(s) + l]; }
    __device__ __forceinline__ float Y(uint32_t s, uint32_t l) const { return hY[row(s) + l]; }
    __device__ __forceinline__ float X(uint32_t s, uint32_t lane) const { return hX[((s & DXm) << 6) + lane]; }
};

// COLUMNS: the far edges of the columns, one row per lane; this wavefront: the entries j = part, part + 2, ... of the steps t % 2 == Q.
template <int Q, bool DBG>
__device__ __forceinline__ void pgm_crit_cols(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *sw_generic, const int hidx,
                                              const uint32_t part, unsigned long long *hst) {
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING, KF = PGM_KF8;
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef __attribute__((address_space(3))) float pgm_lds_float;
    typedef __attribute__((address_space(3))) const pgm_v4f pgm_lds_cf4;
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;
    const uint32_t tsteps = J.tsteps, nblk = J.nblk, D = J.hD, DX = J.hDX;
    const float ge = J.sc.gap_extend;
    PgmCritHist H;
    H.hW = (const pgm_lds_float *)slot; H.hY = H.hW + D * HS; H.hX = H.hY + D * HS; H.Dm = D - 1u; H.DXm = DX - 1u;
    pgm_lds_cf4 *ring = (pgm_lds_cf4 *)(H.hX + DX * 64u);
    pgm_lds_float *res = (pgm_lds_float *)(slot + J.aux_off + PGM_AUX_RES);
    const PgmCritRow R = pgm_crit_row(J, b, lane);
    const float ccy = R.ccy, c2y = R.c2y, c3y = R.c3y, gopen_x = R.gopen_x;
    const uint32_t ncol_row = (64u * b + (uint32_t)lane + 1u < J.n1) ? J.ncol : 0u;
    const float4 *S_band = (const float4 *)(J.S + (size_t)b * nblk * 64u * BL);
    const uint32_t lb = (uint32_t)(VL + lane);
    const int slack = (int)J.far_slack;
    float4 pfs[BL / 4];
    auto load_s_block = [&](uint32_t s0) {
        const uint32_t tb = min(s0 / BL, nblk - 1u);
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) pfs[k] = pgm_gload4(S_band + ((size_t)tb * 64u + (uint32_t)lane) * (BL / 4) + k);
    };
    load_s_block(0);
    int seen = 0;
    unsigned long long hwait = 0;
    const unsigned long long ht0 = (DBG && hst) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto wait_rec = [&](int need) {
        const unsigned long long w0 = (DBG && hst && seen < need) ? __builtin_amdgcn_s_memrealtime() : 0ull;
        while (seen < need) {
            seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (seen < need) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (DBG && w0) hwait += __builtin_amdgcn_s_memrealtime() - w0;
    };
    wait_rec(1);   // (the column ring is staged by the chain wavefront's prologue)
    // entry lists of the wavefront's next step (graph data: fetched a step of its own ahead)
    bool s_any = false;
    int s_nw = 0;
    pgm_v4f s_d0 = {0.f, 0.f, 0.f, 0.f}, s_c0 = s_d0, s_d1 = s_d0, s_c1 = s_d0;
    auto fetch = [&](uint32_t t) {
        const int xs = (int)t - lane;
        const uint32_t rs = ((uint32_t)xs) & (uint32_t)(NR - 1);
        const uint32_t fl = __float_as_uint(((const pgm_lds_float *)ring)[rs * 4u + 3u]);
        const uint32_t nfx = (uint32_t)xs < ncol_row ? (fl & PGM_NF_COUNT) : 0u;
        s_any = __builtin_amdgcn_ballot_w64(nfx > part) != 0ull;
        if (s_any) {
            s_nw = pgm_wave_max8(nfx);
            s_d0 = ring[rs + (uint32_t)NR]; s_c0 = ring[rs + 3u * (uint32_t)NR];
            if (s_nw > 4) { s_d1 = ring[rs + 2u * (uint32_t)NR]; s_c1 = ring[rs + 4u * (uint32_t)NR]; }
        }
    };
    fetch((uint32_t)Q);
    for (uint32_t t0 = 0; t0 < tsteps && seen != 0x7fffffff; t0 += BL) {
        float Sc[BL];
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) { Sc[4 * k] = pfs[k].x; Sc[4 * k + 1] = pfs[k].y; Sc[4 * k + 2] = pfs[k].z; Sc[4 * k + 3] = pfs[k].w; }
        load_s_block(t0 + BL);
#pragma unroll
        for (int i = Q; i < BL; i += 2) {
            const uint32_t t = t0 + (uint32_t)i;
            const bool any = s_any;
            const int nw = s_nw;
            const pgm_v4f d0 = s_d0, c0 = s_c0, d1 = s_d1, c1 = s_c1;
            wait_rec(max(1, (int)t - slack + 2));
            float Mf = PGM_NEG_INF, Xf = PGM_NEG_INF;
            if (any) {
                const float S = Sc[i];
                const uint32_t fdx[KF] = {__float_as_uint(d0.x), __float_as_uint(d0.y), __float_as_uint(d0.z), __float_as_uint(d0.w),
                                          __float_as_uint(d1.x), __float_as_uint(d1.y), __float_as_uint(d1.z), __float_as_uint(d1.w)};
                const float fcx[KF] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int j = 0; j < KF; ++j) {
                    if (j < nw && ((uint32_t)j & 1u) == part) {   // (an absent slot has distance 0 and cost +inf: its terms are -inf)
                        const uint32_t s1 = t - fdx[j];
                        const float cj = (j < 4 || nw > 4) ? fcx[j] : INFINITY;
                        const float Xh = H.X(s1, (uint32_t)lane), Wh = H.W(s1, lb);
                        const float W1 = H.W(s1 - 1u, lb - 1u), W2 = H.W(s1 - 2u, lb - 2u), W3 = H.W(s1 - 3u, lb - 3u);
                        Xf = fmaxf(Xf, __fsub_rn(fmaxf(__fadd_rn(Xh, ge), __fadd_rn(Wh, gopen_x)), cj));
                        Mf = fmaxf(Mf, fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W1, S), ccy), cj),
                                             fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W2, S), c2y), cj), __fsub_rn(__fsub_rn(__fadd_rn(W3, S), c3y), cj))));
                    }
                }
            }
            fetch(t + 2u);
            if (any && __builtin_amdgcn_ballot_w64(Mf > PGM_NEG_INF || Xf > PGM_NEG_INF) != 0ull) {
                pgm_lds_float *rs = res + (t & 3u) * 192u + (uint32_t)lane;
                __builtin_amdgcn_ds_fmaxf(rs, Mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
                __builtin_amdgcn_ds_fmaxf(rs + 64, Xf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            }
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + hidx, (int)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (DBG && hst && lane == 0) { hst[hidx] = hwait; hst[8 + hidx] = __builtin_amdgcn_s_memrealtime() - ht0; }
}

// ROWS: the far edges of the rows, one (row, far edge) entry per lane (pgm_terms_helper<2>: the band's entry list, passes of 64
// entries; a band whose entries fit one pass — the usual case — is split by term: part 0 the Y term and the pairs with the three
// near columns, parts 1 and 2 the pairs with every other far edge of the entry's column; otherwise part k takes the passes k, k + 3,
// k + 6).  This wavefront: the steps t % 2 == Q.  The column summary and the far-edge lists of the FIRST pass are fetched a step of
// its own ahead (see above); `builder`: this wavefront builds the band's entry list.
template <int Q, bool DBG>
__device__ __forceinline__ void pgm_crit_rows(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *sw_generic, const int hidx,
                                              const uint32_t part, float *sblk_generic, const bool builder, unsigned long long *hst) {
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING, KF = PGM_KF8, KQ = 3;
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef __attribute__((address_space(3))) float pgm_lds_float;
    typedef __attribute__((address_space(3))) const pgm_v4f pgm_lds_cf4;
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;
    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nblk = J.nblk, D = J.hD, DX = J.hDX;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap;
    PgmCritHist H;
    H.hW = (const pgm_lds_float *)slot; H.hY = H.hW + D * HS; H.hX = H.hY + D * HS; H.Dm = D - 1u; H.DXm = DX - 1u;
    pgm_lds_cf4 *ring = (pgm_lds_cf4 *)(H.hX + DX * 64u);
    uint8_t *aux = slot + J.aux_off;
    pgm_lds_float *res = (pgm_lds_float *)(aux + PGM_AUX_RES), *sblk = (pgm_lds_float *)sblk_generic;
    uint2 *elist = (uint2 *)(aux + PGM_AUX_EL);
    int *ecnt = (int *)(aux + PGM_AUX_CNT);
    const uint32_t y = 64u * b + (uint32_t)lane;
    const bool rowvalid = y + 1 < n1;
    const uint32_t yc = rowvalid ? y : 0u;
    const float4 *S_band = (const float4 *)(J.S + (size_t)b * nblk * 64u * BL);
    const int slack = (int)J.far_slack;
    // ---- the band's entry list (as pgm_terms_helper<2>; a crit3 job has no remote entries and no generic rows) ----
    if (builder) {
        __hip_atomic_store(ecnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(ecnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int32_t f0 = rowvalid ? pgm_gld(J.fp1 + yc) : 0, f1 = rowvalid ? pgm_gld(J.fp1 + yc + 1) : 0;
        const uint32_t nloc = (uint32_t)(f1 - f0);
        uint32_t base = 0;
        if (nloc) base = (uint32_t)__hip_atomic_fetch_add(ecnt + 1, (int)nloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int32_t e = f0; e < f1; ++e) {
            const uint32_t dx = pgm_gld(&J.fe1[e].x), cb = pgm_gld(&J.fe1[e].y);
            const uint32_t pos = base++;
            if (pos < 512u) elist[pos] = make_uint2((uint32_t)lane | ((dx & 0x7fffffu) << 8), cb);
        }
        asm volatile("" ::: "memory");
        __hip_atomic_store(sw + PGM_C3_FLAG, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(sw + PGM_C3_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    }
    const int ne = min(__builtin_amdgcn_readfirstlane(__hip_atomic_load(ecnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)), 512);
    const bool single = ne <= 64;
    auto p_of = [&](int q) { return single ? (q == 0 ? 0u : 64u) : part + (uint32_t)PGM_CPARTS * (uint32_t)q; };
    uint32_t e_o[KQ], e_dy[KQ];
    float e_cy[KQ];
    bool e_ok[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int idx = (int)p_of(q) * 64 + lane;
        e_ok[q] = idx < ne;
        const uint2 a = e_ok[q] ? elist[idx] : make_uint2((uint32_t)lane | (1u << 8), __float_as_uint(INFINITY));
        e_o[q] = a.x & 255u; e_dy[q] = (a.x >> 8) & 0x7fffffu; e_cy[q] = __uint_as_float(a.y);
    }
    const bool do_near = !single || part == 0u, do_pairs = !single || part >= 1u;
    const uint32_t pj0 = single ? part - 1u : 0u, pjs = single ? 2u : 1u;   // this wavefront's far column entries: j = pj0, pj0 + pjs, ...
    if ((int)p_of(0) * 64 >= ne) {   // no pass of the entry list for this wavefront: nothing to publish but "done"
        __hip_atomic_store(sw + hidx, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    float4 pfs[BL / 4];
    auto load_s_block = [&](uint32_t s0) {
        const uint32_t tb = min(s0 / BL, nblk - 1u);
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) pfs[k] = pgm_gload4(S_band + ((size_t)tb * 64u + (uint32_t)lane) * (BL / 4) + k);
    };
    load_s_block(0);
    int seen = 0;
    unsigned long long hwait = 0;
    const unsigned long long ht0 = (DBG && hst) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto wait_rec = [&](int need) {
        const unsigned long long w0 = (DBG && hst && seen < need) ? __builtin_amdgcn_s_memrealtime() : 0ull;
        while (seen < need) {
            seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (seen < need) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (DBG && w0) hwait += __builtin_amdgcn_s_memrealtime() - w0;
    };
    wait_rec(1);
    // one pass of one step: the entry's column summary `cno` and, for the pairs, the lists g* of that column's far edges (nw: the
    // largest count over the wavefront, 0: no pairs in this step)
    auto pass = [&](const uint32_t t, const int i, const int q, const pgm_v4f cno, const int nw, const pgm_v4f g1, const pgm_v4f g2, const pgm_v4f g1b, const pgm_v4f g2b) {
        const uint32_t o = e_o[q], dy = e_dy[q];
        const float cy = e_cy[q];
        const int xo = (int)t - (int)o;
        const float So = sblk[i * 64 + (int)o];
        const float gopen_y = (xo == 0) ? sg : gi;
        const uint32_t s1 = t - dy, lp = (uint32_t)VL + o - dy;
        float Yt = PGM_NEG_INF, Mt = PGM_NEG_INF;
        if (do_near) {
            const float Yh = H.Y(s1, lp), Wh = H.W(s1, lp), W1 = H.W(s1 - 1u, lp), W2 = H.W(s1 - 2u, lp), W3 = H.W(s1 - 3u, lp);
            Yt = __fsub_rn(fmaxf(__fadd_rn(Yh, ge), __fadd_rn(Wh, gopen_y)), cy);
            Mt = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W1, So), cy), cno.x),
                       fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W2, So), cy), cno.y), __fsub_rn(__fsub_rn(__fadd_rn(W3, So), cy), cno.z)));
        }
        if (nw > 0) {
            const uint32_t gdx[KF] = {__float_as_uint(g1.x), __float_as_uint(g1.y), __float_as_uint(g1.z), __float_as_uint(g1.w),
                                      __float_as_uint(g1b.x), __float_as_uint(g1b.y), __float_as_uint(g1b.z), __float_as_uint(g1b.w)};
            const float gcx[KF] = {g2.x, g2.y, g2.z, g2.w, g2b.x, g2b.y, g2b.z, g2b.w};
#pragma unroll
            for (int j = 0; j < KF; ++j) {
                if (j < nw && (pjs == 1u || ((uint32_t)j & 1u) == pj0)) {
                    const float Wp = H.W(s1 - gdx[j], lp);
                    Mt = fmaxf(Mt, __fsub_rn(__fsub_rn(__fadd_rn(Wp, So), cy), gcx[j]));
                }
            }
        }
        if (e_ok[q]) {
            pgm_lds_float *rs = res + (t & 3u) * 192u + o;
            if (do_near || nw > 0) __builtin_amdgcn_ds_fmaxf(rs, Mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            if (do_near) __builtin_amdgcn_ds_fmaxf(rs + 128, Yt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
        }
    };
    // what a pass needs of its column (graph data)
    pgm_v4f s_cno = {0.f, 0.f, 0.f, 0.f}, s_g1 = s_cno, s_g2 = s_cno, s_g1b = s_cno, s_g2b = s_cno;
    int s_nw = 0;
    auto fetch = [&](const uint32_t t, const int q, pgm_v4f &cno, int &nw, pgm_v4f &g1, pgm_v4f &g2, pgm_v4f &g1b, pgm_v4f &g2b) {
        const int xo = (int)t - (int)e_o[q];
        const uint32_t rso = (uint32_t)xo & (uint32_t)(NR - 1);
        cno = ring[rso];
        nw = 0;
        if (do_pairs) {
            const uint32_t fow = (e_ok[q] && (uint32_t)xo < ncol) ? __float_as_uint(cno.w) : 0u;
            const uint32_t nfo = fow & PGM_NF_COUNT;
            if (__builtin_amdgcn_ballot_w64(nfo > pj0) != 0ull) {
                nw = pgm_wave_max8(nfo);
                g1 = ring[rso + (uint32_t)NR]; g2 = ring[rso + 3u * (uint32_t)NR];
                g1b = pgm_v4f{0.f, 0.f, 0.f, 0.f}; g2b = pgm_v4f{INFINITY, INFINITY, INFINITY, INFINITY};
                if (nw > 4) { g1b = ring[rso + 2u * (uint32_t)NR]; g2b = ring[rso + 4u * (uint32_t)NR]; }
            }
        }
    };
    fetch((uint32_t)Q, 0, s_cno, s_nw, s_g1, s_g2, s_g1b, s_g2b);
    for (uint32_t t0 = 0; t0 < tsteps && seen != 0x7fffffff; t0 += BL) {
        float Sc[BL];
#pragma unroll
        for (int k = 0; k < BL / 4; ++k) { Sc[4 * k] = pfs[k].x; Sc[4 * k + 1] = pfs[k].y; Sc[4 * k + 2] = pfs[k].z; Sc[4 * k + 3] = pfs[k].w; }
#pragma unroll
        for (int i = 0; i < BL; ++i) sblk[i * 64 + lane] = Sc[i];
        load_s_block(t0 + BL);
#pragma unroll
        for (int i = Q; i < BL; i += 2) {
            const uint32_t t = t0 + (uint32_t)i;
            const pgm_v4f cno = s_cno, g1 = s_g1, g2 = s_g2, g1b = s_g1b, g2b = s_g2b;
            const int nw = s_nw;
            wait_rec(max(1, (int)t - slack + 2));
            pass(t, i, 0, cno, nw, g1, g2, g1b, g2b);
            fetch(t + 2u, 0, s_cno, s_nw, s_g1, s_g2, s_g1b, s_g2b);
            if (!single) {   // a band with more than 64 entries: this wavefront's further passes
#pragma unroll
                for (int q = 1; q < KQ; ++q) {
                    if ((int)p_of(q) * 64 < ne) {
                        pgm_v4f c2, h1, h2, h1b, h2b; int n2;
                        fetch(t, q, c2, n2, h1, h2, h1b, h2b);
                        pass(t, i, q, c2, n2, h1, h2, h1b, h2b);
                    }
                }
            }
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + hidx, (int)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (DBG && hst && lane == 0) { hst[hidx] = hwait; hst[8 + hidx] = __builtin_amdgcn_s_memrealtime() - ht0; }
}

// ---------------------------------------------------------------------------------------------
// One workgroup of sixteen wavefronts per CU; items: single bands of crit3 jobs, in list order through an atomic ticket
// (as pgm_fill_kernel: a job's bands ascending, so the band a sweep waits for is already running).
// DBG (tools build, PGM_C3_DBG): per item 64 words of wait statistics — [4 w ..] wavefront w: see the role functions; [32 + w], [48 + w]:
// ticks in the poll loop / in all of helper wavefront w (pgm_terms_helper's timeline words).
template <bool DBG>
__global__ void __launch_bounds__(64 * PGM_C3_WAVES, 1) pgm_crit_kernel(const PgmJob *__restrict__ jobs, const PgmItem *__restrict__ items, uint32_t nitems,
                                                                     int *__restrict__ sync, uint32_t spin_limit, uint32_t stall_job, uint32_t stall_band, uint32_t ticket_off,
                                                                     unsigned long long *__restrict__ dbg_, uint32_t tbq_off) {
    int *abort_flag = sync;
    __shared__ __attribute__((aligned(16))) struct { uint8_t pool[PGM_POOL]; } L;
    __shared__ int item_lds;
    __shared__ __attribute__((aligned(16))) int fsync[24];
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    bool aborted = false;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int it = -1;
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                it = __hip_atomic_fetch_add(sync + ticket_off, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item_lds = (it >= 0 && (uint32_t)it < nitems) ? it : -1;
            for (int k = 0; k < 24; ++k) fsync[k] = 0;
        }
        __syncthreads();
        const int it = item_lds;
        if (it < 0) break;
        const PgmItem item = items[it];
        const PgmJob &J = jobs[item.job];
        const uint32_t b = item.band;
        const bool last_band = (item.band + item.count == J.nb);
        uint8_t *slot = L.pool;
        unsigned long long *dbg = (DBG && dbg_) ? dbg_ + 64 * (size_t)it : nullptr;
        if (role == 0) {
            __builtin_amdgcn_s_setprio(3);
            const bool stall = item.job == stall_job && b == stall_band;
            pgm_crit_sweep<DBG>(J, b, slot, lane, abort_flag, aborted, spin_limit, stall, fsync, dbg);
        } else if (role <= 2) {
            __builtin_amdgcn_s_setprio(2);
            if (role == 1) pgm_crit_fold<0, DBG>(J, b, slot, lane, fsync, abort_flag, dbg ? dbg + 4 : nullptr); else pgm_crit_fold<1, DBG>(J, b, slot, lane, fsync, abort_flag, dbg ? dbg + 8 : nullptr);
        } else if (role <= 14) {
            __builtin_amdgcn_s_setprio(0);
            // helpers: {NEAR, COLUMNS 0, 1, ROWS 0, 1, 2} x {even, odd steps}
            const uint32_t par = (role - 3) & 1u, kind = (uint32_t)(role - 3) >> 1;   // kind 0: near; 1, 2: columns; 3, 4, 5: rows
            const int hidx = PGM_C3_H0 + 8 * (int)par + (int)kind;
            if (kind == 0u) { if (par == 0u) pgm_crit_near<0, DBG>(J, b, slot, lane, fsync, hidx, dbg ? dbg + 12 : nullptr); else pgm_crit_near<1, DBG>(J, b, slot, lane, fsync, hidx, dbg ? dbg + 16 : nullptr); }
#ifdef PGM_X_NOFAR
            else if (false) {
#else
            else if (J.has_far) {
#endif
                // (the helper writes hst[hidx] = ticks in its poll loop and hst[8 + hidx] = ticks in all: words w and w + 8, w = 20 + k for the
                // wavefronts k = role - 5 < 8, 28 + k for the last two)
                unsigned long long *hst = dbg ? dbg + ((role - 5 < 8) ? 20 + (role - 5) : 28 + (role - 5)) - hidx : nullptr;
                if (kind <= 2u) { if (par == 0u) pgm_crit_cols<0, DBG>(J, b, slot, lane, fsync, hidx, kind - 1u, hst); else pgm_crit_cols<1, DBG>(J, b, slot, lane, fsync, hidx, kind - 1u, hst); }
                else {
                    float *sb = (float *)(slot + J.c3_off + PGM_C3_SBLK) + ((kind - 3u) * 2u + par) * 512u;
                    if (par == 0u) pgm_crit_rows<0, DBG>(J, b, slot, lane, fsync, hidx, kind - 3u, sb, kind == 3u, hst);
                    else pgm_crit_rows<1, DBG>(J, b, slot, lane, fsync, hidx, kind - 3u, sb, false, hst);
                }
            }
        }
        if (last_band) {
            __syncthreads();
            if (threadIdx.x == 0) J.times[0] = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {   // an aborted batch leaves its records here
                J.result->score = 0.f; J.result->n_tr_indels = 0; J.result->len = 0; J.result->status = PGM_ERR_DEVICE; J.hresult->score = 0.f; J.hresult->n_tr_indels = 0; J.hresult->len = 0; __threadfence_system(); __hip_atomic_store(&J.hresult->status, (int32_t)PGM_ERR_DEVICE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (threadIdx.x == 0) pgm_tbq_push(sync, tbq_off, item.job);
        }
    }
}

#endif
