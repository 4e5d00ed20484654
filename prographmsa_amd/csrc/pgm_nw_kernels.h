// pgm_nw_kernels.h — all-pairs integer Gotoh alignment + traceback counts
// (reference src/DistanceFactoryAlign.h:59-127 alignPair).  Exact int32 arithmetic.
//
// One wavefront per sequence pair, pairs pulled from a device-side work queue (atomic counter), so
// the kernel is a persistent grid sized to the chip.  The wavefront sweeps seq2 (rows) in bands of 64 R
// (R = PGM_NW_RPL consecutive rows per lane: only the first takes its upper neighbours from the lane above, the others
// from the lane's own registers, so one set of DPP shifts serves R cells) along seq1 (columns) with the same one-column-per-lane skew as the graph DP: the
// (y-1,x-1), (y-1,x) neighbours arrive by one DPP shift, (y,x-1) is the lane's own register, the symbol
// of seq1 travels down the lanes systolically.  The reference keeps three int32 matrices (12 B/cell) for
// its traceback; the traceback only ever asks "which of diag / X / Y equals W at this cell" (priority
// diag > X > Y, DistanceFactoryAlign.h:100-123), so 2 direction bits per cell are stored instead,
// (2 R bits per lane and step, 16 / R steps per 32-bit word), and the last row of a band is kept in a per-wave scratch row for the next band.
#ifndef PGM_NW_KERNELS_H_
#define PGM_NW_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PGM_NW_RPL 8   // rows of seq2 per lane (1, 2, 4 or 8: 2 R direction bits per lane and step)

struct PgmNwArgs {
    uint32_t dim;            // alphabet size D; scoring matrix is (D+1)^2 column-major
    int32_t gap_open, gap_extend;
    uint32_t npairs;
    uint32_t reduced;        // 1: counts holds (ident, total) per pair — Σ diagonal and Σ all entries of the count matrix, what the
                             // p-distance needs (DistanceFactoryML.h:143-146) — instead of the matrix itself
    const int32_t *score;
    const int8_t *syms;
    const uint32_t *offs;
    const uint32_t *pi, *pj;
    const uint32_t *order;   // pair processing order (longest first)
    int32_t *counts;         // npairs x D x D, zero-initialised (reduced: npairs x 2, written)
    uint32_t *gaps;          // npairs
    uint32_t *queue;         // work counter, zero-initialised
    uint32_t *dirs;          // per wave slot: dir_words_per_slot uint32
    int2 *brow;              // per wave slot: brow_per_slot int2
    size_t dir_words_per_slot;
    size_t brow_per_slot;
    int32_t *status;         // set to PGM_ERR_BACKTRACK if a traceback finds no matching source
};

__device__ __forceinline__ int pgm_dpp_shr1_i(int src, int lane0_value) {   // lane l <- src of lane l-1; lane 0 keeps lane0_value
    return __builtin_amdgcn_update_dpp(lane0_value, src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int pgm_dpp_rol1_i(int src) {                    // lane l <- src of lane l+1 (wave_rol:1)
    return __builtin_amdgcn_update_dpp(src, src, 0x134, 0xf, 0xf, false);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) pgm_nw_kernel(PgmNwArgs A) {
    extern __shared__ int nw_score[];  // (D+1)^2
    const int sd = (int)A.dim + 1;
    // The kernel carries Wg = W + gap_open, Xe = X + gap_extend and Ye = Y + gap_extend instead of W, X, Y (every use of a
    // neighbour's W in a gap term needs W + gap_open, every use of X / Y needs + gap_extend; W + gap_open feeds two
    // terms), so the diagonal term is Wg(y-1,x-1) + (score - gap_open): the table holds score - gap_open.  Integer
    // arithmetic: the values of W, X, Y are the same as with the plain recurrence.
    for (int i = threadIdx.x; i < sd * sd; i += WAVES * 64) nw_score[i] = A.score[i] - A.gap_open;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t slot = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    uint32_t *dirs = A.dirs + slot * A.dir_words_per_slot;
    int2 *brow = A.brow + slot * A.brow_per_slot;
    const int go = A.gap_open, ge = A.gap_extend;
    const int MINF = -10000;

    for (;;) {
        // every lane takes part in the dequeue (lane 0 adds 1, the others 0): a lane-0-only branch in front of
        // the readfirstlane was jump-threaded across the loop back-edge by hipcc (ROCm 7.2), which let lanes 1..63
        // re-enter the loop body without lane 0 and spin forever
        uint32_t q = atomicAdd(A.queue, lane == 0 ? 1u : 0u);
        q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
        if (q >= A.npairs) break;
        // everything derived from the queue index is wave-uniform; say so explicitly so that the band and
        // step loops below stay scalar loops (the loads themselves go through the vector path)
        const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.order[q]);
        const uint32_t i1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.pi[p]);
        const uint32_t i2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.pj[p]);
        const uint32_t o1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.offs[i1]);
        const uint32_t o2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.offs[i2]);
        const int8_t *s1 = A.syms + o1;
        const int8_t *s2 = A.syms + o2;
        const int L1 = __builtin_amdgcn_readfirstlane((int)(A.offs[i1 + 1] - o1));  // columns
        const int L2 = __builtin_amdgcn_readfirstlane((int)(A.offs[i2 + 1] - o2));  // rows
        constexpr int R = PGM_NW_RPL, BR = 64 * R;   // rows per lane, rows per band
        constexpr int SPW = 16 / R;                  // steps per 32-bit direction word (2 bits per cell)
        const int nb = (L2 + BR - 1) / BR;
        const int tsteps = L1 + 63;
        const int twords = (tsteps + SPW - 1) / SPW;

        for (int b = 0; b < nb; ++b) {
            // lane l owns the R consecutive rows y0 + r, y0 = BR b + R l + 1: only row 0 takes its upper neighbours from
            // the lane above (DPP), the others from the lane's own registers, so one set of shifts serves R cells
            const int y0 = BR * b + R * lane + 1;  // 1..L2
            int sy[R], W_left[R], X_left[R];              // W_left = Wg(y, x-1), X_left = Xe(y, x-1)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sy[r] = (y0 + r <= L2) ? (int)s2[y0 + r - 1] : 0;
                W_left[r] = go + (y0 + r - 1) * ge + go;   // W(y,0) = Y(y,0)
                X_left[r] = MINF + ge;                     // X(y,0)
            }
            int W_diag0 = ((y0 == 1) ? 0 : go + (y0 - 2) * ge) + go;   // W(y0-1,0)
            int W_o = MINF, Y_o = MINF, sx_o = 0;               // the last row's outputs (Wg, Ye), consumed by the lane below
            uint32_t word = 0;
            for (int t0 = 0; t0 < tsteps; t0 += 64) {
                // block prefetch for lane 0: seq1 symbols and the row above (columns t0+1 .. t0+64)
                const int xc = t0 + lane + 1;
                int pf_s = 0, pf_w = MINF, pf_y = MINF;
                if (xc <= L1) {
                    pf_s = (int)s1[xc - 1];
                    if (b == 0) { pf_w = go + (xc - 1) * ge + go; pf_y = MINF + ge; }   // (Wg, Ye) of row 0
                    else { const int2 v = brow[xc]; pf_w = v.x; pf_y = v.y; }          // (Wg, Ye) of the band above's last row
                }
                const int tend = min(64, tsteps - t0);
                for (int i = 0; i < tend; ++i) {
                    const int t = t0 + i;
                    const int x = t - lane + 1;  // 1..L1
                    const bool incol = x >= 1 && x <= L1;
                    // lane 0's inputs of step i were prefetched by lane i: the prefetch registers rotate one lane down per
                    // step, so lane 0 always holds the current ones (one DPP instead of readlane + move)
                    const int sx = pgm_dpp_shr1_i(sx_o, pf_s);
                    const int W_up0 = pgm_dpp_shr1_i(W_o, pf_w);
                    const int Y_up0 = pgm_dpp_shr1_i(Y_o, pf_y);
                    pf_s = pgm_dpp_rol1_i(pf_s); pf_w = pgm_dpp_rol1_i(pf_w); pf_y = pgm_dpp_rol1_i(pf_y);
                    const int so = sd * sx;   // (sx, sy are valid symbols in every lane, also the idle ones)
                    int W_up = W_up0, Y_up = Y_up0, W_dg = W_diag0;
                    uint32_t dbits = 0;
                    int Wn[R], Xn[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int dc = W_dg + nw_score[sy[r] + so];   // W(y-1,x-1) + scoring_matrix(s2(y), s1(x))
                        const int Xv = max(X_left[r], W_left[r]);
                        const int Yv = max(Y_up, W_up);
                        const int xy = max(Xv, Yv);
                        const int Wv = max(xy, dc);
                        dbits |= ((dc >= xy) ? 0u : (Xv >= Yv ? 1u : 2u)) << (2 * r);
                        // the next row of this lane: the row above it is this one (same column: just computed; previous
                        // column: W_left before the update)
                        W_dg = W_left[r]; W_up = Wv + go; Y_up = Yv + ge;
                        Wn[r] = W_up; Xn[r] = Xv + ge;
                    }
                    word |= dbits << ((t % SPW) * 2 * R);
                    if ((t % SPW) == SPW - 1 || t == tsteps - 1) {
                        dirs[((size_t)b * twords + (t / SPW)) * 64 + lane] = word;
                        word = 0;
                    }
                    if (incol) {
                        // rows beyond L2 compute on, nobody reads them (an invalid row implies everything below it is invalid)
#pragma unroll
                        for (int r = 0; r < R; ++r) { W_left[r] = Wn[r]; X_left[r] = Xn[r]; }
                        if (lane == 63 && b + 1 < nb) brow[x] = make_int2(W_up, Y_up);
                    }
                    if (x >= 1) W_diag0 = W_up0;  // keep W(y0-1,0) until the lane reaches column 1
                    // a lane's outputs are only consumed by the lane below one step later, which is in a column only if this
                    // one was: idle lanes may pass on whatever they computed
                    W_o = W_up;
                    Y_o = Y_up;
                    sx_o = sx;
                }
            }
        }
        // traceback + counts (DistanceFactoryAlign.h:93-124).  The walk is executed by the whole wavefront in
        // lock-step (every lane reads the same direction word, so control flow stays uniform); only lane 0
        // commits the side effects.
        {
            uint32_t gaps = 0;
            bool open1 = false, open2 = false;
            int32_t *cnt = A.counts + (size_t)p * A.dim * A.dim;
            int ident = 0, total = 0;
            int y = L2, x = L1;
            while (y != 0 && x != 0) {
                const int bb = (y - 1) / BR, rr = (y - 1) % BR, l = rr / R;
                const int t = (x - 1) + l;
                const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)dirs[((size_t)bb * twords + (t / SPW)) * 64 + l]);
                const uint32_t dir = (w >> ((t % SPW) * 2 * R + (rr % R) * 2)) & 3u;
                if (dir == 0) {
                    const int a = __builtin_amdgcn_readfirstlane((int)s1[x - 1]);
                    const int c = __builtin_amdgcn_readfirstlane((int)s2[y - 1]);
                    if (a < (int)A.dim && c < (int)A.dim) {
                        if (A.reduced) { ++total; ident += (a == c) ? 1 : 0; }
                        else if (lane == 0) atomicAdd(&cnt[a + (int)A.dim * c], 1);
                    }
                    open1 = false; open2 = false;
                    --x; --y;
                } else if (dir == 1) {
                    if (!open1) ++gaps;
                    open1 = true; open2 = false;
                    --x;
                } else {
                    if (!open2) ++gaps;
                    open1 = false; open2 = true;
                    --y;
                }
            }
            if (lane == 0) A.gaps[p] = gaps;
            if (lane == 0 && A.reduced) { A.counts[2 * (size_t)p] = ident; A.counts[2 * (size_t)p + 1] = total; }
        }
    }
}

#endif
