// pgm_merge_kernels.h — the numeric part of mergeGraphs (reference src/GraphAlign.h:569-620): the
// profile of every node of the merged graph, p = normalized(P1 g1(:,k1) .* P2 g2(:,k2)) (or one factor alone for an
// unmatched / skipped node), fp64.
// Bit-exact contract (tests/test_gpu_merge.py, and every FASTA fixture of tests/test_gpu_e2e.py runs through it):
//   * P v in the association of Eigen 3.0-3.2's column-major gemv, which the reference binary shows: four columns at a time,
//     out_i += (c0_i v0 + c1_i v1) + (c2_i v2 + c3_i v3), leftover columns one by one; one multiply / one add per term
//   * the L2 norm is the sequential sum of squares, sqrt, and a multiplication by the reciprocal (Eigen's scalar quotient)
//   * a zero vector stays zero (START / END columns)
#ifndef PGM_MERGE_KERNELS_H_
#define PGM_MERGE_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

struct PgmMergeJobDev {
    uint32_t dim, nnodes;
    const double *sites1, *sites2, *P1, *P2;
    const uint32_t *k1, *k2;
    const uint8_t *g2p1;
    double *profiles;
    uint32_t first_block;      // blocks [first_block, first_block + ceil(nnodes / nodes per block)) work on this job
};

// block = 256 threads = NB nodes x DP row threads (DP = 32 for 20 states: NB = 8; DP = 64 for 61 states: NB = 4)
template <int DP>
__global__ void __launch_bounds__(256) pgm_merge_profiles_kernel(const PgmMergeJobDev *__restrict__ jobs, uint32_t njobs) {
    constexpr int NB = 256 / DP;
    extern __shared__ double mlds[];   // P1 (D*D), P2 (D*D), v1 [NB][D], v2 [NB][D], p [NB][D], inv [NB]
    // which job does this block belong to (jobs are few: linear scan by every thread, uniform)
    uint32_t j = 0;
    while (j + 1 < njobs && jobs[j + 1].first_block <= blockIdx.x) ++j;
    const PgmMergeJobDev J = jobs[j];
    const int D = (int)J.dim;
    double *P1 = mlds, *P2 = P1 + D * D, *v1 = P2 + D * D, *v2 = v1 + NB * D, *pp = v2 + NB * D, *inv = pp + NB * D;
    for (int i = threadIdx.x; i < D * D; i += 256) { P1[i] = J.P1[i]; P2[i] = J.P2[i]; }
    const int nl = threadIdx.x / DP, row = threadIdx.x % DP;
    const uint32_t node = (blockIdx.x - J.first_block) * NB + (uint32_t)nl;
    const bool valid = node < J.nnodes;
    const uint32_t a = valid ? J.k1[node] : 0xFFFFFFFFu, b = valid ? J.k2[node] : 0xFFFFFFFFu;
    const bool has1 = a != 0xFFFFFFFFu, has2 = b != 0xFFFFFFFFu;
    const bool b_p1 = valid && J.g2p1[node] != 0;
    if (row < D) {
        v1[nl * D + row] = has1 ? J.sites1[(size_t)a * D + row] : 0.0;
        v2[nl * D + row] = has2 ? J.sites2[(size_t)b * D + row] : 0.0;
    }
    __syncthreads();
    auto gemv_row = [&](const double *P, const double *v) {
        double out = 0.0;
        int c = 0;
        for (; c + 4 <= D; c += 4) {
            const double t01 = __dadd_rn(__dmul_rn(P[row + D * c], v[c]), __dmul_rn(P[row + D * (c + 1)], v[c + 1]));
            const double t23 = __dadd_rn(__dmul_rn(P[row + D * (c + 2)], v[c + 2]), __dmul_rn(P[row + D * (c + 3)], v[c + 3]));
            out = __dadd_rn(out, __dadd_rn(t01, t23));
        }
        for (; c < D; ++c) out = __dadd_rn(out, __dmul_rn(P[row + D * c], v[c]));
        return out;
    };
    double p = 0.0;
    if (row < D && valid) {
        if (has1) p = gemv_row(P1, v1 + nl * D);
        if (has2) {
            const double q = gemv_row(b_p1 ? P1 : P2, v2 + nl * D);
            p = has1 ? __dmul_rn(p, q) : q;
        }
        pp[nl * D + row] = p;
    }
    __syncthreads();
    if (row == 0 && valid) {
        double s = 0.0;
        for (int i = 0; i < D; ++i) s = __dadd_rn(s, __dmul_rn(pp[nl * D + i], pp[nl * D + i]));
        const double nrm = __dsqrt_rn(s);
        inv[nl] = nrm == 0.0 ? 0.0 : __ddiv_rn(1.0, nrm);
    }
    __syncthreads();
    if (row < D && valid) {
        const double iv = inv[nl];
        J.profiles[(size_t)node * D + row] = iv == 0.0 ? p : __dmul_rn(p, iv);
    }
}

// Leaf graphs of a progressive pass built where they are used (SequenceGraph.h:101-109): column c of sequence s's dim x (L + 2)
// profile matrix is zero for START / END, one-hot for a residue with a value, uniform 1 / dim for one without (sym < 0).
__global__ void __launch_bounds__(256) pgm_onehot_kernel(uint32_t dim, uint32_t nseq, const int8_t *__restrict__ syms, const uint32_t *__restrict__ offs,
                                                       const uint64_t *__restrict__ col0, double *__restrict__ out, uint64_t ncols) {
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (c >= ncols) return;
    // which sequence: col0[s] <= c < col0[s + 1] (binary search, nseq is small)
    uint32_t lo = 0, hi = nseq;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (col0[mid] <= c) lo = mid; else hi = mid; }
    const uint32_t s = lo, L = offs[s + 1] - offs[s];
    const uint64_t k = c - col0[s];
    double *dst = out + c * dim;
    const bool residue = k >= 1 && k <= L;   // (else START / END: a zero column — not a symbol value: every negative int8 means "no value")
    const int sym = residue ? (int)syms[offs[s] + (uint32_t)(k - 1)] : 0;
    const double uni = 1.0 / (double)dim;
    for (uint32_t r = 0; r < dim; ++r) dst[r] = !residue ? 0.0 : (sym < 0 ? uni : ((uint32_t)sym == r ? 1.0 : 0.0));
}

#endif
