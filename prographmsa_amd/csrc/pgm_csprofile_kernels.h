// pgm_csprofile_kernels.h — context-specific leaf profiles (reference src/CSProfile.cpp:175-225
// createProfile; Biegert & Soding 2009).  fp64 like the reference.
//
// One thread per (sequence, residue).  The sum over the K context profiles is kept sequential in k
// (the reference accumulates `profile.row(i+1) += profiles[k].row(center) * exp(pk)` for k = 0..K-1),
// multiply-then-add without contraction; the 13-column window score pk is accumulated in the
// reference's order (prior first, then window offsets ascending).  The only deviation from the CPU
// result is the device exp() (<= 1 ulp), i.e. ~1e-16 relative on the profile entries.
#ifndef PGM_CSPROFILE_KERNELS_H_
#define PGM_CSPROFILE_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

struct PgmCsArgs {
    uint32_t K, ncols, nseq;
    const double *lprofiles;  // [K][ncols][21]
    const double *centre;     // [K][20]
    const double *priors;     // [K]
    const int8_t *syms;       // 0..19, 20 = invalid
    const uint32_t *offs;     // nseq+1
    const uint32_t *pos_seq;  // sequence index of every residue (flattened over all sequences)
    const double *tau;        // nseq
    const double *pi;         // 20
    const double *p_uniform;  // nseq x 20
    double *out;              // concatenated 20 x (L+2) column-major profiles
    const uint64_t *out_offs; // nseq+1
    uint32_t total;           // total residues
};

__global__ void __launch_bounds__(256) pgm_csprofile_kernel(PgmCsArgs A) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= A.total) return;
    const uint32_t s = A.pos_seq[gid];
    const uint32_t base = A.offs[s];
    const int L = (int)(A.offs[s + 1] - base);
    const int i = (int)(gid - base);
    const int8_t *seq = A.syms + base;
    const int center = (int)A.ncols / 2;
    // window symbols (invalid positions marked -1)
    int win[32];
    for (int j = -center; j <= center; ++j) {
        const int pp = i + j;
        win[j + center] = (pp >= 0 && pp < L) ? (int)seq[pp] : -1;
    }
    double acc[20];
#pragma unroll
    for (int a = 0; a < 20; ++a) acc[a] = 0.0;
    for (uint32_t k = 0; k < A.K; ++k) {
        const double *lp = A.lprofiles + (size_t)k * A.ncols * 21;
        double pk = A.priors[k];
        for (int c = 0; c < (int)A.ncols; ++c)
            if (win[c] >= 0) pk = __dadd_rn(pk, lp[c * 21 + win[c]]);
        const double e = exp(pk);
        const double *ce = A.centre + (size_t)k * 20;
#pragma unroll
        for (int a = 0; a < 20; ++a) acc[a] = __dadd_rn(acc[a], __dmul_rn(ce[a], e));
    }
    // per-row finishing (CSProfile.cpp:205-222)
    const int c = seq[i];
    double sum = 0.0;
#pragma unroll
    for (int a = 0; a < 20; ++a) sum = __dadd_rn(sum, acc[a]);
    double *dst = A.out + A.out_offs[s] + (size_t)20 * (i + 1);
    if (sum <= 0) {
#pragma unroll
        for (int a = 0; a < 20; ++a) dst[a] = A.p_uniform[(size_t)s * 20 + a];
    } else if (c < 0 || c > 19) {
        const double f = __ddiv_rn(1.0, sum);
#pragma unroll
        for (int a = 0; a < 20; ++a) dst[a] = __dmul_rn(__dmul_rn(acc[a], f), __dmul_rn(1.0 / 20.0, __ddiv_rn(1.0, A.pi[a])));
    } else {
        const double tau = A.tau[s];
        const double f = __ddiv_rn(tau, sum);
#pragma unroll
        for (int a = 0; a < 20; ++a) acc[a] = __dmul_rn(acc[a], f);
        double vc = __dadd_rn(acc[0], 0.0);
#pragma unroll
        for (int a = 0; a < 20; ++a) if (a == c) vc = acc[a];
        vc = __dadd_rn(vc, __dsub_rn(1.0, tau));
        if (vc <= 0.0) vc = 1e-3;
#pragma unroll
        for (int a = 0; a < 20; ++a) {
            const double v = (a == c) ? vc : acc[a];
            dst[a] = __dmul_rn(v, __dmul_rn(1.0 / 20.0, __ddiv_rn(1.0, A.pi[a])));
        }
    }
    // START / END columns are zero (CSProfile.cpp:176)
    if (i == 0) {
        double *z = A.out + A.out_offs[s];
        for (int a = 0; a < 20; ++a) { z[a] = 0.0; z[(size_t)20 * (L + 1) + a] = 0.0; }
    }
}

#endif
