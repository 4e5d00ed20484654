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
    const double *lprofiles;  // [K] records of (ncols + 1) * 22 doubles: ncols x 22 window entries, 20 centre values, prior, pad
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

// The library (8.7 MB for K = 4000) is streamed through LDS in chunks of PGM_CS_KC profiles: every thread of the
// workgroup needs every profile, and its 13 window look-ups per profile are gathers (one of 21 entries per column) that
// LDS serves at a fraction of the L2 cost.  pgm_csprofile_load stores it as one record of (ncols + 1) * 22 doubles per
// profile — ncols window columns of 22 entries (the 22nd holds 0.0 for positions outside the sequence, which the
// reference skips: pk + 0.0 == pk, so the sum is the reference's without a branch per column), then the 20 centre
// values, the prior and a pad — so that staging a chunk is a straight 16-byte copy.
#define PGM_CS_KC 16
#define PGM_CS_U 4    // profiles evaluated together (independent chains); divides PGM_CS_KC
__global__ void __launch_bounds__(256) pgm_csprofile_kernel(PgmCsArgs A) {
    extern __shared__ double cs_lds[];   // [KC] records
    const uint32_t tab = A.ncols * 22u, rec = tab + 22u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = gid < A.total;
    const uint32_t s = active ? A.pos_seq[gid] : 0u;
    const uint32_t base = A.offs[s];
    const int L = (int)(A.offs[s + 1] - base);
    const int i = active ? (int)(gid - base) : 0;
    const int8_t *seq = A.syms + base;
    const int center = (int)A.ncols / 2;
    // window symbols; 20 = invalid residue (the library's own 21st entry), 21 = outside the sequence (0.0)
    uint32_t woff[32];
    for (int j = -center; j <= center; ++j) {
        const int pp = i + j;
        int w = (active && pp >= 0 && pp < L) ? (int)seq[pp] : 21;
        if (w < 0 || w > 21) w = 20;
        woff[j + center] = (uint32_t)((j + center) * 22 + w);
    }
    double acc[20];
#pragma unroll
    for (int a = 0; a < 20; ++a) acc[a] = 0.0;
    for (uint32_t k0 = 0; k0 < A.K; k0 += PGM_CS_KC) {
        const uint32_t kc = min((uint32_t)PGM_CS_KC, A.K - k0);
        __syncthreads();
        {
            typedef double pgm_d2 __attribute__((ext_vector_type(2)));
            const pgm_d2 *src = (const pgm_d2 *)(A.lprofiles + (size_t)k0 * rec);
            pgm_d2 *dst = (pgm_d2 *)cs_lds;
            for (uint32_t t = threadIdx.x; t < kc * rec / 2u; t += blockDim.x) dst[t] = src[t];
        }
        __syncthreads();
        // PGM_CS_U profiles at a time: their window sums and exponentials are independent chains (the accumulation below
        // stays in k order), which is what keeps a wavefront's issue slots filled
        uint32_t kk = 0;
        for (; kk + PGM_CS_U <= kc; kk += PGM_CS_U) {
            const double *lp0 = cs_lds + kk * rec;
            // the prior and the 20 centre values are the same for every lane: read them through the scalar cache (constant
            // address space) instead of LDS broadcasts
            const __attribute__((address_space(4))) double *cg =
                (const __attribute__((address_space(4))) double *)(uintptr_t)(A.lprofiles + (size_t)(k0 + kk) * rec + tab);
            double pk[PGM_CS_U], e[PGM_CS_U];
#pragma unroll
            for (int u = 0; u < PGM_CS_U; ++u) pk[u] = cg[u * rec + 20u];
            for (int c = 0; c < (int)A.ncols; ++c) {
                const uint32_t w = woff[c];
#pragma unroll
                for (int u = 0; u < PGM_CS_U; ++u) pk[u] = __dadd_rn(pk[u], lp0[u * rec + w]);
            }
#pragma unroll
            for (int u = 0; u < PGM_CS_U; ++u) e[u] = exp(pk[u]);
#pragma unroll
            for (int a = 0; a < 20; ++a) {
                double v = acc[a];
#pragma unroll
                for (int u = 0; u < PGM_CS_U; ++u) v = __dadd_rn(v, __dmul_rn(cg[u * rec + a], e[u]));
                acc[a] = v;
            }
        }
        for (; kk < kc; ++kk) {
            const double *lp = cs_lds + kk * rec;
            double pk = lp[tab + 20u];
            for (int c = 0; c < (int)A.ncols; ++c) pk = __dadd_rn(pk, lp[woff[c]]);
            const double e = exp(pk);
            const double *ce = lp + tab;
#pragma unroll
            for (int a = 0; a < 20; ++a) acc[a] = __dadd_rn(acc[a], __dmul_rn(ce[a], e));
        }
    }
    if (!active) return;
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
