// pgm_dist_kernels.h — the tail of the guide-tree stages on the GPU:
//   pgm_mldist_kernel      DistanceFactoryML::computeDistance + computeMLDist (reference src/DistanceFactoryML.h:66-190),
//                          one wavefront per sequence pair: bracketed Newton iteration on the distance d, every step a
//                          20 x 20 P(d) = V diag(exp(sigma d)) V^-1, Q P, Q Q P and two sums over the 400 pair counts
//   pgm_prealigned_kernel  DistanceFactoryPrealigned::computePwDistances' pair counts (src/DistanceFactoryPrealigned.h:34-90):
//                          residue-pair counts and gap openings of every pair of rows of an alignment
// fp64 throughout.  Every sum keeps the host mirror's order (host/distance.cpp, host/model_factory.cpp): matrix
// products accumulate k = 0..n-1 from zero with one multiply and one add per term (no FMA), the two sums over the
// count matrix run over the entries in storage order.  What is NOT bit-identical to the host is exp() (and one log()):
// the device library's results can differ from glibc's in the last bit, so distances agree to ~1e-15 relative, not
// always to the bit (tests/test_gpu_dist.py: 1e-12).
#ifndef PGM_DIST_KERNELS_H_
#define PGM_DIST_KERNELS_H_

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

struct PgmMlArgs {
    uint32_t dim, npairs;
    const double *Q, *V, *Vi, *sigma;   // dim x dim column-major (Q, V, V^-1), dim eigenvalues
    const int32_t *counts;              // npairs x dim x dim
    const uint32_t *gaps;               // npairs
    const double *seqlen;               // npairs: (L1 + L2) / 2
    double dist_max, var_max, var_min, cutoff_dist, min_dist, max_dist, indel_rate;
    int mldist, mldist_gap;
    double *dist, *var;                 // npairs each
};

#define PGM_ML_WAVES 4
#define PGM_ML_DMAX 20

// One wavefront per pair; lane l owns the matrix entries e = l, l + 64, ... (entry e = i + n j, column-major like the host).
__global__ void __launch_bounds__(PGM_ML_WAVES * 64) pgm_mldist_kernel(PgmMlArgs A) {
    constexpr int N = PGM_ML_DMAX, NN = N * N, EPL = (NN + 63) / 64;
    __shared__ double sQ[NN], sV[NN], sVi[NN], sSig[N];
    __shared__ double sE[PGM_ML_WAVES][N], sP[PGM_ML_WAVES][NN], sPP[PGM_ML_WAVES][NN], sT1[PGM_ML_WAVES][NN], sT2[PGM_ML_WAVES][NN];
    const int n = (int)A.dim, nn = n * n;
    for (int i = threadIdx.x; i < nn; i += PGM_ML_WAVES * 64) { sQ[i] = A.Q[i]; sV[i] = A.V[i]; sVi[i] = A.Vi[i]; }
    for (int i = threadIdx.x; i < n; i += PGM_ML_WAVES * 64) sSig[i] = A.sigma[i];
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *E = sE[w], *P = sP[w], *PP = sPP[w], *T1 = sT1[w], *T2 = sT2[w];
    for (uint32_t pair = blockIdx.x * PGM_ML_WAVES + w; pair < A.npairs; pair += gridDim.x * PGM_ML_WAVES) {
        const int32_t *cnt = A.counts + (size_t)pair * nn;
        // computeDistance (DistanceFactoryML.h:137-190): ident / total are integer sums (exact in fp64 whatever the order)
        long long ident = 0, total = 0;
        for (int e = lane; e < nn; e += 64) {
            const int c = cnt[e];
            total += c;
            if (e % n == e / n) ident += c;
        }
        for (int o = 32; o > 0; o >>= 1) { ident += __shfl_xor(ident, o); total += __shfl_xor(total, o); }
        const double identd = (double)ident, totald = (double)total;
        double dist0 = __dsub_rn(1.0, __ddiv_rn(identd, totald));
        double dist, var;
        const double gapsd = (double)A.gaps[pair], seqlen = A.seqlen[pair];
        if (A.mldist || A.mldist_gap) {
            if (total == 0 || dist0 > 0.85) { dist = dist0 = A.dist_max; var = A.var_max; }
            else {
                dist = dist0 = -log(__dsub_rn(__dsub_rn(1.0, dist0), __dmul_rn(__dmul_rn(0.2, dist0), dist0)));
                var = __ddiv_rn(dist, totald);
            }
            if (total > 0 && ident != total) {
                // computeMLDist (DistanceFactoryML.h:66-135)
                const double var0 = var;
                double dist_min = 0.0, dist_maxb = INFINITY, delta = 1.0;
                int iteration = 0;
                while (fabs(delta) > 1e-5) {
                    if (iteration > 20) {
                        if (dist_maxb == INFINITY) { dist = A.dist_max; var = A.var_max; }
                        else { dist = dist0; var = var0; }
                        break;
                    }
                    // getModel(dist): parseDistance clamps (ModelFactory.h:104-127), P = V diag(exp(sigma d)) V^-1
                    double dm = fmax(0.0, dist);
                    if (dist != dist) dm = 5.2;
                    dm = fmax(fmin(dm, A.max_dist), A.min_dist);
                    if (lane < n) E[lane] = exp(__dmul_rn(sSig[lane], dm));
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < EPL; ++u) {
                        const int e = lane + 64 * u;
                        if (e < nn) {
                            const int i = e % n, j = e / n;
                            double acc = 0.0;
                            for (int k = 0; k < n; ++k) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(sV[i + n * k], E[k]), sVi[k + n * j]));
                            P[e] = acc;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < EPL; ++u) {
                        const int e = lane + 64 * u;
                        if (e < nn) {
                            const int i = e % n, j = e / n;
                            double acc = 0.0;
                            for (int k = 0; k < n; ++k) acc = __dadd_rn(acc, __dmul_rn(sQ[i + n * k], P[k + n * j]));
                            PP[e] = acc;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < EPL; ++u) {
                        const int e = lane + 64 * u;
                        if (e < nn) {
                            const int i = e % n, j = e / n;
                            double ppp = 0.0;
                            for (int k = 0; k < n; ++k) ppp = __dadd_rn(ppp, __dmul_rn(sQ[i + n * k], PP[k + n * j]));
                            const double c = (double)cnt[e], p = P[e], pp = PP[e];
                            T1[e] = __ddiv_rn(__dmul_rn(c, pp), p);
                            T2[e] = __ddiv_rn(__dmul_rn(c, __dsub_rn(__dmul_rn(ppp, p), __dmul_rn(pp, pp))), __dmul_rn(p, p));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    // the two sums run over the entries in storage order (every lane adds the same 400 terms: uniform result)
                    double f = 0.0, ff = 0.0;
                    for (int e = 0; e < nn; ++e) { f = __dadd_rn(f, T1[e]); ff = __dadd_rn(ff, T2[e]); }
                    __builtin_amdgcn_wave_barrier();
                    if (A.mldist_gap) {
                        const double grate = __dmul_rn(__dmul_rn(A.indel_rate, seqlen), dist);
                        f = __dadd_rn(f, __ddiv_rn(__dadd_rn(-grate, gapsd), dist));
                        ff = __dadd_rn(ff, -__ddiv_rn(gapsd, __dmul_rn(dist, dist)));
                    }
                    var = __ddiv_rn(-1.0, ff);
                    if (f > 0) dist_min = fmax(dist_min, dist); else dist_maxb = fmin(dist_maxb, dist);
                    double new_dist = __dsub_rn(dist, __ddiv_rn(f, ff));
                    if (!(new_dist < dist_maxb && new_dist > dist_min)) {
                        const double upper = (dist_maxb == INFINITY) ? __dmul_rn(dist, 3.0) : dist_maxb;
                        new_dist = __ddiv_rn(__dadd_rn(upper, dist_min), 2.0);
                    }
                    delta = __dsub_rn(1.0, __ddiv_rn(new_dist, dist));
                    dist = new_dist;
                    ++iteration;
                }
            }
        } else {
            if (total == 0) { dist = dist0 = 1.0; var = A.var_max; }
            else { dist = dist0; var = __ddiv_rn(dist0, totald); }
        }
        if (!(dist < A.dist_max)) { dist = A.dist_max; var = A.var_max; }
        if (dist > A.cutoff_dist) dist = A.cutoff_dist;
        if (var < A.var_min) var = A.var_min;
        if (!(var < A.var_max)) var = A.var_max;
        if (lane == 0) { A.dist[pair] = dist; A.var[pair] = var; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair counts of an alignment.  rows: nrows x ncols int8, row-major: residue value() 0..dim-1, -1 = gap, -2 = a residue
// without a value (unknown: it is a residue for the gap bookkeeping, but never counted).  Only values < 20 are counted,
// for every alphabet (the reference's literal 20, DistanceFactoryPrealigned.h).  One wavefront per pair: lane l scans the
// columns [l * chunk, (l + 1) * chunk); the gap openings of the state machine (a run in which exactly one row has a
// residue opens one gap; columns where both rows have a gap are skipped) are counted per chunk with "no run open" at the
// chunk's start and corrected at the chunk boundaries afterwards.
struct PgmPaArgs {
    uint32_t dim, nrows, ncols, npairs;
    const int8_t *rows;
    const uint32_t *pi, *pj;
    int32_t *counts;     // npairs x dim x dim, zero-initialised
    uint32_t *gaps;      // npairs
};

__global__ void __launch_bounds__(256) pgm_prealigned_kernel(PgmPaArgs A) {
    __shared__ int cnt[4][400];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t D = A.dim, L = A.ncols;
    const uint32_t chunk = (L + 63u) / 64u;
    for (uint32_t pair = blockIdx.x * 4 + w; pair < A.npairs; pair += gridDim.x * 4) {
        for (int i = lane; i < 400; i += 64) cnt[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
        const int8_t *r1 = A.rows + (size_t)A.pi[pair] * L, *r2 = A.rows + (size_t)A.pj[pair] * L;
        // type of a column: 0 both residues, 1 only row 1 has a residue, 2 only row 2, 3 both gaps (skipped)
        int first = 3, last = 3;
        uint32_t g = 0;
        const uint32_t k0 = (uint32_t)lane * chunk, k1 = min(L, k0 + chunk);
        for (uint32_t k = k0; k < k1; ++k) {
            const int c1 = r1[k], c2 = r2[k];
            const bool g1 = c1 == -1, g2 = c2 == -1;
            const int ty = (!g1 && !g2) ? 0 : ((g1 && g2) ? 3 : (!g1 ? 1 : 2));
            if (ty == 3) continue;
            if (ty == 0) { if (c1 >= 0 && c1 < 20 && c2 >= 0 && c2 < 20) atomicAdd(&cnt[w][c1 + 20 * c2], 1); }
            else if (ty != last) ++g;            // a run opens (the previous non-skipped column was of another type)
            if (first == 3) first = ty;
            last = ty;
        }
        // chunk boundaries: a chunk whose first non-skipped column continues the run the previous non-empty chunk ended in
        // has counted one opening too many
        int prev_last = 3;
        uint32_t total = 0;
        for (int l = 0; l < 64; ++l) {
            const int f = __builtin_amdgcn_readlane(first, l), la = __builtin_amdgcn_readlane(last, l);
            total += (uint32_t)__builtin_amdgcn_readlane((int)g, l);
            if (f != 3) {
                if (f != 0 && f == prev_last) --total;
                prev_last = la;
            }
        }
        __builtin_amdgcn_wave_barrier();
        int32_t *out = A.counts + (size_t)pair * D * D;
        for (int i = lane; i < 400; i += 64) {
            const int c1 = i % 20, c2 = i / 20;
            if ((uint32_t)c1 < D && (uint32_t)c2 < D) out[c1 + D * c2] = cnt[w][i];
        }
        if (lane == 0) A.gaps[pair] = total;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// k-mer cosine matrix of DistanceFactoryAngle (reference src/DistanceFactoryAngle.h:55-115): the one dense
// contraction of the program.  counts: nseq x ncols int32 (row i = the k-mer counts of sequence i, ncols = DIM^K: 400 for amino
// acids, 3721 for codons).  The reference evaluates diag(1/|c_i|) * C * C^T * diag(1/|c_j|) left to right in double:
//     out(i, j) = (sum_k (c_ik * inv_i) * c_jk) * inv_j,     inv_i = 1 / sqrt(sum_k c_ik^2)
// k ascending, one fp64 multiply and one add per term (no FMA), in depth blocks of `kc` terms whose sums are added to the result
// one after the other: Eigen's GEMM (kc = L1d / 128 terms, 384 on the golden files' host; include/pgm_hip.h).  The norms are
// exact (integer sums of squares, correctly rounded sqrt and reciprocal).  One thread per (i, j), 16 x 16 outputs per workgroup, the two 16-row slabs staged through LDS in
// chunks of 64 columns (the i-slab already scaled).
#define PGM_KC_TILE 16
#define PGM_KC_CHUNK 64
__global__ void __launch_bounds__(PGM_KC_TILE * PGM_KC_TILE) pgm_kmer_cosine_kernel(uint32_t nseq, uint32_t ncols, const int32_t *__restrict__ counts,
                                                                                 const double *__restrict__ inv_norm, double *__restrict__ out, uint32_t kc) {
    __shared__ double sa[PGM_KC_TILE][PGM_KC_CHUNK + 1], sb[PGM_KC_TILE][PGM_KC_CHUNK + 1];
    const uint32_t tx = threadIdx.x % PGM_KC_TILE, ty = threadIdx.x / PGM_KC_TILE;
    const uint32_t i = blockIdx.y * PGM_KC_TILE + ty, j = blockIdx.x * PGM_KC_TILE + tx;
    double acc = 0.0, res = 0.0;
    uint32_t left = kc;   // terms left in the depth block
    for (uint32_t k0 = 0; k0 < ncols; k0 += PGM_KC_CHUNK) {
        for (uint32_t e = threadIdx.x; e < PGM_KC_TILE * PGM_KC_CHUNK; e += PGM_KC_TILE * PGM_KC_TILE) {
            const uint32_t r = e / PGM_KC_CHUNK, c = e % PGM_KC_CHUNK, k = k0 + c;
            const uint32_t ri = blockIdx.y * PGM_KC_TILE + r, rj = blockIdx.x * PGM_KC_TILE + r;
            sa[r][c] = (ri < nseq && k < ncols) ? __dmul_rn((double)counts[(size_t)ri * ncols + k], inv_norm[ri]) : 0.0;
            sb[r][c] = (rj < nseq && k < ncols) ? (double)counts[(size_t)rj * ncols + k] : 0.0;
        }
        __syncthreads();
        const uint32_t kn = min((uint32_t)PGM_KC_CHUNK, ncols - k0);
        for (uint32_t c = 0; c < kn; ++c) {
            acc = __dadd_rn(acc, __dmul_rn(sa[ty][c], sb[tx][c]));
            if (--left == 0u) { res = __dadd_rn(res, acc); acc = 0.0; left = kc; }
        }
        __syncthreads();
    }
    if (left != kc) res = __dadd_rn(res, acc);   // the last, shorter block
    if (i < nseq && j < nseq) out[(size_t)i + (size_t)nseq * j] = __dmul_rn(res, inv_norm[j]);
}
__global__ void pgm_kmer_norm_kernel(uint32_t nseq, uint32_t ncols, const int32_t *__restrict__ counts, double *__restrict__ inv_norm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq) return;
    unsigned long long ss = 0;
    for (uint32_t k = 0; k < ncols; ++k) { const long long c = counts[(size_t)i * ncols + k]; ss += (unsigned long long)(c * c); }
    inv_norm[i] = __ddiv_rn(1.0, __dsqrt_rn((double)ss));
}

#endif
