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
#include <type_traits>
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
// Prep kernel: float casts, T = M^T g2, per-node denominators.  grid = (njobs, 2, slices), block = 256; the nodes of a
// graph are dealt to the slices (one thread per node), so the largest job's graphs do not serialise on one workgroup.
//   side 0: g1f[y][k] = float(sites1(k,y));  a1[y] = sum_k g1f[y][k] * pi_f[k]
//   side 1: t2[x][k]  = sum_j M_f(j,k) * g2f(j,x);  b2[x] = sum_k pi_f[k] g2f(k,x)
// A workgroup takes NODES consecutive nodes: their profile columns (doubles, D per node) are one contiguous run, read
// coalesced and converted into LDS; every thread then works on its node from LDS (padded stride: conflict free), and the
// results go back through LDS so that the stores are coalesced too.  The sums keep the reference's sequential order (one
// multiply and one add per term, no FMA).  DMAX = 20 for amino acids (NODES = 256), 64 for codons (NODES = 64).
// The kernel also zeroes the batch's sync block (progress words, tickets, abort flag) for the fill stage two kernels later: a
// hipMemsetAsync between the emission kernel and the sweeps cost 55 us of fill kernel plus its dependency per step.
template <int DMAX, int NODES>
__global__ void __launch_bounds__(NODES) pgm_prep_kernel(const PgmJob *__restrict__ jobs, int *__restrict__ sync, uint32_t sync_ints) {
    extern __shared__ float prep_lds[];  // Mf (dim*dim), pif (dim), nodes [NODES][dim+1]
    {
        const uint32_t nthreads = gridDim.x * gridDim.y * gridDim.z * NODES;
        const uint32_t gid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NODES + threadIdx.x;
        for (uint32_t i = gid; i < sync_ints; i += nthreads) sync[i] = 0;
    }
    const PgmJob &J = jobs[blockIdx.x];
    const bool first = blockIdx.y == 0;
    const uint32_t n = first ? J.n1 : J.n2;
    const uint32_t v0 = blockIdx.z * NODES;
    if (v0 >= n) return;   // no nodes for this slice
    if (J.tabhdr && J.tabhdr[0] == 0) {
        // two sequence graphs (PgmJob::cls1): the scores come from a table built from ONE node per class, the walk reads decision bits
        // and the END node's edge record — nothing of the other nodes is needed.  This slice only works if it holds one of those nodes.
        const uint32_t nc = J.dim + 2u;
        bool any = n - 1u >= v0 && n - 1u < v0 + (uint32_t)NODES;   // (pgm_traceback_chain: tb1[n1 - 1], tb2[n2 - 1])
        for (uint32_t c = 0; c < nc; ++c) {
            const uint32_t r = (uint32_t)J.tabhdr[1u + (first ? 0u : nc) + c];
            any = any || (r != 0u && r - 1u >= v0 && r - 1u < v0 + (uint32_t)NODES);
        }
        if (!any) return;
    }
    const uint32_t D = J.dim, DP = J.dp, ST = D + 1;
    float *Mf = prep_lds;
    float *pif = prep_lds + D * D;
    float *nl = pif + D;
    const uint32_t nn = min((uint32_t)NODES, n - v0);
    {   // predecessor records of this slice's nodes for the traceback (PgmTbNode)
        const int32_t *pp = first ? J.pp1 : J.pp2;
        const uint32_t *pc = first ? J.pc1 : J.pc2, *pu = first ? J.pu1 : J.pu2;
        const float *pv = first ? J.pv1 : J.pv2;
        PgmTbNode *rec = (first ? J.tb1 : J.tb2) + v0;
        for (uint32_t i = threadIdx.x; i < nn * PGM_TB_PK; i += NODES) {
            const uint32_t node = i / PGM_TB_PK, k = i % PGM_TB_PK;
            const int32_t eb = pp[v0 + node];
            const uint32_t cnt = (uint32_t)(pp[v0 + node + 1] - eb);
            const int32_t e = cnt == 0u ? 0 : eb + (int32_t)min(k, cnt - 1u);
            if (k == 0) rec[node].cnt = cnt;
            rec[node].c[k] = pc[e]; rec[node].v[k] = pv[e]; rec[node].u[k] = pu[e];
        }
    }
    if (!first) for (uint32_t i = threadIdx.x; i < D * D; i += NODES) Mf[i] = (float)J.M[i];
    for (uint32_t i = threadIdx.x; i < D; i += NODES) pif[i] = (float)J.pi[i];
    const double *sites = first ? J.sites1 : J.sites2;
    const uint32_t *smap = first ? J.smap1 : J.smap2;   // (profiles left in HBM by the merge of the level below: gathered through the cleaned graph's node map)
    float *dst = (first ? J.g1f : J.t2) + (size_t)DP * v0;
    for (uint32_t i = threadIdx.x; i < nn * D; i += NODES) {
        const uint32_t node = v0 + i / D;
        const float g = (float)sites[(size_t)D * (smap ? smap[node] : node) + i % D];
        nl[(i / D) * ST + (i % D)] = g;
        if (first && DP == D) dst[i] = g;                     // g1f is the converted column itself
    }
    __syncthreads();
    const uint32_t v = threadIdx.x;
    float g[DMAX];
    float acc = 0.0f;
    if (v < nn) {
#pragma unroll
        for (int k = 0; k < DMAX; ++k) g[k] = ((uint32_t)k < D) ? nl[v * ST + k] : 0.0f;
        // denominators g1^T pi / pi^T g2: the reference's matrix-vector kernel (Eigen SSE) sums four lanes over the packets
        // of four, adds them as (l0 + l2) + (l1 + l3), then the scalar tail (61 states: element 60) -- not sequentially
        float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, l3 = 0.0f;
        const uint32_t aligned = D / 4u * 4u;
#pragma unroll
        for (int k = 0; k + 3 < DMAX; k += 4) {
            if ((uint32_t)k < aligned) {
                l0 = __fadd_rn(l0, __fmul_rn(g[k], pif[k]));
                l1 = __fadd_rn(l1, __fmul_rn(g[k + 1], pif[k + 1]));
                l2 = __fadd_rn(l2, __fmul_rn(g[k + 2], pif[k + 2]));
                l3 = __fadd_rn(l3, __fmul_rn(g[k + 3], pif[k + 3]));
            }
        }
        acc = __fadd_rn(0.0f, __fadd_rn(__fadd_rn(l0, l2), __fadd_rn(l1, l3)));
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
            if ((uint32_t)k >= aligned && (uint32_t)k < D) acc = __fadd_rn(acc, __fmul_rn(g[k], pif[k]));
        (first ? J.a1 : J.b2)[v0 + v] = acc;
    }
    if (first) {
        if (DP != D && v < nn) {
#pragma unroll
            for (int k = 0; k < DMAX; ++k) if ((uint32_t)k < DP) dst[(size_t)DP * v + k] = g[k];
        }
        return;
    }
    __syncthreads();   // every thread has its column in registers: the LDS rows are reused for the results
    if (v < nn) {
        for (uint32_t k = 0; k < D; ++k) {
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < DMAX; ++j)
                if ((uint32_t)j < D) t = __fadd_rn(t, __fmul_rn(Mf[j + D * k], g[j]));
            nl[v * ST + k] = t;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nn * DP; i += NODES) {
        const uint32_t node = i / DP, k = i % DP;
        dst[i] = k < D ? nl[node * ST + k] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Classes of the nodes of the lean jobs' graphs (PgmJob::cls1 / cls2 / tabhdr), once per batch (the inputs of a batch do not change
// between its launches): grid = (njobs, 2, slices), one thread per node.  The class is decided on the float value of the column, as
// pgm_prep_kernel casts it.
#define PGM_TAB_HDR 128   /* ints per job: [0] some node without a class, [1 ..] a node + 1 per class of graph 1, then of graph 2 */
__global__ void __launch_bounds__(256) pgm_classify_kernel(const PgmJob *__restrict__ jobs) {
    const PgmJob &J = jobs[blockIdx.x];
    if (!J.tabhdr) return;
    const bool first = blockIdx.y == 0;
    const uint32_t n = first ? J.n1 : J.n2, D = J.dim;
    const double *sites = first ? J.sites1 : J.sites2;
    const uint32_t *smap = first ? J.smap1 : J.smap2;
    uint8_t *cls = (uint8_t *)(first ? J.cls1 : J.cls2);
    const float uni = (float)(1.0 / (double)D);
    for (uint32_t v = blockIdx.z * blockDim.x + threadIdx.x; v < n; v += gridDim.z * blockDim.x) {
        const double *col = sites + (size_t)D * (smap ? smap[v] : v);
        uint32_t nz = 0, sym = 0;
        bool same = true;
        float vs = 0.f;
        const float g0 = (float)col[0];
        for (uint32_t k = 0; k < D; ++k) {
            const float g = (float)col[k];
            if (g != 0.0f) { ++nz; sym = k; vs = g; }
            same = same && g == g0;
        }
        uint32_t c = 255u;
        if (nz == 0u) c = D + 1u;
        else if (nz == 1u && vs == 1.0f) c = sym;
        else if (same && g0 == uni) c = D;
        cls[v] = (uint8_t)c;
        if (c == 255u) atomicOr(J.tabhdr, 1);
        else atomicCAS(J.tabhdr + 1 + (first ? 0u : D + 2u) + c, 0, (int)v + 1);
    }
}

// ---------------------------------------------------------------------------------------------
// cell addressing (see PgmJob): cell = float4 {M, X, W, Y}
// (jobs of the fill kernel: one row per lane; the lean kernel keeps no cells unless asked to and never reads them back)
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
// Emission kernel (GraphAlign.h:145-163 precomputeScores, fused with ls_log_add): S for every cell, written
// in the skewed (band, step, lane) order in which the fill kernel consumes it.  No dependencies between
// cells, so this part of the reference's per-cell work runs at full occupancy, off the DP's critical path.
// grid = (ceil(nblk / 4), ceil(nb / (4 RB)), njobs); block = 256 threads: thread T owns row T % 64 of the RB bands
// (4 blockIdx.y + T / 64) RB + r and computes their cells of PGM_EM_TB = 4 step blocks (columns t - l, l = T % 64).
// The 95 columns of T = M^T g2 the workgroup touches are staged in LDS as overlapping column PAIRS
// {T[c][k], T[c+1][k]}: two consecutive cells of a row are then one packed multiply and one packed add per k
// (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, the same sequential mul-then-add order as the scalar code).
// Global-address-space accesses: pointers taken from a PgmJob in memory are generic, and generic (flat) loads / stores
// tie vmcnt and lgkmcnt together (every LDS wait then also waits for them); these compile to global_load / global_store.
#define PGM_GLOBAL __attribute__((address_space(1)))
typedef float pgm_v4f __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T pgm_gld(const T *p) { return *(const PGM_GLOBAL T *)(uintptr_t)p; }   // scalar global load
#define PGM_EM_TB 4   // step blocks per emission workgroup
template <int DP, int RB>   // RB: rows (bands) per thread — one LDS read of a column pair serves 2 RB cells
__global__ void __launch_bounds__(4 * PGM_ROWS) pgm_emission_skew_kernel(const PgmJob *__restrict__ jobs) {
    typedef float pgm_v2f __attribute__((ext_vector_type(2)));
    constexpr int NT = DP / 4;
    constexpr int COLS = 64 + PGM_EM_TB * PGM_BLOCK - PGM_HALO - 1;   // columns [t0 - 63, t0 + PGM_EM_TB * PGM_BLOCK - 1 - PGM_HALO]
    // (row stride DP + 1 pairs: lane l reads row ci - l, and with a stride of DP pairs — 128 dwords for 64 states — every lane of
    // a wavefront would hit the same LDS banks)
    constexpr int STR = DP + 1;
    __shared__ pgm_v2f tp[(COLS + 1) * STR];     // tp[c * STR + k] = {T[c][k], T[c+1][k]} (row COLS is scratch)
    __shared__ float bq[COLS];
    const PgmJob &J = jobs[blockIdx.z];
    const uint32_t tb0 = blockIdx.x * PGM_EM_TB;
    // R rows per lane (PgmJob::rshift): the kernel's "bands" are the nb R virtual bands v = b R + r (lane l of virtual band v
    // owns row 64 R b + R l + r, at step t column t - l as ever); scores land at [(b nblk + tb) R + r][lane][8]
    const uint32_t rsh = J.rshift, nvb = J.nb << rsh;
    if (tb0 >= J.nblk || 4u * RB * blockIdx.y >= nvb) return;
    if (J.tabhdr && pgm_gld(J.tabhdr) == 0) return;   // two sequence graphs: pgm_lean_kernel looks the scores up (PgmJob::cls1)
    auto row_of = [&](uint32_t v, uint32_t lane_) { return ((((v >> rsh) << 6) + lane_) << rsh) | (v & ((1u << rsh) - 1u)); };
    auto sblock_of = [&](uint32_t v, uint32_t tb_) { return ((((size_t)(v >> rsh) * J.nblk + tb_) << rsh) | (v & ((1u << rsh) - 1u))); };
    const uint32_t t0 = tb0 * PGM_BLOCK;
    const int cbase = (int)t0 - 63;
    if (DP == 20 && RB == 1) {
        // Rows of a sequence graph (a leaf of the guide tree: half of the cells of a progressive pass) hold a single 1.0, START
        // and the rows beyond the graph hold nothing.  The sum of such a row with column c is ((+0 + 0 T[c][0]) + ...) + 1 T[c][s]
        // + ... = 0.0f + T[c][s] (the other terms are zeros of either sign), and a1 depends on s alone: the score of a cell is
        // a function of (s, c).  If every row of the workgroup is of that kind, the 21 x COLS scores are evaluated once (the
        // same operations on the same operands as below) and the cells look them up.
        float *const tab = (float *)tp;      // [COLS][21], in the space of the staged columns (not staged on this path)
        static_assert(COLS * 21 <= (COLS + 1) * STR * 2, "score table must fit the column staging area");
        __shared__ float As[32];
        const uint32_t fb0 = 4u * blockIdx.y + threadIdx.x / PGM_ROWS;
        const uint32_t fl = threadIdx.x % PGM_ROWS;
        const bool active = fb0 < nvb;
        int sym = 20;
        bool plain = true;
        float a_row = 0.f;
        if (active) {
            const uint32_t y = row_of(fb0, fl);
            const uint32_t yc = (y + 1 < J.n1) ? y : 0u;
            const float4 *src = (const float4 *)(J.g1f + (size_t)DP * yc);
            int nz = 0;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float4 v = src[q];
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (vv[u] != 0.0f) { ++nz; sym = 4 * q + u; plain = plain && vv[u] == 1.0f; }
            }
            plain = plain && nz <= 1;
            a_row = J.a1[yc];
        }
        if (threadIdx.x < 32) As[threadIdx.x] = 0.f;
        if (__syncthreads_and(plain ? 1 : 0)) {
            if (active) As[sym] = a_row;   // (rows with the same content have the same a1: any of them writes it)
            __syncthreads();
            const float mi = J.sc.match_init;
            constexpr int NE = (COLS * 21 + 4 * PGM_ROWS - 1) / (4 * PGM_ROWS);   // table entries per thread: all loads first
            float tt[NE], tb2[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int i = (int)threadIdx.x + e * 4 * PGM_ROWS;
                const int ci = i / 21, k = i % 21, c = cbase + ci;
                const bool in = i < COLS * 21 && c >= 0 && c <= (int)J.ncol;
                tb2[e] = in ? J.b2[c] : 0.f;
                tt[e] = (in && k < 20) ? J.t2[(size_t)DP * c + k] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int i = (int)threadIdx.x + e * 4 * PGM_ROWS;
                if (i < COLS * 21) tab[i] = pgm_emission_finish(0.0f + tt[e], As[i % 21], tb2[e], mi);
            }
            __syncthreads();
            if (!active) return;
            PGM_GLOBAL pgm_v4f *const Sq = (PGM_GLOBAL pgm_v4f *)(uintptr_t)J.S;
            const int l = PGM_HALO + (int)fl;
#pragma unroll 1
            for (int w = 0; w < PGM_EM_TB; ++w) {
                const uint32_t tb = tb0 + w;
                if (tb >= J.nblk) break;
                float out[PGM_BLOCK];
#pragma unroll
                for (int i = 0; i < PGM_BLOCK; ++i) out[i] = tab[((int)(tb * PGM_BLOCK) + i - l - cbase) * 21 + sym];
                PGM_GLOBAL pgm_v4f *dst = Sq + (sblock_of(fb0, tb) * 64u + (uint32_t)l) * (PGM_BLOCK / 4);
#pragma unroll
                for (int q = 0; q < PGM_BLOCK / 4; ++q) dst[q] = pgm_v4f{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
            }
            return;
        }
    }
    // one float4 load per (column, 4 k): the value T[c][k] is the low half of pair c and the high half of pair c - 1
    const float4 *t2q = (const float4 *)J.t2;
    float *tpf = (float *)tp;
    for (int i = threadIdx.x; i < (COLS + 1) * NT; i += 4 * PGM_ROWS) {
        const int ci = i / NT, q = i % NT, c = cbase + ci;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c >= 0 && c <= (int)J.ncol) v = t2q[(size_t)NT * c + q];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * q + u;
            if (ci < COLS) tpf[2 * (ci * STR + k)] = vv[u];
            if (ci > 0) tpf[2 * ((ci - 1) * STR + k) + 1] = vv[u];
        }
    }
    for (int i = threadIdx.x; i < COLS; i += 4 * PGM_ROWS) {
        const int col = cbase + i;
        bq[i] = (col >= 0 && col <= (int)J.ncol) ? J.b2[col] : 0.f;
    }
    __syncthreads();
    const uint32_t b0 = (4u * blockIdx.y + threadIdx.x / PGM_ROWS) * RB;   // this thread's bands: b0 .. b0 + RB - 1 (same lane, same columns)
    const int l = PGM_HALO + (int)(threadIdx.x % PGM_ROWS);
    if (b0 >= nvb) return;
    float gy[RB][DP], ay[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const uint32_t y = row_of(b0 + (uint32_t)r, (uint32_t)(l - PGM_HALO));
        const uint32_t yc = (y + 1 < J.n1) ? y : 0u;
        const float4 *src = (const float4 *)(J.g1f + (size_t)DP * yc);
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const float4 v = src[q];
            gy[r][4 * q] = v.x; gy[r][4 * q + 1] = v.y; gy[r][4 * q + 2] = v.z; gy[r][4 * q + 3] = v.w;
        }
        ay[r] = J.a1[yc];
    }
    const float mi = J.sc.match_init;
    const uint32_t nblk = J.nblk;
    PGM_GLOBAL pgm_v4f *const Sq = (PGM_GLOBAL pgm_v4f *)(uintptr_t)J.S;
#pragma unroll 1
    for (int w = 0; w < PGM_EM_TB; ++w) {
        const uint32_t tb = tb0 + w;
        if (tb >= nblk) break;
        float out[RB][PGM_BLOCK];
#pragma unroll
        for (int i = 0; i < PGM_BLOCK; i += 2) {
            const int ci = (int)(tb * PGM_BLOCK) + i - l - cbase;   // column of cell i in the staged window; cell i + 1: column ci + 1
            const pgm_v2f *tc = tp + ci * STR;
            pgm_v2f acc[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = pgm_v2f{0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                const pgm_v2f tk = tc[k];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const pgm_v2f g = {gy[r][k], gy[r][k]};
                    acc[r] = acc[r] + g * tk;
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                out[r][i] = pgm_emission_finish(acc[r].x, ay[r], bq[ci], mi);
                out[r][i + 1] = pgm_emission_finish(acc[r].y, ay[r], bq[ci + 1], mi);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (b0 + (uint32_t)r < nvb) {
                PGM_GLOBAL pgm_v4f *dst = Sq + (sblock_of(b0 + (uint32_t)r, tb) * 64u + (uint32_t)l) * (PGM_BLOCK / 4);
#pragma unroll
                for (int q = 0; q < PGM_BLOCK / 4; ++q) dst[q] = pgm_v4f{out[r][4 * q], out[r][4 * q + 1], out[r][4 * q + 2], out[r][4 * q + 3]};
            }
        }
    }
}

// wave-uniform maximum of a per-lane value in 0..15 (four ballots, no cross-lane data movement)
__device__ __forceinline__ int pgm_wave_max8(uint32_t v) {
    int m = __builtin_amdgcn_ballot_w64(v >= 8u) != 0 ? 8 : 0;
    m += __builtin_amdgcn_ballot_w64(v >= (uint32_t)(m + 4)) != 0 ? 4 : 0;
    m += __builtin_amdgcn_ballot_w64(v >= (uint32_t)(m + 2)) != 0 ? 2 : 0;
    m += __builtin_amdgcn_ballot_w64(v >= (uint32_t)(m + 1)) != 0 ? 1 : 0;
    return m;
}

// ---------------------------------------------------------------------------------------------
// device-wide (agent scope) accesses to the DP storage: a band may be continued on another CU / XCD, whose L1
// never sees our stores and whose L2 is not coherent with ours, so cells are written through (sc1) and read
// with L1-bypassing loads; the hand-off itself is the progress counter below (MI355X guide, Guideline 16 R1).
// The cell store is a raw buffer store with aux = 16 (sc1, write-through to device scope): unlike an inline-asm
// store it is counted by hipcc's s_waitcnt bookkeeping, so the counted waits on the prefetch loads stay exact.
// The descriptor covers one band (tsteps * 64 cells); step and lane both go into the VGPR offset (see below).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pgm_band_rsrc(float4 *base, uint32_t bytes) {
    const uint64_t p = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// Same store issued by every lane: lanes with `on == false` get an offset beyond the descriptor's num_records and the
// buffer range check drops them.  No branch around the store, so every step issues exactly one vector memory
// instruction, which is what lets the compiler turn the waits on the block prefetch into counted waits (vmcnt(8)).
// The step offset goes into the VGPR offset, NOT into the scalar offset: a 128-bit buffer store reads its data registers
// over several cycles, and a VALU instruction right behind it that overwrites one of them needs wait states.  hipcc
// (ROCm 7.2) inserts them only for stores WITHOUT an SGPR soffset (it assumes the hazard does not exist with one); on
// gfx950 it does: with `soffset = t * 1024` the M word of lanes 12-15 of every 16 was overwritten by the next step's
// column index whenever the register allocator reused the register (found with the 64-row re-lay of round 2).
__device__ __forceinline__ void pgm_store_cell_masked(__amdgpu_buffer_rsrc_t rsrc, uint32_t t, int lane, bool on, float Mv, float Xv, float Wv, float Yv) {
    typedef uint32_t pgm_v4u __attribute__((ext_vector_type(4)));
    pgm_v4u v;
    v.x = __float_as_uint(Mv); v.y = __float_as_uint(Xv); v.z = __float_as_uint(Wv); v.w = __float_as_uint(Yv);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (on ? (uint32_t)lane * 16u : 0x80000000u) + t * 1024u, 0, 16);
}
// Loads through global-address-space pointers (global_load, returned in issue order and counted by vmcnt) instead of
// generic ones (flat_load, which the compiler can only wait for with vmcnt(0)).
__device__ __forceinline__ float4 pgm_gload4(const float4 *p) {
    const pgm_v4f v = *(const PGM_GLOBAL pgm_v4f *)(uintptr_t)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 pgm_gload_cell_wy(const float4 *p) {   // {W, Y}, device-coherent
    const unsigned long long v = __hip_atomic_load((const PGM_GLOBAL unsigned long long *)(uintptr_t)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}
__device__ __forceinline__ float2 pgm_load_cell_mx(const float4 *p) {   // {M, X}
    const unsigned long long v = __hip_atomic_load((const PGM_GLOBAL unsigned long long *)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}
__device__ __forceinline__ float2 pgm_load_cell_wy(const float4 *p) {   // {W, Y}
    const unsigned long long v = __hip_atomic_load((const PGM_GLOBAL unsigned long long *)(uintptr_t)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}

// ---------------------------------------------------------------------------------------------
// Traceback (GraphAlign.h:264-521), run by the fill worker that completes a job's last band (all cells of the job are
// then written through to memory).  Thread 0 walks END -> START with the reference's tie rules (smallest
// |current - recomputed|, first candidate in PredIterator order wins, extension tested before opening, state after a
// W source chosen by equality in the order M, Y, X).  The walk is a pointer chase (one cell decides which cell is
// read next), so the whole workgroup stages a 32 x 32 TILE of cells, emission scores and predecessor lists around the
// walker's position in LDS; the walker then reads LDS (~100 cycles) instead of L2/HBM (~1000 cycles) per dependent
// access and asks for a new tile when it gets within a few rows or columns of the tile's low edge.  Anything
// outside the tile (far skip / repeat edges, nodes with more than 8 predecessors) is read from memory as before.
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

// S(y,x) as the emission kernel stored it (skewed order, see PgmJob::S): one load instead of recomputing the dot product
__device__ __forceinline__ float pgm_emission_at(const PgmJob &J, uint32_t y, uint32_t x) {
    const uint32_t b = y / PGM_ROWS, l = PGM_HALO + (y - b * PGM_ROWS), t = x + l;
    return J.S[(((size_t)b * J.nblk + (t / PGM_BLOCK)) * 64u + l) * PGM_BLOCK + (t % PGM_BLOCK)];
}

#ifndef PGM_POLL_PREFETCH
#define PGM_POLL_PREFETCH 1
#endif
#define PGM_TB_T 32        // tile edge (rows and columns)
#define PGM_TB_BAND 6      // successor links are built for the diagonals within this distance of the walker's (half of it for chain-only jobs)
#define PGM_TB_LK 4        // an M-state link is precomputed for cells whose two nodes have at most this many predecessors each
struct PgmTbLds {
    float4 cell[PGM_TB_T * PGM_TB_T];          // {M, X, W, Y} of rows ty0.., columns tx0..
    float S[PGM_TB_T * PGM_TB_T];
    uint32_t p_cnt[2 * PGM_TB_T];              // slots 0..T-1: rows ty0.. of graph 1; T..2T-1: columns tx0.. of graph 2
    uint32_t p_c[2 * PGM_TB_T * PGM_TB_PK];    // predecessor node, PredIterator order
    float p_v[2 * PGM_TB_T * PGM_TB_PK];       // edge value
    uint32_t p_u[2 * PGM_TB_T * PGM_TB_PK];    // repeat marker
    uint16_t succ[3 * PGM_TB_T * PGM_TB_T];    // [state][cell]: precomputed successor (see pgm_tb_succ), 0 = not available
    uint32_t ty0, tx0;                         // tile origin
    uint32_t ay, ax;                           // walker position the tile was requested for (centre of the link band)
    int req;                                   // 1: stage a tile at (ty0, tx0); 2: walk finished
    uint32_t len;
    int nbig;                                  // cells of the link band whose nodes have 5..8 predecessors (M link by a whole wavefront)
    uint16_t big[128];
    uint16_t gtab[PGM_LK_NR][PGM_LK_W * PGM_LK_TAB] __attribute__((aligned(16)));   // pre-linked tables of up to PGM_LK_NR grid rows (copies of PgmJob::ltab[row]), ring slot = row % PGM_LK_NR
    uint2 mbuf[64];                            // mapping entries of a link chase, written out 64 at a time
    uint32_t grow, gcnt;                       // request: load the rows grow, grow - 1, ..., grow - gcnt + 1
    uint32_t gnext[PGM_LK_NR];                 // [k] = 1: row grow - gcnt - k was complete when the request was served (looked at in the same round trip)
};

// Staging of one grid tile for pgm_prelink_tile: the tile's cells with a halo of PGM_LK_H rows above and columns to the left
// (predecessor cells), the scores and the predecessor records of the tile's own rows and columns, and the table being built.
#define PGM_LK_HW (PGM_LK_T + PGM_LK_H)
struct PgmLkLds {
    float4 cell[PGM_LK_HW * PGM_LK_HW];        // rows hy0.., columns hx0..
    float S[PGM_LK_T * PGM_LK_T];
    uint32_t p_cnt[2 * PGM_LK_T];
    uint32_t p_c[2 * PGM_LK_T * PGM_TB_PK];
    float p_v[2 * PGM_LK_T * PGM_TB_PK];
    uint32_t p_u[2 * PGM_LK_T * PGM_TB_PK];
    uint16_t tab[PGM_LK_TAB] __attribute__((aligned(16)));
    int nbig;
    uint16_t big[PGM_LK_T * PGM_LK_T];
};
// A pre-linked successor: bit 15 valid, bits 12..13 next state (0 M, 1 X, 2 Y), bits 6..11 / 0..5 rows / columns back to the next cell.
__device__ __forceinline__ uint16_t pgm_lk_code(uint32_t st, uint32_t dy, uint32_t dx) { return (uint16_t)(0x8000u | (st << 12) | (dy << 6) | dx); }

// predecessor list of one node: either in the tile (LDS) or in memory
struct PgmPredView {
    uint32_t cnt;
    bool lds;
    uint32_t base;     // LDS: slot * PGM_TB_PK; memory: first entry index
};

// Mappings of a finished walk: reversed into alignment order (GraphAlign.h:520-521) by all threads of the worker, straight
// into the pinned host block (posted PCIe writes, 1 KB per wavefront-iteration), then the result record, status word last: a
// host that copies finished jobs while the kernel is still running (pgm_align_batch_fetch) polls that word.
__device__ static void pgm_traceback_publish(const PgmJob &J, const uint32_t len, const int tid, const int nthreads) {
    for (uint32_t i = (uint32_t)tid; i < len; i += (uint32_t)nthreads) {
        const uint32_t j = len - 1 - i;
        J.hmap1[i] = J.map1[j];
        J.hmap2[i] = J.map2[j];
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const PgmJob::Result r = *J.result;
        J.hresult->score = r.score; J.hresult->n_tr_indels = r.n_tr_indels; J.hresult->len = r.len;
        __threadfence_system();
        __hip_atomic_store(&J.hresult->status, r.status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// lq: the batch's queue of pre-link tasks (pgm_fill_kernel's idle phase): lq[0] tasks announced, lq[1] tracebacks finished, lq_ids[k] = job + 1
__device__ static void pgm_traceback_job(const PgmJob &J, PgmTbLds &T, const int tid, const int nthreads, unsigned long long *stat,
                                         int *lq, int *lq_ids, const uint32_t jid) {
    constexpr uint32_t TT = PGM_TB_T;
    const uint32_t n1 = J.n1, n2 = J.n2;

    const PgmPred P1 = {J.pp1, J.pc1, J.pv1, J.pu1};
    const PgmPred P2 = {J.pp2, J.pc2, J.pv2, J.pu2};

    // ---- tile staging, all threads (nthreads == 512: 2 cells and 1 predecessor entry each).  All loads of a thread
    // are issued before the first LDS store (clamped, always valid addresses instead of branches), so a tile costs two
    // memory round trips (row pointers -> entries), not one per element. -------------------------------------------
    const bool loader = tid >= 64;   // wavefronts 1..: they only serve the walker (wavefront 0)
    auto stage = [&]() {
        const uint32_t ty0 = T.ty0, tx0 = T.tx0;
        constexpr uint32_t NT = 64u * PGM_WAVES;
        constexpr int NC = (int)((TT * TT) / NT), NP = (int)((2 * TT * PGM_TB_PK) / NT);
        static_assert(NC * NT == TT * TT && NP * NT == 2 * TT * PGM_TB_PK, "tile staging: whole rounds of the workgroup");
        // The cells of one anti-diagonal of the tile (y + x constant) are one contiguous run of the cell storage (same step of the
        // band, consecutive lanes), the cells of a row are 1 KB apart: the threads therefore take the tile's cells in
        // anti-diagonal order — position k of diagonal d, the 63 diagonals laid end to end — so that a wavefront's 64 cell loads
        // touch 8 cache lines instead of 64.  (The scores are laid out in blocks of 8 steps per lane: row order suits them.)
        float4 cv[NC];
        float sv[NC];
        uint32_t cslot[NC];
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const uint32_t i = (uint32_t)tid + NT * u;
            // (d, k) of the i-th cell in anti-diagonal order: the first TT diagonals hold 1, 2, ..., TT cells, the rest mirror them
            const uint32_t half = TT * (TT + 1u) / 2u;
            const bool up = i < half;
            const uint32_t m = up ? i : TT * TT - 1u - i;
            uint32_t d = (uint32_t)((sqrtf(8.0f * (float)m + 1.0f) - 1.0f) * 0.5f);
            while (d * (d + 1u) / 2u > m) --d;
            while ((d + 1u) * (d + 2u) / 2u <= m) ++d;
            const uint32_t k = m - d * (d + 1u) / 2u;             // 0..d: row k, column d - k of the upper triangle
            const uint32_t ly = up ? k : TT - 1u - k, lx = up ? d - k : TT - 1u - (d - k);
            cslot[u] = ly * TT + lx;
            const uint32_t yy = min(ty0 + ly, n1 - 2), xx = min(tx0 + lx, n2 - 2);
            cv[u] = J.cells[pgm_cell_index(J, yy, xx)];
            const uint32_t ys = min(ty0 + i / TT, n1 - 2), xs = min(tx0 + i % TT, n2 - 2);
            sv[u] = pgm_emission_at(J, ys, xs);
        }
        // predecessor entries: slot = row (0..TT-1) or column (TT..2TT-1) of the tile, k = entry; straight from the
        // per-node records (same round trip as the cells)
        uint32_t cnt[NP], ec[NP], eu[NP];
        float ev[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const uint32_t i = (uint32_t)tid + NT * u, slot = i / PGM_TB_PK, k = i % PGM_TB_PK;
            const bool row = slot < TT;
            const uint32_t v = min(row ? ty0 + slot : tx0 + (slot - TT), (row ? n1 : n2) - 1);
            const PgmTbNode *r = (row ? J.tb1 : J.tb2) + v;
            cnt[u] = r->cnt; ec[u] = r->c[k]; ev[u] = r->v[k]; eu[u] = r->u[k];
        }
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const uint32_t i = (uint32_t)tid + NT * u;
            T.cell[cslot[u]] = cv[u];
            T.S[i] = sv[u];
        }
        for (uint32_t i = (uint32_t)tid; i < 3u * TT * TT / 2u; i += NT) ((uint32_t *)T.succ)[i] = 0u;   // no links yet
        if (tid == 0) T.nbig = 0;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const uint32_t i = (uint32_t)tid + NT * u, slot = i / PGM_TB_PK, k = i % PGM_TB_PK;
            const bool row = slot < TT;
            const bool inr = (row ? ty0 + slot : tx0 + (slot - TT)) < (row ? n1 : n2);
            if (k == 0) T.p_cnt[slot] = inr ? cnt[u] : 0xFFFFFFFFu;
            T.p_c[i] = ec[u]; T.p_v[i] = ev[u]; T.p_u[i] = eu[u];
        }
    };
    // ---- successor table, all threads: for every cell of the tile and each of the three states the walk's next
    // (cell, state), decided exactly as the walker would (same candidates, same order, same strict comparisons).  The
    // walker then follows 16-bit links (one LDS read per step) and only evaluates candidates itself where no link
    // exists: predecessors outside the tile, more than PGM_TB_PK predecessors, tandem-repeat edges, inconsistencies.
    // Link: bit 15 valid, bits 10..11 next state (0 M, 1 X, 2 Y), bits 5..9 / 0..4 next row / column inside the tile.
    const pgm_scores sc = J.sc;
    // Only for the cells within PGM_TB_BAND diagonals of the walker's position (T.ay, T.ax) when the tile was requested:
    // an alignment path mostly runs along that diagonal; off the band the walker asks for a new tile.
    const int band = J.has_extras ? PGM_TB_BAND : PGM_TB_BAND / 2;   // alignments of merged graphs wander more (gaps)
    auto links = [&]() {
        const uint32_t ty0 = T.ty0, tx0 = T.tx0;
        const int adiag = (int)(T.ay - ty0) - (int)(T.ax - tx0);
        // one task = (state, diagonal of the band, position on it); a thread's tasks mix the three states
        const uint32_t NDIAG = 2u * (uint32_t)band + 1u;
        for (uint32_t task = (uint32_t)tid; task < 3u * NDIAG * TT; task += (uint32_t)nthreads) {
            const uint32_t lx = task % TT, st_task = task / (NDIAG * TT);
            const int lyi = (int)lx + adiag + (int)((task / TT) % NDIAG) - band;
            if (lyi < 0 || lyi >= (int)TT) continue;
            const uint32_t ly = (uint32_t)lyi, ci = ly * TT + lx, y = ty0 + ly, x = tx0 + lx;
            uint16_t lm = 0, lxs = 0, lys = 0;
            if (y + 1 < n1 && x + 1 < n2 && (y | x) != 0u) {
                const float4 c0 = T.cell[ci];
                const uint32_t cy = T.p_cnt[ly], cx = T.p_cnt[TT + lx];
                auto pick = [&](const float4 &c, uint32_t yp, uint32_t xp) -> uint16_t {   // W source resolved by equality: M, Y, X
                    uint32_t st;
                    if ((yp | xp) == 0u) st = 0u;
                    else if (c.z == c.x) st = 0u;
                    else if (c.z == c.w) st = 2u;
                    else if (c.z == c.y) st = 1u;
                    else return (uint16_t)0;
                    return (uint16_t)(0x8000u | (st << 10) | ((yp - ty0) << 5) | (xp - tx0));
                };
                // state M: pairs (row predecessor outer, column predecessor inner)
                // (cells with more predecessors get no M link: one such cell would hold up the whole tile; the walker
                // evaluates them itself, one pair per lane, if the path really visits them)
                const int doff = (int)((task / TT) % NDIAG) - band;   // diagonal relative to the walker's
                if (st_task == 0u && doff >= -1 && doff <= 1 && (cy > PGM_TB_LK || cx > PGM_TB_LK) && cy <= PGM_TB_PK && cx <= PGM_TB_PK && cy != 0u && cx != 0u && c0.x > PGM_NEG_INF) {
                    // a node with 5..8 predecessors on one of the three diagonals next to the walker's: see links_big
                    const int slot = __hip_atomic_fetch_add(&T.nbig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (slot < 128) T.big[slot] = (uint16_t)ci;
                }
                if (st_task == 0u && cy <= PGM_TB_LK && cx <= PGM_TB_LK && cy != 0u && cx != 0u && c0.x > PGM_NEG_INF) {
                    const float S = T.S[ci];
                    // all operands first (the list slots beyond a node's count hold a copy of its last entry, so every read
                    // is a valid one and nothing here branches), then the comparisons in PredIterator order
                    uint32_t yp[PGM_TB_LK], xp[PGM_TB_LK];
                    float yv[PGM_TB_LK], xv[PGM_TB_LK];
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < PGM_TB_LK; ++k) {
                        yp[k] = T.p_c[ly * PGM_TB_PK + k]; yv[k] = T.p_v[ly * PGM_TB_PK + k];
                        xp[k] = T.p_c[(TT + lx) * PGM_TB_PK + k]; xv[k] = T.p_v[(TT + lx) * PGM_TB_PK + k];
                        ok = ok && ((uint32_t)k >= cy || yp[k] >= ty0) && ((uint32_t)k >= cx || xp[k] >= tx0);
                    }
                    float wz[PGM_TB_LK][PGM_TB_LK];
#pragma unroll
                    for (int ky = 0; ky < PGM_TB_LK; ++ky)
#pragma unroll
                        for (int kx = 0; kx < PGM_TB_LK; ++kx)
                            wz[ky][kx] = T.cell[min((yp[ky] - ty0) * TT + (xp[kx] - tx0), TT * TT - 1u)].z;
                    float best = INFINITY;
                    uint32_t wy = 0, wx = 0, wky = 0, wkx = 0;
#pragma unroll
                    for (int ky = 0; ky < PGM_TB_LK; ++ky)
#pragma unroll
                        for (int kx = 0; kx < PGM_TB_LK; ++kx) {
                            const float d = fabsf(__fsub_rn(c0.x, __fsub_rn(__fsub_rn(__fadd_rn(wz[ky][kx], S), yv[ky]), xv[kx])));
                            if ((uint32_t)ky < cy && (uint32_t)kx < cx && best > d) { best = d; wy = yp[ky]; wx = xp[kx]; wky = ky; wkx = kx; }
                        }
                    bool rep_edge = false;
                    float4 wc = c0;
                    if (ok && best < INFINITY) {
                        rep_edge = (T.p_u[ly * PGM_TB_PK + wky] | T.p_u[(TT + lx) * PGM_TB_PK + wkx]) != 0u;
                        wc = T.cell[(wy - ty0) * TT + (wx - tx0)];
                    }
                    if (ok && best < INFINITY && !rep_edge) lm = pick(wc, wy, wx);
                }
                // state Y: row predecessors, extension before opening
                // (gap states are rare on a path: links only where they cost two candidates; other cells are evaluated by the
                // walker if it really arrives there in a gap state)
                if (st_task == 2u && cy == 1u && c0.w > PGM_NEG_INF) {
                    float best = INFINITY;
                    bool ok = true, rep_edge = false, open = false;
                    uint32_t wy = 0;
                    float4 wc = c0;
                    for (uint32_t k = 0; k < cy; ++k) {
                        const uint32_t yp = T.p_c[ly * PGM_TB_PK + k];
                        if (yp < ty0) { ok = false; break; }
                        const float yv = T.p_v[ly * PGM_TB_PK + k];
                        const float4 c = T.cell[(yp - ty0) * TT + lx];
                        const bool ru = T.p_u[ly * PGM_TB_PK + k] != 0u;
                        float d = fabsf(__fsub_rn(c0.w, __fsub_rn(__fadd_rn(c.w, sc.gap_extend), yv)));
                        if (best > d) { best = d; wy = yp; wc = c; rep_edge = ru; open = false; }
                        d = fabsf(__fsub_rn(c0.w, __fsub_rn(__fadd_rn(c.z, sc.gap_init), yv)));
                        if (best > d) { best = d; wy = yp; wc = c; rep_edge = ru; open = true; }
                    }
                    if (ok && best < INFINITY && !rep_edge)
                        lys = open ? pick(wc, wy, x) : (uint16_t)(0x8000u | (2u << 10) | ((wy - ty0) << 5) | lx);
                }
                // state X: column predecessors
                if (st_task == 1u && cx == 1u && c0.y > PGM_NEG_INF) {
                    float best = INFINITY;
                    bool ok = true, rep_edge = false, open = false;
                    uint32_t wx = 0;
                    float4 wc = c0;
                    for (uint32_t k = 0; k < cx; ++k) {
                        const uint32_t xp = T.p_c[(TT + lx) * PGM_TB_PK + k];
                        if (xp < tx0) { ok = false; break; }
                        const float xv = T.p_v[(TT + lx) * PGM_TB_PK + k];
                        const float4 c = T.cell[ly * TT + (xp - tx0)];
                        const bool ru = T.p_u[(TT + lx) * PGM_TB_PK + k] != 0u;
                        float d = fabsf(__fsub_rn(c0.y, __fsub_rn(__fadd_rn(c.y, sc.gap_extend), xv)));
                        if (best > d) { best = d; wx = xp; wc = c; rep_edge = ru; open = false; }
                        d = fabsf(__fsub_rn(c0.y, __fsub_rn(__fadd_rn(c.z, sc.gap_init), xv)));
                        if (best > d) { best = d; wx = xp; wc = c; rep_edge = ru; open = true; }
                    }
                    if (ok && best < INFINITY && !rep_edge)
                        lxs = open ? pick(wc, y, wx) : (uint16_t)(0x8000u | (1u << 10) | (ly << 5) | (wx - tx0));
                }
            }
            T.succ[st_task * (TT * TT) + ci] = st_task == 0u ? lm : (st_task == 1u ? lxs : lys);
        }
    };
    // M links of the band's cells whose nodes have 5..8 predecessors (collected by links(); merged graphs of deep trees have a
    // few per tile, and without a link each costs the walker 2 us): one wavefront per cell, one candidate pair per lane in
    // PredIterator order (lane = 8 ky + kx), exactly the walker's evaluation.  No link if a predecessor lies outside the
    // tile, if no candidate recomputes the stored value exactly, or for a repeat edge: the walker then decides itself.
    auto links_big = [&]() {
        const int wave = tid >> 6, lane_b = tid & 63, nwaves = nthreads >> 6;
        const uint32_t ty0 = T.ty0, tx0 = T.tx0;
        const int nbig = min(T.nbig, 128);
        for (int i = wave; i < nbig; i += nwaves) {
            const uint32_t ci = T.big[i], ly = ci / TT, lx = ci % TT;
            const uint32_t cy = T.p_cnt[ly], cx = T.p_cnt[TT + lx];
            const uint32_t ky = (uint32_t)lane_b >> 3, kx = (uint32_t)lane_b & 7u;
            const bool valid = ky < cy && kx < cx;
            const uint32_t yp = T.p_c[ly * PGM_TB_PK + ky], xp = T.p_c[(TT + lx) * PGM_TB_PK + kx];
            const float yv = T.p_v[ly * PGM_TB_PK + ky], xv = T.p_v[(TT + lx) * PGM_TB_PK + kx];
            const uint32_t rep = T.p_u[ly * PGM_TB_PK + ky] | T.p_u[(TT + lx) * PGM_TB_PK + kx];
            const bool inside = yp >= ty0 && xp >= tx0;
            if (__builtin_amdgcn_ballot_w64(valid && !inside) != 0ull) continue;
            const float4 c0 = T.cell[ci];
            const float S = T.S[ci];
            const float4 c = T.cell[(valid && inside) ? (yp - ty0) * TT + (xp - tx0) : ci];
            const float d = fabsf(__fsub_rn(c0.x, __fsub_rn(__fsub_rn(__fadd_rn(c.z, S), yv), xv)));
            const unsigned long long zero = __builtin_amdgcn_ballot_w64(valid && d == 0.0f);
            if (zero == 0ull) continue;
            const int win = __ffsll((long long)zero) - 1;
            if (lane_b == win && rep == 0u) {
                uint32_t st = 3u;
                if ((yp | xp) == 0u) st = 0u;
                else if (c.z == c.x) st = 0u;
                else if (c.z == c.w) st = 2u;
                else if (c.z == c.y) st = 1u;
                if (st != 3u) T.succ[ci] = (uint16_t)(0x8000u | (st << 10) | ((yp - ty0) << 5) | (xp - tx0));
            }
        }
    };
    // (Deciding every visited cell in the walker instead of precomputing links — tried for the MODE 2 jobs, whose paths visit 11
    // cells per tile: a lone wavefront needs 1.1 us per cell for the two LDS round trips and ~100 dependent instructions of the
    // fast evaluation below, the root's traceback went from 0.93 to 1.27 ms.  The links stay; the fast evaluation serves the
    // cells without one.)
    const bool use_links = true;
    // the pre-linked tables of the grid rows T.grow, T.grow - 1, ... (T.gcnt of them, all known to be complete) -> LDS, all
    // threads: six 8-byte device-coherent loads per row and thread, all rows in one round trip; with them the completion counts
    // of the PGM_LK_NR rows above, which is where the walker goes next
    auto load_rows = [&]() {
        const uint32_t gy = T.grow, cnt = T.gcnt;
        constexpr uint32_t NW = PGM_LK_W * PGM_LK_TAB * 2u / 8u, NT = 64u * PGM_WAVES, PER = NW / NT;
        static_assert(NW % NT == 0u, "row table: whole rounds of the workgroup");
        unsigned long long v[PGM_LK_NR][PER];
#pragma unroll
        for (uint32_t r = 0; r < PGM_LK_NR; ++r) {
            const PGM_GLOBAL unsigned long long *src = (const PGM_GLOBAL unsigned long long *)(uintptr_t)(J.ltab + (size_t)(gy - min(r, cnt - 1u)) * (PGM_LK_W * PGM_LK_TAB));
#pragma unroll
            for (uint32_t q = 0; q < PER; ++q) v[r][q] = r < cnt ? __hip_atomic_load(src + (uint32_t)tid + NT * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        uint32_t nx = 0u;
        const uint32_t k = (uint32_t)tid - 64u;
        if (k < PGM_LK_NR && gy >= cnt + k) nx = (uint32_t)__hip_atomic_load(J.lready + (gy - cnt - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= min(J.lcols, PGM_LK_W) ? 1u : 0u;
#pragma unroll
        for (uint32_t r = 0; r < PGM_LK_NR; ++r)
            if (r < cnt) {
#pragma unroll
                for (uint32_t q = 0; q < PER; ++q) ((unsigned long long *)T.gtab[(gy - r) % PGM_LK_NR])[(uint32_t)tid + NT * q] = v[r][q];
            }
        if (k < PGM_LK_NR) T.gnext[k] = nx;
    };
    if (tid == 0) {
        T.grow = 0xFFFFFFFFu; T.gcnt = 0u;
        if (J.lrows != 0u) __hip_atomic_store(J.lready + J.lrows + 1, (int)(J.lrows - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T.ty0 = n1 - 2 >= TT - 1 ? n1 - 2 - (TT - 1) : 0u;
        T.tx0 = n2 - 2 >= TT - 1 ? n2 - 2 - (TT - 1) : 0u;
        T.ay = n1 - 2; T.ax = n2 - 2;
        T.req = 1;
        if (J.lrows != 0u && lq) {   // idle workers may start on this job's corridor now
            const int slot = __hip_atomic_fetch_add(lq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(lq_ids + slot, (int)(jid + 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    stage();
    __syncthreads();
    if (use_links) {
        links();
        __syncthreads();
        links_big();
        __syncthreads();
    }

    if (loader) {
        // ---- loaders (wavefronts 1..): serve the walker's tile requests ----
        for (;;) {
            __syncthreads();            // request posted
            if (T.req == 2) break;
            if (T.req == 3) { load_rows(); __syncthreads(); continue; }
            stage();
            __syncthreads();            // cells, scores, predecessor lists staged
            if (use_links) {
                links();
                __syncthreads();        // successor table complete but for the cells of links_big
                links_big();
                __syncthreads();
            }
        }
    } else {
        // ---- walker: wavefront 0, uniform control flow; the one-off END step is evaluated by lane 0 alone ----
        const int lane = tid;
        const pgm_scores s = J.sc;
        uint32_t ty0 = T.ty0, tx0 = T.tx0, ay = T.ay, ax = T.ax;
        int adiag = (int)(ay - ty0) - (int)(ax - tx0);
        auto in_tile = [&](uint32_t y, uint32_t x) { return (y - ty0) < TT && (x - tx0) < TT; };
        auto cell_at = [&](uint32_t y, uint32_t x) -> float4 {
            if (in_tile(y, x)) return T.cell[(y - ty0) * TT + (x - tx0)];
            return pgm_load_cell(J, y, x);
        };
        auto s_at = [&](uint32_t y, uint32_t x) -> float {
            if (in_tile(y, x)) return T.S[(y - ty0) * TT + (x - tx0)];
            return pgm_emission_at(J, y, x);
        };
        auto preds = [&](bool row, uint32_t v) -> PgmPredView {
            const uint32_t slot = row ? v - ty0 : (v - tx0) + TT;
            if ((row ? v - ty0 : v - tx0) < TT) {
                const uint32_t cnt = T.p_cnt[slot];
                if (cnt <= PGM_TB_PK) return PgmPredView{cnt, true, slot * PGM_TB_PK};
            }
            const PgmPred &P = row ? P1 : P2;
            const int32_t eb = P.pp[v];
            return PgmPredView{(uint32_t)(P.pp[v + 1] - eb), false, (uint32_t)eb};
        };
        auto pc_of = [&](bool row, const PgmPredView &V, uint32_t k) { return V.lds ? T.p_c[V.base + k] : (row ? P1 : P2).pc[V.base + k]; };
        auto pv_of = [&](bool row, const PgmPredView &V, uint32_t k) { return V.lds ? T.p_v[V.base + k] : (row ? P1 : P2).pv[V.base + k]; };
        auto pu_of = [&](bool row, const PgmPredView &V, uint32_t k) { return V.lds ? T.p_u[V.base + k] : (row ? P1 : P2).pu[V.base + k]; };

        int status = PGM_OK;
        uint32_t n_tr = 0;
        enum { State_m = 0, State_x = 1, State_y = 2 };
        int current_state = State_m;
        float current_score = PGM_NEG_INF, Wend = PGM_NEG_INF;
        uint32_t y = n1 - 1, x = n2 - 1;
        PgmMapOut mo = {J.map1, J.map2, 0u, n1 + n2};
        if (lane == 0) {
            // end node (GraphAlign.h:264-280)
            const PgmPredView Ey = preds(true, n1 - 1), Ex = preds(false, n2 - 1);
            for (uint32_t ky = 0; ky < Ey.cnt; ++ky) {
                for (uint32_t kx = 0; kx < Ex.cnt; ++kx) {
                    const uint32_t yp = pc_of(true, Ey, ky), xp = pc_of(false, Ex, kx);
                    const float yv = pv_of(true, Ey, ky), xv = pv_of(false, Ex, kx);
                    if (xp == 0 && yp == 0) {
                        Wend = fmaxf(__fsub_rn(__fsub_rn(s.end_skip, yv), xv), Wend);
                    } else {
                        const float4 c = cell_at(yp, xp);
                        Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.y, s.end_gap), yv), xv), Wend);
                        Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.w, s.end_gap), yv), xv), Wend);
                        Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.x, s.end_match), yv), xv), Wend);
                    }
                }
            }
            mo.push(n1 - 1, n2 - 1);
            bool tr_x = false, tr_y = false;
            float best = INFINITY;
            for (uint32_t ky = 0; ky < Ey.cnt; ++ky) {
                for (uint32_t kx = 0; kx < Ex.cnt; ++kx) {
                    const uint32_t yp = pc_of(true, Ey, ky), xp = pc_of(false, Ex, kx);
                    const float yv = pv_of(true, Ey, ky), xv = pv_of(false, Ex, kx);
                    const bool ry = pu_of(true, Ey, ky) != 0, rx = pu_of(false, Ex, kx) != 0;
                    const float4 c = cell_at(yp, xp);
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
        }
        // lane 0's result becomes the wavefront's (uniform) walker state
        y = __builtin_amdgcn_readfirstlane(y); x = __builtin_amdgcn_readfirstlane(x);
        current_state = __builtin_amdgcn_readfirstlane(current_state);
        current_score = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(current_score)));
        Wend = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(Wend)));
        n_tr = __builtin_amdgcn_readfirstlane(n_tr);
        mo.len = __builtin_amdgcn_readfirstlane(mo.len);

        // ---- the walk: every candidate predecessor of the current cell is evaluated by its own lane --------------------
        // Lane order = PredIterator order (row predecessor outer, column predecessor inner; in the gap states extension
        // before opening), so "first candidate with the smallest |current - recomputed|" is the first lane holding the
        // minimum.  In a consistent DP the producing candidate recomputes the stored value exactly (difference 0), which
        // one ballot finds; only if no lane is exact the minimum is searched lane by lane.
        auto rl_u = [](uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); };
        auto rl_f = [](float v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); };
        // mapping entries of a link chase: one LDS word pair per path node (lane 0), written to memory by all lanes 64 at a time
        // (two global stores per node cost the lone walker more than the link read itself)
        uint32_t nbuf = 0;
        auto buf_flush = [&]() {
            if (nbuf != 0u) {
                if ((uint32_t)lane < nbuf && mo.len + (uint32_t)lane < mo.cap) { const uint2 v = T.mbuf[lane]; mo.m1[mo.len + (uint32_t)lane] = v.x; mo.m2[mo.len + (uint32_t)lane] = v.y; }
                mo.len += nbuf;
                nbuf = 0u;
            }
        };
        auto buf_push = [&](uint32_t a, uint32_t b2) {
            if (lane == 0) T.mbuf[nbuf] = make_uint2(a, b2);
            if (++nbuf == 64u) buf_flush();
        };
        // a new tile is staged when the walker is closer than `margin` to the tile's low edge: predecessors of chain-only
        // graphs are 1 node back; in merged graphs most skip edges span a few nodes (farther ones are read from memory)
        const uint32_t margin = J.has_extras ? 4u : 1u;
        uint32_t guard = 0;
        unsigned long long st_reload = 0, st_nreload = 0, st_slow = 0, st_stage = 0, st_grid = 0;
        bool score_stale = false;
        uint32_t nl_y = 0xFFFFFFFFu, nl_x = 0xFFFFFFFFu;   // last cell at which the pre-linked tables had nothing to follow
        uint32_t glo = 1u, ghi = 0u;                       // grid rows whose tables are in LDS (none: lo > hi)
        uint32_t known = 0u;                               // rows below glo known to be complete
        while ((x != 0 || y != 0) && status == PGM_OK) {
            if (++guard > n1 + n2 + 4) { status = PGM_ERR_BACKTRACK; break; }
            // the walker state is the same in all lanes: keep it in scalar registers so that the branches below are scalar
            y = __builtin_amdgcn_readfirstlane(y); x = __builtin_amdgcn_readfirstlane(x);
            current_state = __builtin_amdgcn_readfirstlane(current_state);
            // new tile when the walker is near the low edge of the current one (or outside it, after a far edge)
            const int offb = ((int)(y - ty0) - (int)(x - tx0)) - adiag;
            const bool reload = ((y - ty0) >= TT || ((y - ty0) < margin && ty0 != 0)) || ((x - tx0) >= TT || ((x - tx0) < margin && tx0 != 0)) ||
                                (use_links && (offb < -band || offb > band) && (y != ay || x != ax));
            // Before staging a tile of its own the walker looks whether idle workers have pre-linked the grid tile it stands in
            // (pgm_prelink_tile): it then copies that tile's table (6 KB) and follows its links from grid tile to grid tile — one
            // flag and one table round trip per tile, no staging, no link pass — until it meets a cell without a link or a tile
            // that is not ready, and goes on there as before.
            bool nolink = false;
            if (reload && J.lrows != 0u && !(y == nl_y && x == nl_x)) {
                bool moved = false;
                uint32_t st = (uint32_t)(current_state == State_m ? 0 : (current_state == State_x ? 1 : 2));
                const uint32_t gw = min(J.lcols, PGM_LK_W);
                for (;;) {
                    const uint32_t gy = y >> 5;
                    if (gy >= J.lrows) break;
                    const uint32_t first = pgm_lk_first(n1, n2, J.lcols, gy);
                    if ((x >> 5) - first >= gw) break;                      // outside the corridor
                    if (!(gy >= glo && gy <= ghi)) {                        // row gy is not among the rows in LDS
                        // how many rows to fetch: gy, and as many below it as are KNOWN to be complete (from the look-ahead of the
                        // previous request, when gy continues the rows in LDS); gy itself costs a flag round trip otherwise
                        const bool cont = gy + 1u == glo && known != 0u;
                        if (!cont) {
                            if ((uint32_t)__hip_atomic_load(J.lready + gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gw) break;
                            known = 1u;
                        }
                        const uint32_t cnt = min(min(known, (uint32_t)PGM_LK_NR), gy + 1u);
                        if (lane == 0) {
                            T.grow = gy; T.gcnt = cnt; T.req = 3;
                            __hip_atomic_store(J.lready + J.lrows + 1, (int)(gy >= cnt ? gy - cnt : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the rows from there up are still wanted
                        }
                        __syncthreads();        // request posted
                        load_rows();
                        __syncthreads();
                        ghi = gy; glo = gy + 1u - cnt;
                        known = 0u;
                        while (known < (uint32_t)PGM_LK_NR && T.gnext[known] != 0u) ++known;
                    }
                    bool left_row = false;
                    const uint16_t *tab = T.gtab[gy % PGM_LK_NR];
                    for (;;) {
                        const uint32_t j = (x >> 5) - first;
                        if (j >= gw) break;
                        const uint32_t code = tab[j * PGM_LK_TAB + st * (TT * TT) + (y & 31u) * TT + (x & 31u)];
                        if (!(code & 0x8000u)) { nolink = true; break; }
                        y -= (code >> 6) & 63u; x -= code & 63u; st = (code >> 12) & 3u;
                        moved = true;
                        if ((y | x) == 0u) break;
                        buf_push(st == 1u ? 0xFFFFFFFFu : y, st == 2u ? 0xFFFFFFFFu : x);
                        if ((y >> 5) != gy) { left_row = true; break; }
                    }
                    y = __builtin_amdgcn_readfirstlane(y); x = __builtin_amdgcn_readfirstlane(x); st = __builtin_amdgcn_readfirstlane(st);
                    if (!left_row) break;
                }
                buf_flush();
                if (moved) {
                    current_state = st == 0u ? State_m : (st == 1u ? State_x : State_y);
                    score_stale = true;
                    ++st_grid;
                    continue;
                }
                nl_y = y; nl_x = x;   // nothing to follow from here: decide this cell the usual way, then look again
            }
            // A cell without a link inside a pre-linked row (a predecessor beyond the halo, more than eight predecessors, a repeat
            // edge): the general code below decides it from memory, and the walk goes on in the pre-linked rows — staging and
            // linking a tile of the walker's own for that one cell costs twice as much
            if (reload && !(nolink && y == nl_y && x == nl_x)) {
                ty0 = y >= TT - 1 ? y - (TT - 1) : 0u;
                tx0 = x >= TT - 1 ? x - (TT - 1) : 0u;
                const unsigned long long r0 = stat ? __builtin_amdgcn_s_memrealtime() : 0ull;
                ay = y; ax = x; adiag = (int)(y - ty0) - (int)(x - tx0);
                if (lane == 0) { T.ty0 = ty0; T.tx0 = tx0; T.ay = y; T.ax = x; T.req = 1; }
                __syncthreads();        // request posted
                stage();
                __syncthreads();        // cells, scores, predecessor lists staged
                if (stat) st_stage += __builtin_amdgcn_s_memrealtime() - r0;
                if (use_links) {
                    links();
                    __syncthreads();    // successor table complete but for the cells of links_big
                    links_big();
                    __syncthreads();
                }
                if (stat) { st_reload += __builtin_amdgcn_s_memrealtime() - r0; ++st_nreload; }
            }
            // fast path: follow the precomputed links.  A link's low 12 bits are the table index of the next (state, cell),
            // so the chase is one LDS read per step; it ends at the first (cell, state) without a link (off the band, near
            // the tile's low edge, repeat edge, ...), which the code below handles.
            if (in_tile(y, x)) {
                uint32_t pos = (uint32_t)(current_state == State_m ? 0 : (current_state == State_x ? 1 : 2)) * (TT * TT) + (y - ty0) * TT + (x - tx0);
                bool moved = false;
                for (;;) {
                    const uint32_t code = T.succ[pos];
                    if (!(code & 0x8000u)) break;
                    pos = code & 0xfffu;
                    moved = true;
                    const uint32_t ny = ty0 + ((pos >> 5) & 31u), nx = tx0 + (pos & 31u), st = pos >> 10;
                    if ((ny | nx) == 0u) break;
                    buf_push(st == 1u ? 0xFFFFFFFFu : ny, st == 2u ? 0xFFFFFFFFu : nx);
                }
                buf_flush();
                if (moved) {
                    y = ty0 + ((pos >> 5) & 31u); x = tx0 + (pos & 31u);
                    const uint32_t st = pos >> 10;
                    current_state = st == 0u ? State_m : (st == 1u ? State_x : State_y);
                    score_stale = true;
                    continue;
                }
            }
            // fast evaluation: the candidates of the current (cell, state) one per lane in PredIterator order (state M: lane = 8 ky
            // + kx; gap states: lane = 2 k + (0 extension | 1 opening)), every operand from the staged tile: two LDS round trips and
            // one ballot per path node.  Anything else — a list longer than the tile's records, a candidate cell outside the tile,
            // no candidate that recomputes the stored value exactly, a repeat edge, an inconsistent source — is left to the
            // general code below, which decides the same cell again from scratch.
            if (in_tile(y, x)) {
                const uint32_t ly = y - ty0, lx = x - tx0, ci = ly * TT + lx;
                const uint32_t cy = T.p_cnt[ly], cx = T.p_cnt[TT + lx];
                const bool in_m = current_state == State_m, in_y = current_state == State_y;
                if ((in_m || in_y ? cy - 1u < (uint32_t)PGM_TB_PK : true) && (in_m || !in_y ? cx - 1u < (uint32_t)PGM_TB_PK : true)) {
                    const float4 c0 = T.cell[ci];
                    const float S0 = T.S[ci];
                    const uint32_t ka = in_m ? (uint32_t)lane >> 3 : (uint32_t)lane >> 1, kb = in_m ? (uint32_t)lane & 7u : (uint32_t)lane & 1u;
                    const bool valid = in_m ? (ka < cy && kb < cx) : (ka < (in_y ? cy : cx));
                    const uint32_t yi = ly * PGM_TB_PK + (ka & 7u), xi = (TT + lx) * PGM_TB_PK + ((in_m ? kb : ka) & 7u);
                    uint32_t yp = y, xp = x, uu = 0u;
                    float yv = 0.0f, xv = 0.0f;
                    if (in_m || in_y) { yp = T.p_c[yi]; yv = T.p_v[yi]; uu |= T.p_u[yi]; }
                    if (in_m || !in_y) { xp = T.p_c[xi]; xv = T.p_v[xi]; uu |= T.p_u[xi]; }
                    const bool inside = yp >= ty0 && xp >= tx0;
                    if (__builtin_amdgcn_ballot_w64(valid && !inside) == 0ull) {
                        const float4 c = T.cell[valid ? (yp - ty0) * TT + (xp - tx0) : ci];
                        float d;
                        if (in_m) d = fabsf(__fsub_rn(c0.x, __fsub_rn(__fsub_rn(__fadd_rn(c.z, S0), yv), xv)));
                        else if (in_y) d = fabsf(__fsub_rn(c0.w, __fsub_rn(__fadd_rn(kb == 0u ? c.w : c.z, kb == 0u ? s.gap_extend : s.gap_init), yv)));
                        else d = fabsf(__fsub_rn(c0.y, __fsub_rn(__fadd_rn(kb == 0u ? c.y : c.z, kb == 0u ? s.gap_extend : s.gap_init), xv)));
                        const unsigned long long zero = __builtin_amdgcn_ballot_w64(valid && d == 0.0f);
                        if (zero != 0ull) {
                            const int win = __ffsll((long long)zero) - 1;
                            const uint32_t w_yp = rl_u(yp, win), w_xp = rl_u(xp, win), w_rep = rl_u(uu, win), w_kind = rl_u(kb, win);
                            const float wz = rl_f(c.z, win), wm = rl_f(c.x, win), wx = rl_f(c.y, win), wy = rl_f(c.w, win);
                            int next_state = current_state;
                            bool ok = w_rep == 0u;
                            if ((in_m || w_kind == 1u) && (w_yp | w_xp) != 0u) {   // a W source: its state by equality in the order M, Y, X
                                if (wz == wm) next_state = State_m;
                                else if (wz == wy) next_state = State_y;
                                else if (wz == wx) next_state = State_x;
                                else ok = false;
                            }
                            if (ok) {
                                y = w_yp; x = w_xp;
                                current_state = next_state;
                                score_stale = true;
                                if (x != 0 || y != 0) {
                                    if (lane == 0 && mo.len < mo.cap) {
                                        mo.m1[mo.len] = current_state == State_x ? 0xFFFFFFFFu : y;
                                        mo.m2[mo.len] = current_state == State_y ? 0xFFFFFFFFu : x;
                                    }
                                    ++mo.len;
                                }
                                continue;
                            }
                        }
                    }
                }
            }
            ++st_slow;
            if (score_stale) {   // the score of (cell, state) is the cell's value for that state
                const float4 c = cell_at(y, x);
                current_score = current_state == State_m ? c.x : (current_state == State_x ? c.y : c.w);
                score_stale = false;
            }
            const bool need_y = current_state != State_x, need_x = current_state != State_y;
            PgmPredView Vy = {1u, false, 0u}, Vx = {1u, false, 0u};
            if (need_y) Vy = preds(true, y);
            if (need_x) Vx = preds(false, x);
            const float S = (current_state == State_m) ? s_at(y, x) : 0.0f;
            // candidates: state M: (ky, kx) pairs; state Y / X: (k, extend | open)
            const uint32_t per = (current_state == State_m) ? Vx.cnt : 2u;
            // state M with at most 8 x 8 pairs: lane = ky * 8 + kx (one chunk, lane order = PredIterator order, no division)
            const bool small = Vy.cnt <= 8u && Vx.cnt <= 8u;
            const uint32_t total = (current_state == State_m) ? (small ? (Vy.cnt && Vx.cnt ? 64u : 0u) : Vy.cnt * Vx.cnt) : 2u * (need_y ? Vy.cnt : Vx.cnt);
            float best = INFINITY;
            uint32_t w_yp = 0xFFFFFFFFu, w_xp = 0xFFFFFFFFu, w_uy = 0, w_ux = 0, w_kind = 0;
            float4 w_c = make_float4(0.f, 0.f, 0.f, 0.f);
            bool found = false;
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t i = base + (uint32_t)lane;
                const bool valid = (current_state == State_m && small) ? (((uint32_t)lane >> 3) < Vy.cnt && ((uint32_t)lane & 7u) < Vx.cnt) : i < total;
                uint32_t ka, kb;   // (ky, kx) or (k, kind)
                if (current_state != State_m) { ka = i >> 1; kb = i & 1u; }
                else if (small) { ka = (uint32_t)lane >> 3; kb = (uint32_t)lane & 7u; }
                else { ka = valid ? i / per : 0u; kb = valid ? i % per : 0u; }
                uint32_t yp = y, xp = x, uy = 0, ux = 0;
                float yv = 0.0f, xv = 0.0f;
                if (need_y) { const uint32_t k = ka; if (valid) { yp = pc_of(true, Vy, k); yv = pv_of(true, Vy, k); uy = pu_of(true, Vy, k); } }
                if (need_x) { const uint32_t k = (current_state == State_m) ? kb : ka; if (valid) { xp = pc_of(false, Vx, k); xv = pv_of(false, Vx, k); ux = pu_of(false, Vx, k); } }
                float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) c = cell_at(yp, xp);
                float d;
                if (current_state == State_m) d = fabsf(__fsub_rn(current_score, __fsub_rn(__fsub_rn(__fadd_rn(c.z, S), yv), xv)));
                else if (current_state == State_y) d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(kb == 0 ? c.w : c.z, kb == 0 ? s.gap_extend : s.gap_init), yv)));
                else d = fabsf(__fsub_rn(current_score, __fsub_rn(__fadd_rn(kb == 0 ? c.y : c.z, kb == 0 ? s.gap_extend : s.gap_init), xv)));
                if (!valid) d = INFINITY;
                // first lane with the smallest d, strictly better than what earlier chunks found
                int win = -1;
                const unsigned long long zero = __builtin_amdgcn_ballot_w64(valid && d == 0.0f);
                if (zero != 0ull) {
                    if (best > 0.0f) { win = __ffsll((long long)zero) - 1; best = 0.0f; }
                } else {
                    float m = best;
                    for (int l = 0; l < 64; ++l) { const float dl = rl_f(d, l); if (m > dl) { m = dl; win = l; } }
                    best = m;
                }
                if (win >= 0) {
                    found = true;
                    w_yp = rl_u(yp, win); w_xp = rl_u(xp, win); w_uy = rl_u(uy, win); w_ux = rl_u(ux, win); w_kind = rl_u(kb, win);
                    w_c = make_float4(rl_f(c.x, win), rl_f(c.y, win), rl_f(c.z, win), rl_f(c.w, win));
                }
                if (best == 0.0f) break;   // nothing later in the order can be strictly better
            }
            if (!found) { status = PGM_ERR_BACKTRACK; break; }
            // state of the chosen source (GraphAlign.h:400-411: a W source is resolved by equality in the order M, Y, X)
            const uint32_t next_y = w_yp, next_x = w_xp;
            int next_state = current_state;
            float next_score = INFINITY;
            bool tr_x = false, tr_y = false;
            const bool from_w = (current_state == State_m) || w_kind == 1u;
            if (current_state != State_x) tr_y = w_uy != 0;
            if (current_state != State_y) tr_x = w_ux != 0;
            if (!from_w) {
                next_score = (current_state == State_y) ? w_c.w : w_c.y;      // gap extension: stays in the gap state
            } else if (next_x != 0 || next_y != 0) {
                if (w_c.z == w_c.x) { next_score = w_c.x; next_state = State_m; }
                else if (w_c.z == w_c.w) { next_score = w_c.w; next_state = State_y; }
                else if (w_c.z == w_c.y) { next_score = w_c.y; next_state = State_x; }
                else { status = PGM_ERR_BACKTRACK; break; }
            }
            n_tr += (uint32_t)tr_x + (uint32_t)tr_y;
            if (tr_y || tr_x) {
                if (lane == 0) {
                    if (tr_y) pgm_mark_alternative_path(J, next_y, y, P1, mo, true);
                    if (tr_x) pgm_mark_alternative_path(J, next_x, x, P2, mo, false);
                }
                mo.len = __builtin_amdgcn_readfirstlane(mo.len);
            }
            x = next_x; y = next_y;
            current_state = next_state;
            current_score = next_score;
            if (x != 0 || y != 0) {
                uint32_t py = y, px = x;
                if (current_state == State_x) py = 0xFFFFFFFFu;
                else if (current_state == State_y) px = 0xFFFFFFFFu;
                if (lane == 0 && mo.len < mo.cap) { mo.m1[mo.len] = py; mo.m2[mo.len] = px; }
                ++mo.len;
            }
        }
        if (lane == 0 && mo.len < mo.cap) { mo.m1[mo.len] = 0u; mo.m2[mo.len] = 0u; }
        ++mo.len;
        if (mo.len > mo.cap) { status = PGM_ERR_BACKTRACK; mo.len = mo.cap; }
        if (lane == 0) {
            J.result->score = Wend;
            J.result->n_tr_indels = n_tr;
            J.result->len = mo.len;
            J.result->status = status;
            T.len = mo.len;
            T.req = 2;
            if (stat) { stat[0] = (st_stage << 32) | st_reload; stat[1] = (st_nreload << 32) | ((st_grid & 0xffffull) << 16) | (st_slow & 0xffffull); }
        }
        __syncthreads();                // "request" that ends the loaders' loop
    }
    // reverse the two mappings (GraphAlign.h:520-521), all threads; thread 0's pushes are ordered by the barrier
    __threadfence_block();
    __syncthreads();
    pgm_traceback_publish(J, T.len, tid, nthreads);
    if (tid == 0 && lq) {
        if (J.lrows != 0u) __hip_atomic_store(J.lready + J.lrows, (int)(J.lrows * PGM_LK_W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no tile of this job is wanted any more
        __hip_atomic_fetch_add(lq + (PGM_SY_TB_DONE - PGM_SY_LQ_N), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One grid tile of a job's corridor, all threads of a worker that has no band to sweep (pgm_fill_kernel's idle phase): the
// successor of every (cell, state) of the tile, decided as the walker of pgm_traceback_job decides it — same candidates in
// PredIterator order, same strict comparisons, W sources resolved by equality in the order M, Y, X — from the tile's cells
// plus a halo of PGM_LK_H rows / columns.  No link where the walker's general code has to look: a predecessor beyond the
// halo, more than PGM_TB_PK predecessors, a repeat edge, an inconsistent source.  The job is complete (its traceback has
// started), so every cell read here is final.
__device__ static void pgm_prelink_tile(const PgmJob &J, PgmLkLds &G, const uint32_t k, const int tid) {
    constexpr uint32_t TT = PGM_LK_T, HW = PGM_LK_HW, NT = 64u * PGM_WAVES, PK = PGM_TB_PK;
    const uint32_t n1 = J.n1, n2 = J.n2;
    const uint32_t w = min(J.lcols, PGM_LK_W), r = k / PGM_LK_W, jj = k % PGM_LK_W;
    if (jj >= w || r >= J.lrows) return;
    const uint32_t gx = pgm_lk_first(n1, n2, J.lcols, r) + jj;
    const uint32_t cy0 = TT * r, cx0 = TT * gx;
    const uint32_t hy0 = cy0 >= PGM_LK_H ? cy0 - PGM_LK_H : 0u, hx0 = cx0 >= PGM_LK_H ? cx0 - PGM_LK_H : 0u;
    const pgm_scores sc = J.sc;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // cells written by other XCDs
    __syncthreads();                                     // (the previous tile's table has been copied out)
    {
        // cells in anti-diagonal order (one anti-diagonal of the region is one contiguous run of the cell storage, see stage())
        constexpr int NC = (int)((HW * HW + NT - 1) / NT);
        float4 cv[NC];
        uint32_t cslot[NC];
#pragma unroll
        for (int u = 0; u < NC; ++u) {
            const uint32_t i = min((uint32_t)tid + NT * u, HW * HW - 1u);
            const uint32_t half = HW * (HW + 1u) / 2u;
            const bool up = i < half;
            const uint32_t m = up ? i : HW * HW - 1u - i;
            uint32_t d = (uint32_t)((sqrtf(8.0f * (float)m + 1.0f) - 1.0f) * 0.5f);
            while (d * (d + 1u) / 2u > m) --d;
            while ((d + 1u) * (d + 2u) / 2u <= m) ++d;
            const uint32_t kk = m - d * (d + 1u) / 2u;
            const uint32_t ly = up ? kk : HW - 1u - kk, lx = up ? d - kk : HW - 1u - (d - kk);
            cslot[u] = ly * HW + lx;
            cv[u] = J.cells[pgm_cell_index(J, min(hy0 + ly, n1 - 2), min(hx0 + lx, n2 - 2))];
        }
        float sv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t i = (uint32_t)tid + NT * u;
            sv[u] = pgm_emission_at(J, min(cy0 + i / TT, n1 - 2), min(cx0 + i % TT, n2 - 2));
        }
        const uint32_t slot = (uint32_t)tid / PK, kk = (uint32_t)tid % PK;
        const bool row = slot < TT;
        const uint32_t v = row ? cy0 + slot : cx0 + (slot - TT);
        const PgmTbNode *rec = (row ? J.tb1 : J.tb2) + min(v, (row ? n1 : n2) - 1);
        const uint32_t cnt = rec->cnt, ec = rec->c[kk], eu = rec->u[kk];
        const float ev = rec->v[kk];
#pragma unroll
        for (int u = 0; u < NC; ++u)
            if ((uint32_t)tid + NT * u < HW * HW) G.cell[cslot[u]] = cv[u];
#pragma unroll
        for (int u = 0; u < 2; ++u) G.S[(uint32_t)tid + NT * u] = sv[u];
        if (kk == 0) G.p_cnt[slot] = v < (row ? n1 : n2) ? cnt : 0xFFFFFFFFu;
        G.p_c[tid] = ec; G.p_v[tid] = ev; G.p_u[tid] = eu;
        if (tid == 0) G.nbig = 0;
    }
    __syncthreads();
    auto hc = [&](uint32_t yy, uint32_t xx) { return min((yy - hy0) * HW + (xx - hx0), HW * HW - 1u); };
    auto pick = [&](const float4 &c, uint32_t y, uint32_t x, uint32_t yp, uint32_t xp) -> uint16_t {
        uint32_t st;
        if ((yp | xp) == 0u) st = 0u;
        else if (c.z == c.x) st = 0u;
        else if (c.z == c.w) st = 2u;
        else if (c.z == c.y) st = 1u;
        else return (uint16_t)0;
        return pgm_lk_code(st, y - yp, x - xp);
    };
    for (uint32_t task = (uint32_t)tid; task < PGM_LK_TAB; task += NT) {
        const uint32_t st_task = task / (TT * TT), ci = task % (TT * TT), ly = ci / TT, lx = ci % TT, y = cy0 + ly, x = cx0 + lx;
        uint16_t link = 0;
        if (y + 1 < n1 && x + 1 < n2 && (y | x) != 0u) {
            const float4 c0 = G.cell[hc(y, x)];
            const uint32_t cy = G.p_cnt[ly], cx = G.p_cnt[TT + lx];
            if (st_task == 0u && cy - 1u < PK && cx - 1u < PK && c0.x > PGM_NEG_INF) {
                if (cy > PGM_TB_LK || cx > PGM_TB_LK) {   // 5..8 predecessors: a wavefront per cell (below)
                    const int slot = __hip_atomic_fetch_add(&G.nbig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    G.big[slot] = (uint16_t)ci;
                } else {
                    const float S = G.S[ci];
                    uint32_t yp[PGM_TB_LK], xp[PGM_TB_LK];
                    float yv[PGM_TB_LK], xv[PGM_TB_LK];
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < PGM_TB_LK; ++q) {
                        yp[q] = G.p_c[ly * PK + q]; yv[q] = G.p_v[ly * PK + q];
                        xp[q] = G.p_c[(TT + lx) * PK + q]; xv[q] = G.p_v[(TT + lx) * PK + q];
                        ok = ok && ((uint32_t)q >= cy || yp[q] >= hy0) && ((uint32_t)q >= cx || xp[q] >= hx0);
                    }
                    float wz[PGM_TB_LK][PGM_TB_LK];
#pragma unroll
                    for (int ky = 0; ky < PGM_TB_LK; ++ky)
#pragma unroll
                        for (int kx = 0; kx < PGM_TB_LK; ++kx) wz[ky][kx] = G.cell[hc(yp[ky], xp[kx])].z;
                    float best = INFINITY;
                    uint32_t wy = 0, wx = 0, wky = 0, wkx = 0;
#pragma unroll
                    for (int ky = 0; ky < PGM_TB_LK; ++ky)
#pragma unroll
                        for (int kx = 0; kx < PGM_TB_LK; ++kx) {
                            const float d = fabsf(__fsub_rn(c0.x, __fsub_rn(__fsub_rn(__fadd_rn(wz[ky][kx], S), yv[ky]), xv[kx])));
                            if ((uint32_t)ky < cy && (uint32_t)kx < cx && best > d) { best = d; wy = yp[ky]; wx = xp[kx]; wky = ky; wkx = kx; }
                        }
                    if (ok && best < INFINITY && (G.p_u[ly * PK + wky] | G.p_u[(TT + lx) * PK + wkx]) == 0u) link = pick(G.cell[hc(wy, wx)], y, x, wy, wx);
                }
            }
            if (st_task == 2u && cy - 1u < PK && c0.w > PGM_NEG_INF) {   // state Y: row predecessors, extension before opening
                float best = INFINITY;
                bool ok = true, rep_edge = false, open = false;
                uint32_t wy = 0;
                float4 wc = c0;
                for (uint32_t q = 0; q < cy; ++q) {
                    const uint32_t yp = G.p_c[ly * PK + q];
                    if (yp < hy0) { ok = false; break; }
                    const float yv = G.p_v[ly * PK + q];
                    const float4 c = G.cell[hc(yp, x)];
                    const bool ru = G.p_u[ly * PK + q] != 0u;
                    float d = fabsf(__fsub_rn(c0.w, __fsub_rn(__fadd_rn(c.w, sc.gap_extend), yv)));
                    if (best > d) { best = d; wy = yp; wc = c; rep_edge = ru; open = false; }
                    d = fabsf(__fsub_rn(c0.w, __fsub_rn(__fadd_rn(c.z, sc.gap_init), yv)));
                    if (best > d) { best = d; wy = yp; wc = c; rep_edge = ru; open = true; }
                }
                if (ok && best < INFINITY && !rep_edge) link = open ? pick(wc, y, x, wy, x) : pgm_lk_code(2u, y - wy, 0u);
            }
            if (st_task == 1u && cx - 1u < PK && c0.y > PGM_NEG_INF) {   // state X: column predecessors
                float best = INFINITY;
                bool ok = true, rep_edge = false, open = false;
                uint32_t wx = 0;
                float4 wc = c0;
                for (uint32_t q = 0; q < cx; ++q) {
                    const uint32_t xp = G.p_c[(TT + lx) * PK + q];
                    if (xp < hx0) { ok = false; break; }
                    const float xv = G.p_v[(TT + lx) * PK + q];
                    const float4 c = G.cell[hc(y, xp)];
                    const bool ru = G.p_u[(TT + lx) * PK + q] != 0u;
                    float d = fabsf(__fsub_rn(c0.y, __fsub_rn(__fadd_rn(c.y, sc.gap_extend), xv)));
                    if (best > d) { best = d; wx = xp; wc = c; rep_edge = ru; open = false; }
                    d = fabsf(__fsub_rn(c0.y, __fsub_rn(__fadd_rn(c.z, sc.gap_init), xv)));
                    if (best > d) { best = d; wx = xp; wc = c; rep_edge = ru; open = true; }
                }
                if (ok && best < INFINITY && !rep_edge) link = open ? pick(wc, y, x, y, wx) : pgm_lk_code(1u, 0u, x - wx);
            }
        }
        G.tab[task] = link;
    }
    __syncthreads();
    {   // M links of the cells whose nodes have 5..8 predecessors: one wavefront per cell, lane = 8 ky + kx (PredIterator order)
        const int wave = tid >> 6, lane_b = tid & 63;
        const int nbig = G.nbig;
        for (int i = wave; i < nbig; i += PGM_WAVES) {
            const uint32_t ci = G.big[i], ly = ci / TT, lx = ci % TT, y = cy0 + ly, x = cx0 + lx;
            const uint32_t cy = G.p_cnt[ly], cx = G.p_cnt[TT + lx];
            const uint32_t ky = (uint32_t)lane_b >> 3, kx = (uint32_t)lane_b & 7u;
            const bool valid = ky < cy && kx < cx;
            const uint32_t yp = G.p_c[ly * PK + ky], xp = G.p_c[(TT + lx) * PK + kx];
            const float yv = G.p_v[ly * PK + ky], xv = G.p_v[(TT + lx) * PK + kx];
            const uint32_t rep = G.p_u[ly * PK + ky] | G.p_u[(TT + lx) * PK + kx];
            const bool inside = yp >= hy0 && xp >= hx0;
            if (__builtin_amdgcn_ballot_w64(valid && !inside) != 0ull) continue;
            const float4 c0 = G.cell[hc(y, x)];
            const float S = G.S[ci];
            const float4 c = G.cell[(valid && inside) ? hc(yp, xp) : hc(y, x)];
            const float d = fabsf(__fsub_rn(c0.x, __fsub_rn(__fsub_rn(__fadd_rn(c.z, S), yv), xv)));
            const unsigned long long zero = __builtin_amdgcn_ballot_w64(valid && d == 0.0f);
            if (zero == 0ull) continue;   // (no exact candidate: the walker searches the minimum itself)
            const int win = __ffsll((long long)zero) - 1;
            if (lane_b == win && rep == 0u) G.tab[ci] = pick(c, y, x, yp, xp);
        }
    }
    __syncthreads();
    // the table -> memory, then the flag
    uint16_t *dst = J.ltab + (size_t)k * PGM_LK_TAB;
    for (uint32_t i = (uint32_t)tid; i < PGM_LK_TAB * 2u / 16u; i += NT) ((uint4 *)dst)[i] = ((const uint4 *)G.tab)[i];
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(J.lready + r, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // tiles of grid row r complete
}

#define PGM_SPIN_LIMIT (1u << 24)
#define PGM_IDLE_LIMIT_TICKS 400000000ull   /* 4 s of the 100 MHz real-time counter: a worker that has seen nothing happen for that long raises the abort flag */

// A job's last band is complete (and every store of it waited for): append the job to the ready queue of the traceback kernel that
// runs beside the sweeps (tbq_off == 0: the tracebacks follow on the stream instead).
__device__ __forceinline__ void pgm_tbq_push(int *sync, const uint32_t tbq_off, const uint32_t job) {
    if (tbq_off == 0u) return;
    const int k = __hip_atomic_fetch_add(sync + PGM_SY_TBQ_TAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sync + tbq_off + k, (int)job + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Fill kernel (GraphAlign.h:212-260 incl. the border initialisation as row/column 0).
//
// Work unit = one BAND of one job: 64 consecutive rows of graph 1 against all columns of graph 2, swept by ONE
// wavefront (no barrier inside a sweep; MODE 2 adds seven helper wavefronts that only meet it through LDS words).  Lane l
// owns row y = 64 b + l and at step t computes column x = t - l: the cells of one step are independent of each other,
// every predecessor cell was produced by a lower or the same lane at an earlier step.  An item of the work list is up to
// eight consecutive bands of one job, one per wavefront of the worker (512-thread workgroup, one per CU) that takes it,
// or one band of a MODE 2 job; all bands of all jobs of the batch sit in
// one list ordered by the host (longest remaining path first), a persistent grid of workers takes them through an
// atomic ticket.  Band b of a job is always listed after band b-1, so the wavefront it waits for is already running.
//
// Predecessors of a cell (y, x) are pairs (y - dy, x - dx) of a predecessor of node y and one of node x.
//   NEAR (dy, dx in 1..3; 90 % of all skip edges of a merged graph span 2 or 3 nodes): a systolic register window.
//       W(y-1, x) is what lane l-1 produced in the previous step (one DPP wave_shr:1); W(y-2, x) is what lane l-1
//       received as ITS upper neighbour one step ago, W(y-3, x) what it received as its second one: three DPP shifts
//       per step keep, in every lane, the last four columns of the three rows above (u1W, u2W, u3W), the lane's own
//       last three columns (ow, ox) and Y of the three rows above.  All 9 + 3 + 3 near terms are then evaluated
//       unconditionally with the edge costs {cc, c2, c3} of the row (registers) and of the column (one ds_read_b128
//       from a ring of column summaries); an absent edge has cost +inf and contributes -inf.  No LDS round trip, no
//       divergence, no loop.
//   FAR (any other edge, up to PGM_KF per node, distance <= PGM_DCAP): served from an LDS history of the band's
//       W, Y, X of the last hD steps (hW[step][16 + lane]): (y - dy, x - dx) was produced by lane l - dy at step
//       t - dy - dx.  The far code runs only in steps in which some lane has a far edge (wave-uniform branch).
//   Rows above the band: the last 16 rows of band b-1 are "virtual lanes" -16..-1 of the history (read back from the
//       cell storage a block ahead: because of the skew they are ONE contiguous 256 B run per step); lane 0's three
//       upper neighbours are injected from there (the DPP's `old` operand).
//   GENERIC (self-contained sweeps: more than PGM_KF far edges, farther than PGM_DCAP or above the virtual lanes; MODE 2:
//       what the helpers' long / remote entries and the overflow table do not cover, see PgmNode2): the node's edges are
//       read from the cell storage through the CSR lists (device-scope loads), as rare as pathological graphs are.
// Bands hand over through the cell storage itself; prog[b] (global, agent scope) = number of steps of band b that are
// complete AND visible; the producer publishes behind a counted s_waitcnt, never vmcnt(0).
// MODE 0: chain-only job (every node has at most its chain predecessor): chain terms only, no history of the band's own rows
// MODE 1: merged graphs, self-contained: near window + far history + generic path in this wavefront
// MODE 2: merged graphs on the batch's critical path or with long edges (PgmJob::mode2): this wavefront only evaluates the
//         terms it holds in its register windows (chain terms, X from columns x-2 / x-3, Y from row y-2, two M pairs), merges
//         the maxima seven helper wavefronts have folded ahead of it (pgm_terms_helper), stores the cell and records W, Y, X
//         in the history
// DBG = false (production): no timeline accumulators, no experiment switches (their branches and scalar registers are gone)
template <int MODE, bool DBG>
__device__ __forceinline__ void pgm_sweep_band(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *abort_flag, bool &aborted,
                                               const uint32_t spin_limit, const bool stall, unsigned long long *wait_acc_, int *sw_generic, const uint32_t dbg_flags_) {
    constexpr bool NEAR = MODE == 1, HELPED = MODE == 2, EXTRAS = MODE != 0;
    unsigned long long *const wait_acc = DBG ? wait_acc_ : nullptr;
    const uint32_t dbg_flags = DBG ? dbg_flags_ : 0u;
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING, KF = PGM_KF;
    constexpr int RS = HELPED ? 5 : 3;   // float4 per column of the ring: {q0, fd0-3, fc0-3}, or the whole node summary for the far helper
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;   // LDS pointer: a generic one would make every access a flat one (waits for vmcnt too)
    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nb = J.nb, nblk = J.nblk;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap, s_init = J.sc.start_init;
    const uint32_t D = J.hD, Dm = D - 1u, DX = J.hDX, DXm = DX - 1u;
    const bool has_far = EXTRAS && J.has_far != 0;
    const bool record = HELPED || has_far;   // the band's own W, Y, X go into the history
    // LDS of this sweep: W / Y history [D][80] (columns 0..15: virtual lanes), X history [DX][64], column summaries [128][3]
    float *hW = (float *)slot, *hY = hW + D * HS, *hX = hY + D * HS;
    float4 *ring3 = (float4 *)(hX + DX * 64u);
    const uint32_t y = 64u * b + (uint32_t)lane;
    const bool rowvalid = y + 1 < n1;
    const uint32_t yc = rowvalid ? y : 0u;
    const float4 *niq = (const float4 *)(J.ni1 + yc);
    const float4 r0 = pgm_gload4(niq), r1 = pgm_gload4(niq + 1), r2 = pgm_gload4(niq + 3);
    const float ccy = r0.x;
    const uint32_t fy = rowvalid ? __float_as_uint(r0.w) : 0u;
    const bool geny = EXTRAS && ((fy & PGM_NF_GENERIC) != 0 || ((fy & PGM_NF_COUNT) != 0 && ((fy >> 8) & 255u) > (uint32_t)(lane + VL)));
    const bool ykill = (fy & PGM_NF_KILL) != 0;
    const float c2y = (EXTRAS && !geny) ? r0.y : INFINITY, c3y = (NEAR && !geny) ? r0.z : INFINITY;
    uint32_t fdy[KF];
    float fcy[KF];
    {
        const float dsrc[KF] = {r1.x, r1.y, r1.z, r1.w}, csrc[KF] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            fdy[k] = (NEAR && rowvalid && !geny) ? __float_as_uint(dsrc[k]) : 0u;
            fcy[k] = (NEAR && rowvalid && !geny) ? csrc[k] : INFINITY;
        }
    }
    const int nfyw = NEAR ? pgm_wave_max8(geny ? 0u : (fy & PGM_NF_COUNT)) : 0;
    const uint32_t xby = (uint32_t)J.xp1[yc], xey = (uint32_t)J.xp1[yc + 1];
    const float gopen_x = (rowvalid && y == 0) ? sg : gi;
    const uint32_t ncol_row = rowvalid ? ncol : 0u;
    const int x_init = (rowvalid && y == 0) ? 0 : -0x40000000;   // W(0,0) = start_init: only row 0 ever sees xs == x_init
    const bool has_next = (b + 1 < nb), has_prev = (b > 0);
    float4 *cells_band = J.cells + (size_t)b * tsteps * 64u;
    const __amdgpu_buffer_rsrc_t cells_rsrc = pgm_band_rsrc(cells_band, tsteps * 1024u);
    const float4 *cells_prev = J.cells + (size_t)(b - 1) * tsteps * 64u;
    const float4 *S_band = (const float4 *)(J.S + (size_t)b * nblk * 64u * BL);
    const float4 *ni2q = (const float4 *)J.ni2;
    const uint32_t lb = (uint32_t)(VL + lane);
    float *res = (float *)(slot + J.aux_off + PGM_AUX_RES);
    // HELPED: the helper wavefronts whose terms this sweep waits for (bit h = wavefront h of the worker): 1, 2 the near terms,
    // 3, 4 the far edges of the columns, 5-7 the far edges of the rows (experiment switches: 32 without the former, 16 the latter)
    const uint32_t hmask = HELPED ? ((((dbg_flags & 32u) ? 0u : 0x06u) | ((has_far && !(dbg_flags & 16u)) ? 0xf8u : 0u)) & ~(dbg_flags >> 16)) : 0u;   // (bits 17-23 of the switches: do not wait for that wavefront)
    int seen_min = hmask ? 0 : 0x7fffffff;   // steps whose terms every helper has published

    for (uint32_t i = (uint32_t)lane; i < D * HS; i += 64u) { hW[i] = PGM_NEG_INF; hY[i] = PGM_NEG_INF; }
    for (uint32_t i = (uint32_t)lane; i < DX * 64u; i += 64u) hX[i] = PGM_NEG_INF;
    for (int i = lane; i < NR * RS; i += 64) ring3[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // "column < 0" slots

    // ---- block prefetch (global -> registers a block ahead -> LDS / registers of the block) ------------------------
    float4 pfq, pfs[BL / 4];
    float2 pfr[2];
    const int rq_col = lane / RS, rq_part = lane % RS;   // lanes 0..8 RS - 1: one float4 of the 8 column summaries of a block
    const int rq_quad = (RS == 3 && rq_part == 2) ? 3 : rq_part;   // which float4 of the 5 of a node summary
    auto load_ring_block = [&](uint32_t c0) {
        const uint32_t col = c0 + (uint32_t)rq_col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < RS * BL && col <= ncol) v = pgm_gload4(ni2q + 5u * col + (uint32_t)rq_quad);
        pfq = v;
    };
    auto store_ring_block = [&](uint32_t c0) {
        if (lane < RS * BL) ring3[(uint32_t)rq_part * (uint32_t)NR + ((c0 + (uint32_t)rq_col) & (uint32_t)(NR - 1))] = pfq;
    };
    auto load_s_block = [&](uint32_t s0) {
        const uint32_t tb = min(s0 / BL, nblk - 1u);   // (a block beyond the sweep is never consumed: any valid address will do)
#pragma unroll
        for (int q = 0; q < BL / 4; ++q) pfs[q] = pgm_gload4(S_band + ((size_t)tb * 64u + (uint32_t)lane) * (BL / 4) + q);
    };
    // virtual lanes of my steps s0 .. s0+7: virtual lane v (= lane v - 16, i.e. row 64 b - 16 + v) at my step s is the cell
    // lane 48 + v of band b-1 produced at ITS step s + 64 (column s + 16 - v); cells outside the matrix read as -inf
    auto load_rep_block = [&](int s0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane * 2 + u, s = s0 + e / VL, v = e % VL;
            const int col = s + (VL - v);
            float2 val = make_float2(PGM_NEG_INF, PGM_NEG_INF);
            if (has_prev && col >= 0 && col < (int)ncol) val = pgm_gload_cell_wy(cells_prev + (size_t)(s + 64) * 64u + (uint32_t)(64 - VL + v));
            pfr[u] = val;
        }
    };
    auto store_rep_block = [&](int s0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane * 2 + u, s = s0 + e / VL, v = e % VL;
            const uint32_t idx = ((uint32_t)s & Dm) * HS + (uint32_t)v;
            hW[idx] = pfr[u].x;
            hY[idx] = pfr[u].y;
        }
    };
    // HELPED: the same in half blocks of four steps, one entry per lane (requested four steps before use: see the sweep)
    auto load_rep_half = [&](int s0) {
        const int s = s0 + lane / VL, v = lane % VL;
        const int col = s + (VL - v);
        float2 val = make_float2(PGM_NEG_INF, PGM_NEG_INF);
        if (has_prev && col >= 0 && col < (int)ncol) val = pgm_gload_cell_wy(cells_prev + (size_t)(s + 64) * 64u + (uint32_t)(64 - VL + v));
        pfr[0] = val;
    };
    auto store_rep_half = [&](int s0) {
        const int s = s0 + lane / VL, v = lane % VL;
        const uint32_t idx = ((uint32_t)s & Dm) * HS + (uint32_t)v;
        hW[idx] = pfr[0].x;
        hY[idx] = pfr[0].y;
    };
    int seen = has_prev ? 0 : 0x7fffffff, pend = 0;
    auto wait_prev = [&](uint32_t steps_needed) {   // band b-1 has completed (and made visible) that many steps
        if (seen != 0x7fffffff && !aborted) {
            const int need = (int)min(steps_needed, tsteps);
            uint32_t spins = 0;
            const unsigned long long w0 = (wait_acc && seen < need) ? __builtin_amdgcn_s_memrealtime() : 0ull;
            while (seen < need) {
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)&J.prog[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= need) break;
                __builtin_amdgcn_s_sleep(4);
                if (++spins > spin_limit || __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    aborted = true;
                    break;
                }
            }
            if (w0) wait_acc[0] += __builtin_amdgcn_s_memrealtime() - w0;
        }
    };
    // split-phase poll: the progress word is read one block ahead (no round trip on the band's own critical path);
    // wait_prev only spins when that value is not far enough, i.e. when this band really has to wait for band b-1
    auto poll_issue = [&]() { if (seen != 0x7fffffff) pend = __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)&J.prog[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto poll_collect = [&]() { if (seen != 0x7fffffff) seen = max(seen, __builtin_amdgcn_readfirstlane(pend)); };

    // prologue: column summaries of blocks 0 and 1 straight in, block 2 in flight; scores of block 0 in registers, block 1
    // in flight; virtual lanes of the steps before 0 (they reach back min(16, D - 8) steps) and of block 0 in, block 1 in flight
    float Sc[BL];
    load_ring_block(0); store_ring_block(0);
    load_ring_block(BL); store_ring_block(BL);
    load_s_block(0);
#pragma unroll
    for (int q = 0; q < BL / 4; ++q) { Sc[4 * q] = pfs[q].x; Sc[4 * q + 1] = pfs[q].y; Sc[4 * q + 2] = pfs[q].z; Sc[4 * q + 3] = pfs[q].w; }
    load_s_block(BL);
    wait_prev(HELPED ? BL + 64 : BL + 64 + BL);   // (HELPED: virtual lanes in half blocks, stored just in time: steps 0-3 now, 4-7 requested)
    if (has_prev) {
        if (D >= (uint32_t)(VL + BL)) { load_rep_block(-2 * BL); store_rep_block(-2 * BL); }
        load_rep_block(-BL); store_rep_block(-BL);
        if (HELPED) { load_rep_half(0); store_rep_half(0); }
        else { load_rep_block(0); store_rep_block(0); }
    }
    load_ring_block(2 * BL);
    if (HELPED) load_rep_half(BL / 2); else load_rep_block(BL);
    poll_issue();
    if (HELPED) {   // history initialised, first blocks staged: the helpers may start (sw[0] = last recorded step + 2)
        for (int i = lane; i < 4 * 192; i += 64) res[i] = PGM_NEG_INF;
        if (lane >= 1 && lane < PGM_WAVES && !(hmask & (1u << lane))) sw[lane] = 0x7fffffff;   // wavefronts that publish nothing for this item
        asm volatile("" ::: "memory");
        __hip_atomic_store(sw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // ---- the sweep ---------------------------------------------------------------------------------------------------
    float u1W[4], u2W[4], u3W[4], ow[4], ox[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { u1W[k] = PGM_NEG_INF; u2W[k] = PGM_NEG_INF; u3W[k] = PGM_NEG_INF; ow[k] = PGM_NEG_INF; ox[k] = PGM_NEG_INF; }
    float u1Y = PGM_NEG_INF, u2Y = PGM_NEG_INF, u3Y = PGM_NEG_INF, W_o = PGM_NEG_INF, Y_o = PGM_NEG_INF;
    // LDS operands of a step are read one step ahead: the column summary and lane 0's upper neighbours (virtual lanes of the
    // history; all lanes read the same word, only lane 0 keeps it: `old` operand of the DPP shift)
    typedef int pgm_v4i __attribute__((ext_vector_type(4)));
    pgm_v4i hw_a = {0, 0, 0, 0}, hw_b = {0, 0, 0, 0};   // HELPED: the helpers' counters as read a step ago
    float4 cn_n = ring3[((uint32_t)(-lane)) & (uint32_t)(NR - 1)];
    float inW1 = hW[(0xFFFFFFFFu & Dm) * HS + VL - 1], inY1 = hY[(0xFFFFFFFFu & Dm) * HS + VL - 1];
    float inW2 = hW[(0xFFFFFFFEu & Dm) * HS + VL - 2], inY2 = hY[(0xFFFFFFFEu & Dm) * HS + VL - 2];
    float inW3 = hW[(0xFFFFFFFDu & Dm) * HS + VL - 3], inY3 = hY[(0xFFFFFFFDu & Dm) * HS + VL - 3];
    for (uint32_t t0 = 0; !aborted; t0 += BL) {
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const uint32_t t = t0 + i;
            const int xs = (int)t - lane;
            // one unsigned compare per predicate, no mask arithmetic (a VALU compare combined on the scalar unit and fed back into a
            // VALU select stalls the in-order wavefront twice): ncol_row = ncol for a valid row, 0 for the rows below the matrix
            const bool active = (uint32_t)xs < ncol_row;
            const uint32_t x = (uint32_t)xs;
            const uint32_t rslot = x & (uint32_t)(NR - 1);   // column summaries: part-major, part k of column c at ring3[k * NR + (c & (NR - 1))]
            const float4 cn = cn_n;
            const float iW1 = inW1, iY1 = inY1, iW2 = inW2, iY2 = inY2, iW3 = inW3, iY3 = inY3;
            if (HELPED && !(dbg_flags & 2u)) {
                // The helper wavefronts run ahead.  Their seven counters (words of wavefronts that do not take part were set to
                // "far ahead" in the prologue) are read with the look-ahead operands of the PREVIOUS step — two 128-bit reads of the
                // same words by every lane, minimum on the vector unit — so normally this step only compares; it reads them
                // again (and waits) only if the helpers really are behind.
                const int want = (int)t + 1;
#if PGM_POLL_PREFETCH
                {
                    int m = min(min(hw_a.y, hw_a.z), min(hw_a.w, hw_b.x));
                    m = min(m, min(min(hw_b.y, hw_b.z), hw_b.w));
                    seen_min = max(seen_min, __builtin_amdgcn_readfirstlane(m));
                }
#endif
                if (__builtin_expect(seen_min < want, 0)) {
                    uint32_t spins = 0;
                    const unsigned long long h0 = wait_acc ? __builtin_amdgcn_s_memrealtime() : 0ull;
                    for (;;) {
                        const pgm_v4i wa = *(const __attribute__((address_space(3))) pgm_v4i *)sw, wb = *(const __attribute__((address_space(3))) pgm_v4i *)(sw + 4);
                        int m = min(min(wa.y, wa.z), min(wa.w, wb.x));
                        m = min(m, min(min(wb.y, wb.z), wb.w));
                        seen_min = __builtin_amdgcn_readfirstlane(m);
                        if (seen_min >= want) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
                    }
                    if (wait_acc) wait_acc[1] += __builtin_amdgcn_s_memrealtime() - h0;
                }
                asm volatile("" ::: "memory");
#if PGM_POLL_PREFETCH
                hw_a = *(const __attribute__((address_space(3))) pgm_v4i *)sw; hw_b = *(const __attribute__((address_space(3))) pgm_v4i *)(sw + 4);   // for step t + 1
#endif
            }
            if (!(dbg_flags & 64u)) {   // operands of step t + 1 (their virtual-lane entries and column summaries were staged at least a block ago)
                cn_n = ring3[(x + 1u) & (uint32_t)(NR - 1)];
                const uint32_t tm0 = (t & Dm) * HS, tm1 = ((t - 1u) & Dm) * HS, tm2 = ((t - 2u) & Dm) * HS;
                inW1 = hW[tm0 + VL - 1]; inY1 = hY[tm0 + VL - 1];
                if (EXTRAS) { inW2 = hW[tm1 + VL - 2]; inY2 = hY[tm1 + VL - 2]; }
                if (NEAR) { inW3 = hW[tm2 + VL - 3]; inY3 = hY[tm2 + VL - 3]; }
            }
            // maxima over the helpers' terms of this step: read with the other LDS operands, merged after the chain terms; the
            // words are reset for step t + 4 (no helper writes that step before this one is recorded: far_slack <= 4)
            float rM = PGM_NEG_INF, rX = PGM_NEG_INF, rY = PGM_NEG_INF;
            if (HELPED && !(dbg_flags & 2u)) {
                const uint32_t ro = (t & 3u) * 192u + (uint32_t)lane;
                rM = res[ro]; rX = res[ro + 64]; rY = res[ro + 128];
                res[ro] = PGM_NEG_INF; res[ro + 64] = PGM_NEG_INF; res[ro + 128] = PGM_NEG_INF;
            }
            const float ccx = cn.x, c2x = EXTRAS ? cn.y : INFINITY, c3x = EXTRAS ? cn.z : INFINITY;
            const uint32_t fx = __float_as_uint(cn.w);
            const bool genx = EXTRAS && active && (fx & PGM_NF_GENERIC) != 0;
            const bool xkill = (fx & PGM_NF_KILL) != 0;
            const float gopen_y = (xs == 0) ? sg : gi;
            const float S = Sc[i];
            const int s0 = i & 3, sm1 = (i + 3) & 3, sm2 = (i + 2) & 3, sm3 = (i + 1) & 3;
            if (NEAR) {
                const float n3W = pgm_dpp_wave_shr1(u2W[sm1], iW3), n3Y = pgm_dpp_wave_shr1(u2Y, iY3);
                u3W[s0] = n3W; u3Y = n3Y;
            }
            if (EXTRAS) {
                const float n2W = pgm_dpp_wave_shr1(u1W[sm1], iW2), n2Y = pgm_dpp_wave_shr1(u1Y, iY2);
                u2W[s0] = n2W; u2Y = n2Y;
            }
            if (!(dbg_flags & 128u)) {
                u1W[s0] = pgm_dpp_wave_shr1(W_o, iW1);
                u1Y = pgm_dpp_wave_shr1(Y_o, iY1);
            } else { u1W[s0] = W_o; u1Y = Y_o; }
            auto mterm = [&](float w, float cy, float cx) { return __fsub_rn(__fsub_rn(__fadd_rn(w, S), cy), cx); };
            auto xterm = [&](float xp, float wp, float cx) { return __fsub_rn(fmaxf(__fadd_rn(xp, ge), __fadd_rn(wp, gopen_x)), cx); };
            auto yterm = [&](float yp, float wp, float cy) { return __fsub_rn(fmaxf(__fadd_rn(yp, ge), __fadd_rn(wp, gopen_y)), cy); };
            float Mv = mterm(u1W[sm1], ccy, ccx);
            float Xv = xterm(ox[sm1], ow[sm1], ccx);
            float Yv = yterm(u1Y, u1W[s0], ccy);
            if (HELPED) {
                // the terms that read steps t - 2 and t - 3 from this wavefront's own register windows are evaluated here too (X from
                // columns x-2 and x-3, Y from row y-2, the M pairs (y-1, x-2) and (y-2, x-1)); every other term of this step comes from
                // the helper wavefronts: Y from row y-3 (history of step t - 3), the other near pairs and the far edges (step t - 4 and older)
                Xv = fmaxf(Xv, fmaxf(xterm(ox[sm2], ow[sm2], c2x), xterm(ox[sm3], ow[sm3], c3x)));
                Yv = fmaxf(Yv, yterm(u2Y, u2W[s0], c2y));
                Mv = fmaxf(Mv, fmaxf(mterm(u1W[sm2], ccy, c2x), mterm(u2W[sm1], c2y, ccx)));
            }
            if (HELPED) {
                Mv = fmaxf(Mv, rM);
                Xv = fmaxf(Xv, rX);
                Yv = fmaxf(Yv, rY);
            }
            if (NEAR) {
                Mv = fmaxf(Mv, fmaxf(mterm(u1W[sm2], ccy, c2x), mterm(u1W[sm3], ccy, c3x)));
                Mv = fmaxf(Mv, fmaxf(mterm(u2W[sm1], c2y, ccx), fmaxf(mterm(u2W[sm2], c2y, c2x), mterm(u2W[sm3], c2y, c3x))));
                Mv = fmaxf(Mv, fmaxf(mterm(u3W[sm1], c3y, ccx), fmaxf(mterm(u3W[sm2], c3y, c2x), mterm(u3W[sm3], c3y, c3x))));
                Xv = fmaxf(Xv, fmaxf(xterm(ox[sm2], ow[sm2], c2x), xterm(ox[sm3], ow[sm3], c3x)));
                Yv = fmaxf(Yv, fmaxf(yterm(u2Y, u2W[s0], c2y), yterm(u3Y, u3W[s0], c3y)));
                // ---- far edges: LDS history ----
                const uint32_t nfx = fx & PGM_NF_COUNT;
                if (__builtin_expect(nfyw != 0 || __builtin_amdgcn_ballot_w64(nfx != 0u) != 0ull, 0)) {
                    const float4 f1 = ring3[rslot + (uint32_t)NR], f2 = ring3[rslot + 2u * (uint32_t)NR];
                    const uint32_t fdx[KF] = {__float_as_uint(f1.x), __float_as_uint(f1.y), __float_as_uint(f1.z), __float_as_uint(f1.w)};
                    const float fcx[KF] = {f2.x, f2.y, f2.z, f2.w};
                    const int nfxw = pgm_wave_max8(nfx);
#pragma unroll
                    for (int j = 0; j < KF; ++j) {
                        if (j < nfxw) {
                            const uint32_t s1 = t - fdx[j];
                            const float Xh = hX[(s1 & DXm) * 64u + (uint32_t)lane], Wh = hW[(s1 & Dm) * HS + lb];
                            const float W1 = hW[((s1 - 1u) & Dm) * HS + lb - 1u], W2 = hW[((s1 - 2u) & Dm) * HS + lb - 2u], W3 = hW[((s1 - 3u) & Dm) * HS + lb - 3u];
                            Xv = fmaxf(Xv, xterm(Xh, Wh, fcx[j]));
                            Mv = fmaxf(Mv, fmaxf(mterm(W1, ccy, fcx[j]), fmaxf(mterm(W2, c2y, fcx[j]), mterm(W3, c3y, fcx[j]))));
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KF; ++k) {
                        if (k < nfyw) {
                            const uint32_t s1 = t - fdy[k], lp = lb - fdy[k];
                            const float Yh = hY[(s1 & Dm) * HS + lp], Wh = hW[(s1 & Dm) * HS + lp];
                            const float W1 = hW[((s1 - 1u) & Dm) * HS + lp], W2 = hW[((s1 - 2u) & Dm) * HS + lp], W3 = hW[((s1 - 3u) & Dm) * HS + lp];
                            Yv = fmaxf(Yv, yterm(Yh, Wh, fcy[k]));
                            Mv = fmaxf(Mv, fmaxf(mterm(W1, fcy[k], ccx), fmaxf(mterm(W2, fcy[k], c2x), mterm(W3, fcy[k], c3x))));
#pragma unroll
                            for (int j = 0; j < KF; ++j) {
                                if (j < nfxw) {
                                    const float Wp = hW[((s1 - fdx[j]) & Dm) * HS + lp];
                                    Mv = fmaxf(Mv, mterm(Wp, fcy[k], fcx[j]));
                                }
                            }
                        }
                    }
                }
            }
            if (EXTRAS) {
                // ---- generic nodes: every non-chain predecessor through the CSR lists and the cell storage ----
                const bool gen = active && (geny || genx);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(gen) != 0ull, 0)) {
                    if (gen) {
                        const uint32_t xbx = (uint32_t)pgm_gld(J.xp2 + x), xex = (uint32_t)pgm_gld(J.xp2 + x + 1);
                        for (uint32_t e = xby; e < xey; ++e) {
                            const uint32_t yp = pgm_gld(J.xc1 + e);
                            const float cy = pgm_gld(J.xv1 + e);
                            const float2 c = pgm_load_cell_wy(J.cells + pgm_cell_index(J, yp, x));
                            Yv = fmaxf(Yv, __fsub_rn(fmaxf(__fadd_rn(c.y, ge), __fadd_rn(c.x, gopen_y)), cy));
                            if (x > 0) {
                                const float2 c2 = pgm_load_cell_wy(J.cells + pgm_cell_index(J, yp, x - 1));
                                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c2.x, S), cy), ccx));
                            }
                            for (uint32_t f = xbx; f < xex; ++f) {
                                const uint32_t xp = pgm_gld(J.xc2 + f);
                                const float cx = pgm_gld(J.xv2 + f);
                                const float2 c3 = pgm_load_cell_wy(J.cells + pgm_cell_index(J, yp, xp));
                                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c3.x, S), cy), cx));
                            }
                        }
                        for (uint32_t f = xbx; f < xex; ++f) {
                            const uint32_t xp = pgm_gld(J.xc2 + f);
                            const float cx = pgm_gld(J.xv2 + f);
                            const float4 *cp = J.cells + pgm_cell_index(J, y, xp);
                            const float2 cm = pgm_load_cell_mx(cp), cw = pgm_load_cell_wy(cp);
                            Xv = fmaxf(Xv, __fsub_rn(fmaxf(__fadd_rn(cm.y, ge), __fadd_rn(cw.x, gopen_x)), cx));
                            if (y > 0) {
                                const float2 c2 = pgm_load_cell_wy(J.cells + pgm_cell_index(J, y - 1, xp));
                                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(c2.x, S), ccy), cx));
                            }
                        }
                    }
                }
            }
            if (ykill) Xv = PGM_NEG_INF;
            if (xkill) Yv = PGM_NEG_INF;
            float Wv = fmaxf(Mv, fmaxf(Xv, Yv));
            if (xs == x_init) Wv = s_init;
            if (!active) { Mv = PGM_NEG_INF; Xv = PGM_NEG_INF; Yv = PGM_NEG_INF; Wv = PGM_NEG_INF; }
            pgm_store_cell_masked(cells_rsrc, t, lane, active && !(dbg_flags & 1u), Mv, Xv, Wv, Yv);
            if (record && !(dbg_flags & 8u)) {
                const uint32_t ho = (t & Dm) * HS + lb;
                hW[ho] = Wv;
                hY[ho] = Yv;
                hX[(t & DXm) * 64u + (uint32_t)lane] = Xv;
                if (HELPED) {   // (LDS operations of a wavefront execute in order: the word follows the three stores)
                    asm volatile("" ::: "memory");
                    __hip_atomic_store(sw, (int)t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            W_o = Wv;
            Y_o = Yv;
            ow[s0] = Wv;
            ox[s0] = Xv;
            if (HELPED && i == BL / 2 - 1 && has_next && !stall) {
                // a band on the critical path publishes its progress twice per block (the band below follows 4 steps closer):
                // all but the last 4 stores — i.e. every step before this block — are complete
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BL / 2) : "memory");
                if (lane == 0) __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)&J.prog[b], (int)t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (HELPED && i == BL / 2 - 1) {
                // ... and handles the virtual lanes in half blocks: the four steps ahead are stored into the history now (their loads
                // had four steps to complete), the four after them are requested: the band needs the one above 72 steps ahead of
                // its next step, not 80 ahead of its next block
                store_rep_half((int)(t0 + BL / 2));
                poll_collect();
                wait_prev(t0 + BL + 64 + BL / 2);
                load_rep_half((int)(t0 + BL));
                poll_issue();
            }
        }
        if (has_next && !stall) {
            if (HELPED) {   // (since the first four stores of the block: the word store, two virtual-lane loads, the poll, four cell stores)
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BL / 2 + 1) : "memory");
                if (lane == 0) __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)&J.prog[b], (int)t0 + BL / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BL) : "memory");
                if (lane == 0) __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)&J.prog[b], (int)t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // Stage the next blocks.  This sits at the END of the iteration so that, on every path into it, exactly the BL
        // cell stores of this block were issued after the prefetch loads consumed here: the compiler then waits for them
        // with vmcnt(BL) instead of vmcnt(0), i.e. the wavefront never drains the stores it has just issued.
        // (the loop is left BEFORE the staging, so that the registers the prefetch loads write are the ones the next iteration
        // reads: a path around the staging would merge at the loop head and the merge copies would wait for the loads)
        const uint32_t t1 = t0 + BL;
        if (t1 >= tsteps) break;
        store_ring_block(t1 + BL);     // one block before use
#pragma unroll
        for (int q = 0; q < BL / 4; ++q) { Sc[4 * q] = pfs[q].x; Sc[4 * q + 1] = pfs[q].y; Sc[4 * q + 2] = pfs[q].z; Sc[4 * q + 3] = pfs[q].w; }
        if (HELPED) store_rep_half((int)t1); else store_rep_block((int)t1);
        load_ring_block(t1 + 2 * BL);
        load_s_block(t1 + BL);
        poll_collect();
        if (HELPED) {
            wait_prev(t1 + BL + 64);
            load_rep_half((int)(t1 + BL / 2));
        } else {
            wait_prev(t1 + BL + BL + 64);
            load_rep_block((int)(t1 + BL));
        }
        poll_issue();
    }
    if (HELPED) __hip_atomic_store(sw, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // releases the helpers (also after an abort)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && !stall) __hip_atomic_store(&J.prog[b], aborted ? (int)0 : (int)0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Lean sweep of a CHAIN-ONLY job (PgmJob::lean: both graphs are plain chains 0 -> 1 -> ... -> n-1 with finite edge costs — a
// sequence graph against a sequence graph: the leaf level of the guide tree, half of the cells of a progressive pass;
// GraphAlign.h:212-260 with one-entry predecessor lists).  Same recurrence, same float operations in the same order as
// pgm_sweep_band<0>; what differs is the schedule and what is kept:
//   * a lane owns R consecutive rows (a band = 64 R rows): only its first row takes W / Y of the row above from the lane above
//     (two DPP shifts per step), the others from the lane's own registers, and the column's edge cost travels down the lanes
//     with the column (one more shift) — no LDS on the step's path at all;
//   * ONE worker sweeps the whole job: wavefront w takes the bands w, w + 8, ..., and the last row of a band reaches the band
//     below through an LDS ring of {W, Y} per column (lane 63 writes it, lanes 0-7 of the consumer read a block of eight
//     columns) with two LDS counters per ring: columns produced / columns consumed (the producer never runs more than the ring
//     ahead of its consumer).  No hand-off through memory, no progress words, no virtual lanes: the band below follows
//     64 + 8 steps behind, whatever the memory system is doing;
//   * the traceback's decisions are taken HERE, where the operands are in registers: four bits per cell (PgmJob::codes)
//       bit 3 / 2  W equals M / W equals Y: the state of a walk that arrives at this cell through W is M, else Y, else X — the
//                  reference's order of tests (GraphAlign.h:400-411)
//       bit 1 / 0  the X / Y state of this cell came from an opening (W + gap_init), not from an extension: the reference's
//                  walk takes the extension unless the opening recomputes the stored value strictly better (:382-392,
//                  :421-431), i.e. — max(a, b) - c being max(a - c, b - c) exactly — unless (a - c) differs from the stored
//                  value.  (Row 0 / column 0 were filled with start_gap and are re-evaluated with gap_init by the walk; there
//                  either decision leads to the same next cell in the same state, so the bit is never looked at.)
//     and pgm_traceback_chain walks the codes.  The four float matrices go to memory only if the caller wants to read them
//     (KEEP: pgm_align_batch_create_ex with PGM_BATCH_KEEP_MATRICES, the test hook); the END node's one predecessor cell
//     (n1-2, n2-2) is always left in PgmJob::endcell.
// Cells go out as R coalesced 1 KB runs per step, cells[((b tsteps + t) R + r) 64 + lane] (PgmJob::rshift); codes as one
// 32-bit word per lane, row and block of eight steps (step i of the block in bits 31-4i .. 28-4i),
// codes[((b nblk + t / 8) R + r) 64 + lane].
#define PGM_LEAN_RING 512   // columns per ring (4 KB per wavefront)
// TAB: the scores are looked up in the job's class table (LDS, behind the rings: PGM_LEAN_TAB_OFF) instead of read from S — the
// class of the column travels down the lanes with the column's chain cost.
#define PGM_LEAN_TAB_OFF (8 * PGM_LEAN_RING * 8)
template <int R, bool KEEP, bool TAB = false>
__device__ __forceinline__ void pgm_sweep_chain(const PgmJob &J, const int wave, const int lane, uint8_t *pool, int *lsync_generic,
                                                int *abort_flag, bool &aborted, const uint32_t spin_limit) {
    constexpr int BL = PGM_BLOCK, RING = PGM_LEAN_RING;
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef float pgm_v2f __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) pgm_v2f pgm_lds_f2;
    typedef uint32_t pgm_v4u __attribute__((ext_vector_type(4)));
    pgm_lds_int *P = (pgm_lds_int *)lsync_generic, *Cn = P + 8;   // columns produced by wavefront w / consumed by wavefront w (cumulative over its bands)
    const int prod = (wave + 7) & 7, cons = (wave + 1) & 7;
    pgm_lds_f2 *ring_out = (pgm_lds_f2 *)pool + wave * RING, *ring_in = (pgm_lds_f2 *)pool + prod * RING;
    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nb = J.nb, nblk = J.nblk;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap, s_init = J.sc.start_init;
    const float4 *ni2q = (const float4 *)J.ni2;
    // the END node's predecessor cell (n1-2, n2-2): band, lane, row of the lane, step (all wave-uniform)
    const uint32_t ye = n1 - 2u, be = ye / (64u * R), le = (ye % (64u * R)) / R, re = ye % R, te = (ncol - 1u) + le;
    const __attribute__((address_space(3))) float *tab = (const __attribute__((address_space(3))) float *)(pool + PGM_LEAN_TAB_OFF);
    const uint32_t NC = J.dim + 2u;
    uint32_t q = 0;
    for (uint32_t b = (uint32_t)wave; b < nb && !aborted; b += PGM_WAVES, ++q) {
        const uint32_t y0 = (64u * b + (uint32_t)lane) * (uint32_t)R;
        float ccy[R], gox[R];
        uint32_t soff[R], trow[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t y = y0 + (uint32_t)r;
            const bool valid = y + 1 < n1;
            trow[r] = TAB ? (uint32_t)pgm_gld(J.cls1 + (valid ? y : 0u)) * NC : 0u;   // (rows beyond the graph: the START row's scores, as the emission kernel)
            ccy[r] = pgm_gload4((const float4 *)(J.ni1 + (valid ? y : 0u))).x;   // (+inf for START and for the rows below the matrix)
            gox[r] = (valid && y == 0) ? sg : gi;
            // (a row below the matrix: 0x40000000, a step outside the columns: 0x80000000 — any sum of the two lies beyond the
            //  band's num_records, which the host keeps below 1 GiB, and the buffer range check drops the store)
            soff[r] = valid ? (uint32_t)lane * 16u + (uint32_t)r * 1024u : 0x40000000u;
        }
        const bool origin = b == 0 && lane == 0;   // this lane's first row is row 0: W(0,0) = start_init
        const bool has_prev = b > 0, has_next = b + 1 < nb;
        const uint32_t pbase = q * ncol, cbase = (wave == 0 ? q - 1u : q) * ncol;
        const __amdgpu_buffer_rsrc_t rsrc = pgm_band_rsrc(J.cells + (KEEP ? (size_t)b * tsteps * (64u * R) : (size_t)0), KEEP ? tsteps * (1024u * R) : 16u);
        PGM_GLOBAL uint32_t *codes_band = (PGM_GLOBAL uint32_t *)(uintptr_t)(J.codes + (size_t)b * nblk * (64u * R)) + lane;
        const float4 *S_band = (const float4 *)J.S + (size_t)b * nblk * (uint32_t)(R * 64 * (BL / 4));
        // prefetch, a SUPER-BLOCK of 16 steps ahead (a block ahead does not cover the memory latency once the steps are this
        // short): emission scores (R x 16 per lane) and, in every lane l, the chain cost of column T0 + (l & 15) (column 0 has
        // no predecessor; its cost is never looked at with a finite left operand: 0 stands in)
        float4 pfs[R][2 * BL / 4];
        float pfc = 0.f, pfk = 0.f;   // (pfk: the class of the column, an integer in a float register like the cost it travels with)
        auto load_s_super = [&](uint32_t tb) {   // blocks tb, tb + 1
            if (TAB) return;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t tbh = min(tb + (uint32_t)h, nblk - 1u);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int k = 0; k < BL / 4; ++k) pfs[r][h * (BL / 4) + k] = pgm_gload4(S_band + ((size_t)(tbh * R + r) * 64u + (uint32_t)lane) * (BL / 4) + k);
            }
        };
        auto load_c_super = [&](uint32_t c0) {
            const uint32_t col = c0 + (uint32_t)(lane & 15);
            float v = 0.f;
            if (col != 0 && col < ncol) v = pgm_gload4(ni2q + 5u * col).x;
            pfc = v;
            if (TAB) pfk = __uint_as_float(col <= ncol ? (uint32_t)pgm_gld(J.cls2 + col) : NC - 1u);   // (columns beyond the graph: empty)
        };
        float Sc[R][2 * BL];
        auto take_super = [&]() {
            if (TAB) return;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < 2 * BL / 4; ++k) { Sc[r][4 * k] = pfs[r][k].x; Sc[r][4 * k + 1] = pfs[r][k].y; Sc[r][4 * k + 2] = pfs[r][k].z; Sc[r][4 * k + 3] = pfs[r][k].w; }
        };
        load_s_super(0);
        load_c_super(0);
        take_super();
        float cblk = pfc, kblk = pfk;
        load_s_super(2);
        load_c_super(2 * BL);
        float W_left[R], X_left[R];
        uint32_t cw[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { W_left[r] = PGM_NEG_INF; X_left[r] = PGM_NEG_INF; cw[r] = 0u; }
        float W_diag0 = PGM_NEG_INF, W_o = PGM_NEG_INF, Y_o = PGM_NEG_INF, ccx_o = 0.f, kc_o = __uint_as_float(NC - 1u);
        int p_seen = has_prev ? 0 : 0x7fffffff, c_seen = has_next ? 0 : 0x7fffffff;
        // one block of eight steps (hc: first / second half of the super-block)
        auto block = [&](const uint32_t t0, auto hc) {
            constexpr int H = decltype(hc)::value;
            // ---- the eight columns of the row above this band that lane 0 meets in this block (every lane l holds column t0 + (l & 7)) ----
            float inW = PGM_NEG_INF, inY = PGM_NEG_INF;
            if (has_prev) {
                const int need = (int)(cbase + min(t0 + (uint32_t)BL, ncol));
                uint32_t spins = 0;
                while (p_seen < need) {
                    p_seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(P + prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (p_seen >= need) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit || ((spins & 1023u) == 0u && __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = true;
                        break;
                    }
                }
                asm volatile("" ::: "memory");
                const uint32_t col = t0 + (uint32_t)(lane & 7);
                const pgm_v2f v = ring_in[(cbase + min(col, ncol - 1u)) & (uint32_t)(RING - 1)];
                if (col < ncol) { inW = v.x; inY = v.y; }
                asm volatile("" ::: "memory");
                // (LDS operations of a wavefront execute in order: the producer sees the counter after the reads above were served)
                __hip_atomic_store(Cn + wave, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (has_next && t0 + (uint32_t)BL > 63u) {
                // lane 63 writes the columns t0 - 63 .. t0 - 56 of this band's last row in this block: their ring slots must be free
                const int need = (int)(pbase + min(t0 + (uint32_t)BL - 63u, ncol)) - RING;
                uint32_t spins = 0;
                while (c_seen < need) {
                    c_seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(Cn + cons, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (c_seen >= need) break;
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > spin_limit || ((spins & 1023u) == 0u && __hip_atomic_load((const PGM_GLOBAL int *)(uintptr_t)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_store((PGM_GLOBAL int *)(uintptr_t)abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        aborted = true;
                        break;
                    }
                }
                asm volatile("" ::: "memory");
            }
            // ec: the block in which the END node's predecessor cell is computed (its store is compiled into that variant only)
            auto step = [&](auto ic, auto ec) {
                constexpr int i = decltype(ic)::value;
                constexpr bool ENDBLK = decltype(ec)::value;
                const uint32_t t = t0 + (uint32_t)i;
                const int xs = (int)t - lane;
                // lane 0's inputs of this step: the block registers rotate one lane down per step within their row of 16 lanes
                // (row_ror:15), so lane 0 holds those of the current step; the other lanes receive what the lane above produced in
                // the previous step (wave_shr:1)
                auto rot = [&](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x12F, 0xf, 0xf, false)); };
                const float ccx = pgm_dpp_wave_shr1(ccx_o, cblk);
                const float W_up0 = pgm_dpp_wave_shr1(W_o, inW);
                const float Y_up0 = pgm_dpp_wave_shr1(Y_o, inY);
                float kc = 0.f, Sv[R];
                if (TAB) {
                    kc = pgm_dpp_wave_shr1(kc_o, kblk);
                    if (H == 0 || i + 1 < BL) kblk = rot(kblk);
#pragma unroll
                    for (int r = 0; r < R; ++r) Sv[r] = tab[trow[r] + __float_as_uint(kc)];
                }
                if (H == 0 || i + 1 < BL) cblk = rot(cblk);
                if (i + 1 < BL) { inW = rot(inW); inY = rot(inY); }
                const float gopen_y = (xs == 0) ? sg : gi;
                uint32_t toff = 0u;
                if (KEEP) toff = ((uint32_t)xs < ncol) ? t * (1024u * R) : 0x80000000u;
                float W_up = W_up0, Y_up = Y_up0, W_dg = W_diag0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float Mv = __fsub_rn(__fsub_rn(__fadd_rn(W_dg, TAB ? Sv[r] : Sc[r][H * BL + i]), ccy[r]), ccx);
                    const float ax = __fadd_rn(X_left[r], ge), ay = __fadd_rn(Y_up, ge);
                    const float Xv = __fsub_rn(fmaxf(ax, __fadd_rn(W_left[r], gox[r])), ccx);
                    const float Yv = __fsub_rn(fmaxf(ay, __fadd_rn(W_up, gopen_y)), ccy[r]);
                    float Wv = fmaxf(Mv, fmaxf(Xv, Yv));
                    // the walk's decisions at this cell (see above): each comparison shifts one bit into the block's word
                    uint32_t w = (i == 0) ? 0u : cw[r];
                    w = w + w + ((Wv == Mv) ? 1u : 0u);
                    w = w + w + ((Wv == Yv) ? 1u : 0u);
                    w = w + w + ((__fsub_rn(ax, ccx) != Xv) ? 1u : 0u);
                    w = w + w + ((__fsub_rn(ay, ccy[r]) != Yv) ? 1u : 0u);
                    cw[r] = w;
                    if (i == 0 && r == 0) { if (origin && t == 0u) Wv = s_init; }   // (xs == 0 in lane 0 only at the band's very first step)
                    if (KEEP) {
                        pgm_v4u cv;
                        cv.x = __float_as_uint(Mv); cv.y = __float_as_uint(Xv); cv.z = __float_as_uint(Wv); cv.w = __float_as_uint(Yv);
                        __builtin_amdgcn_raw_buffer_store_b128(cv, rsrc, soff[r] + toff, 0, 16);
                    }
                    if (ENDBLK) {
                        if (t == te && (uint32_t)r == re && (uint32_t)lane == le) *(PGM_GLOBAL pgm_v4f *)(uintptr_t)J.endcell = pgm_v4f{Mv, Xv, Wv, Yv};
                    }
                    W_dg = W_left[r];
                    W_left[r] = Wv; X_left[r] = Xv;
                    W_up = Wv; Y_up = Yv;
                    if (i == BL - 1) codes_band[(size_t)((t0 / BL) * R + r) * 64u] = w;
                }
                W_diag0 = W_up0;
                W_o = W_up; Y_o = Y_up; ccx_o = ccx; kc_o = kc;
                if (has_next) {   // lane 63 completes column t - 63 of the band's last row: slot and validity are wave-uniform
                    const uint32_t xc = t - 63u;
                    if (xc < ncol) {
                        const uint32_t slot = (pbase + xc) & (uint32_t)(RING - 1);
                        if (lane == 63) ring_out[slot] = pgm_v2f{W_up, Y_up};
                    }
                }
            };
            static_assert(BL == 8, "eight steps per block");
            if (b == be && (te & ~(uint32_t)(BL - 1)) == t0) {
                constexpr std::true_type E{};
                step(std::integral_constant<int, 0>{}, E); step(std::integral_constant<int, 1>{}, E); step(std::integral_constant<int, 2>{}, E); step(std::integral_constant<int, 3>{}, E);
                step(std::integral_constant<int, 4>{}, E); step(std::integral_constant<int, 5>{}, E); step(std::integral_constant<int, 6>{}, E); step(std::integral_constant<int, 7>{}, E);
            } else {
                constexpr std::false_type E{};
                step(std::integral_constant<int, 0>{}, E); step(std::integral_constant<int, 1>{}, E); step(std::integral_constant<int, 2>{}, E); step(std::integral_constant<int, 3>{}, E);
                step(std::integral_constant<int, 4>{}, E); step(std::integral_constant<int, 5>{}, E); step(std::integral_constant<int, 6>{}, E); step(std::integral_constant<int, 7>{}, E);
            }
            if (has_next) {
                asm volatile("" ::: "memory");
                const int done = (int)t0 + BL - 63;   // columns of the last row complete after this block
                if (done > 0) __hip_atomic_store(P + wave, (int)(pbase + min((uint32_t)done, ncol)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        for (uint32_t T0 = 0; T0 < tsteps && !aborted; T0 += 2 * BL) {
            block(T0, std::integral_constant<int, 0>{});
            if (T0 + (uint32_t)BL < tsteps && !aborted) block(T0 + (uint32_t)BL, std::integral_constant<int, 1>{});
            // next super-block's operands (loaded a super-block ago), then the loads of the one after it
            const uint32_t T1 = T0 + 2 * BL;
            if (T1 >= tsteps) break;
            take_super();
            cblk = pfc; kblk = pfk;
            load_s_super(T1 / BL + 2u);
            load_c_super(T1 + 2 * BL);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // codes (and cells) of this wavefront's bands are on their way before the worker's traceback starts
}

// Traceback of a lean job (GraphAlign.h:264-521 for two chains): wavefront 0 of the worker walks the decision bits the sweep
// left (see pgm_sweep_chain): state M goes to (y-1, x-1), a gap state to (y-1, x) / (y, x-1) and stays in the gap state
// unless the cell's opening bit is set; the state after a W source is read from the destination's bits 3-2.
// An alignment path is mostly diagonal runs, and a run needs no walker: lane k looks at the cell k + 1 steps up the diagonal,
// one ballot finds the first cell the walk does not enter in state M, the lanes before it write their mapping entries at once.
// Only the cells between runs (gaps) are walked one at a time.  The codes around the walker — the 128 steps of its band that
// end at its position: 16 words per lane and row — are staged in LDS by the walking wavefront itself; the 128 steps before
// them are requested at the same time and arrive while it walks (only a change of band is not foreseen).  Mapping entries
// are collected in LDS and go out in blocks.
template <int R>
__device__ static void pgm_traceback_chain(const PgmJob &J, uint8_t *pool, uint32_t *len_lds, const int tid, const int nthreads, unsigned long long *stat) {
    constexpr int WB = 16, NW = WB * R;          // window: WB blocks of 8 steps x R rows x 64 lanes, one word each
    constexpr uint32_t MB = 2048u;
    const uint32_t n1 = J.n1, n2 = J.n2, nblk = J.nblk;
    if (tid < 64) {
        const int lane = tid;
        typedef __attribute__((address_space(3))) uint32_t pgm_lds_u32;
        pgm_lds_u32 *win = (pgm_lds_u32 *)pool;                       // [2][NW][64]
        pgm_lds_u32 *mb1 = win + 2 * NW * 64, *mb2 = mb1 + MB;       // mapping entries not yet written out
        const PGM_GLOBAL uint32_t *codes = (const PGM_GLOBAL uint32_t *)(uintptr_t)J.codes;
        const pgm_scores s = J.sc;
        int status = PGM_OK;
        enum { State_m = 0, State_x = 1, State_y = 2 };
        int state = State_m;
        uint32_t y = n1 - 1, x = n2 - 1, len = 0, flushed = 0;
        auto flush = [&](uint32_t keep_room) {   // (LDS operations of a wavefront execute in order)
            if (len - flushed + keep_room <= MB) return;
            for (uint32_t i = (uint32_t)lane; i < len - flushed; i += 64u)
                if (flushed + i < n1 + n2) { J.map1[flushed + i] = mb1[i]; J.map2[flushed + i] = mb2[i]; }
            flushed = len;
        };
        auto push = [&](uint32_t a, uint32_t c) {
            flush(1u);
            if (lane == 0) { mb1[len - flushed] = a; mb2[len - flushed] = c; }
            ++len;
        };
        // ---- END node (GraphAlign.h:264-280 and the first step of the walk, :300-350): one predecessor pair ----
        const float yv = J.tb1[n1 - 1].v[0], xv = J.tb2[n2 - 1].v[0];
        const uint32_t yp = n1 - 2, xp = n2 - 2;
        float Wend = PGM_NEG_INF;
        push(n1 - 1, n2 - 1);
        {
            const float4 c = (yp | xp) != 0u ? pgm_gload4(J.endcell) : make_float4(PGM_NEG_INF, PGM_NEG_INF, s.start_init, PGM_NEG_INF);   // {M, X, W, Y}
            if ((yp | xp) == 0u) {
                Wend = fmaxf(__fsub_rn(__fsub_rn(s.end_skip, yv), xv), Wend);
            } else {
                Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.y, s.end_gap), yv), xv), Wend);
                Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.w, s.end_gap), yv), xv), Wend);
                Wend = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(c.x, s.end_match), yv), xv), Wend);
            }
            float best = INFINITY;
            bool found = false;
            float d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.x, s.end_match), yv), xv)));
            if (best > d) { best = d; state = State_m; found = true; }
            d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.w, s.end_gap), yv), xv)));
            if (best > d) { best = d; state = State_y; found = true; }
            d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(__fadd_rn(c.y, s.end_gap), yv), xv)));
            if (best > d) { best = d; state = State_x; found = true; }
            d = fabsf(__fsub_rn(Wend, __fsub_rn(__fsub_rn(s.end_skip, yv), xv)));
            if ((yp | xp) == 0u && best > d) { best = d; found = true; }
            if (!found) status = PGM_ERR_BACKTRACK;
            // (the same in every lane; said explicitly so that the walk below is scalar code)
            state = __builtin_amdgcn_readfirstlane(state);
            status = __builtin_amdgcn_readfirstlane(status);
            y = yp; x = xp;
            if ((x | y) != 0u) push(state == State_x ? 0xFFFFFFFFu : y, state == State_y ? 0xFFFFFFFFu : x);
        }
        // ---- code windows: blocks wq0 - WB + 1 .. wq0 of band wb ----
        int wb = -1, wq0 = -1, cur = 0, pf_b = -1, pf_q0 = -1;
        uint32_t pf[NW];
        auto issue = [&](int b, int q0) {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int qq = max(q0 - (WB - 1) + k / R, 0);
                pf[k] = codes[((size_t)((uint32_t)b * nblk + (uint32_t)qq) * R + (uint32_t)(k % R)) * 64u + (uint32_t)lane];
            }
            pf_b = b; pf_q0 = q0;
        };
        auto land = [&]() {                 // the requested window becomes the current one
            cur ^= 1;
#pragma unroll
            for (int k = 0; k < NW; ++k) win[(cur * NW + k) * 64 + lane] = pf[k];
            wb = pf_b; wq0 = pf_q0;
        };
        unsigned long long st_win = 0, st_nwin = 0, st_miss = 0, st_iter = 0;   // timeline build only
        auto need_window = [&](uint32_t yy, uint32_t xx) {   // (wave-uniform arguments) the window that holds cell (yy, xx)
            const int b = (int)(yy / (64u * R)), qw = (int)((xx + (yy % (64u * R)) / R) >> 3);
            if (b != wb || qw > wq0 || qw < wq0 - (WB - 1)) {
                const unsigned long long r0 = stat ? __builtin_amdgcn_s_memrealtime() : 0ull;
                if (pf_b != b || qw > pf_q0 || qw < pf_q0 - (WB - 1)) { issue(b, qw); ++st_miss; }
                land();
                if (wq0 >= WB) issue(wb, wq0 - WB);   // what the walk needs next unless it changes band
                if (stat) { st_win += __builtin_amdgcn_s_memrealtime() - r0; ++st_nwin; }
            }
        };
        auto code_of = [&](uint32_t yy, uint32_t xx, bool &inwin) -> uint32_t {   // per-lane cell; inwin: its word is in the current window
            const uint32_t b = yy / (64u * R), l = (yy % (64u * R)) / R, r = yy % R, t = xx + l;
            const int qw = (int)(t >> 3);
            inwin = (int)b == wb && qw <= wq0 && qw >= wq0 - (WB - 1);
            const uint32_t w = win[(cur * NW + (inwin ? (qw - (wq0 - (WB - 1))) * R + (int)r : 0)) * 64 + (int)l];
            return (w >> (28u - 4u * (t & 7u))) & 15u;
        };
        auto state_of = [](uint32_t code) { return (code & 8u) ? 0 : ((code & 4u) ? 2 : 1); };   // M, else Y, else X
        uint32_t code_cur = 0u;
        if ((x | y) != 0u && status == PGM_OK) {
            bool iw;
            need_window(y, x);
            code_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)code_of(y, x, iw));
        }
        uint32_t guard = 0;
        while ((x | y) != 0u && status == PGM_OK) {
            if (++guard > n1 + n2 + 4) { status = PGM_ERR_BACKTRACK; break; }
            ++st_iter;
            if (state == State_m) {
                // ---- a diagonal run: lane k looks at cell (y - 1 - k, x - 1 - k) ----
                if (y == 0u || x == 0u) { status = PGM_ERR_BACKTRACK; break; }
                need_window(y - 1u, x - 1u);
                const uint32_t kmax = min(min(y, x), 64u);            // cells that exist on this diagonal (lanes >= kmax: none)
                const bool exists = (uint32_t)lane < kmax;
                const uint32_t yy = exists ? y - 1u - (uint32_t)lane : 0u, xx = exists ? x - 1u - (uint32_t)lane : 0u;
                bool iw;
                const uint32_t c = code_of(yy, xx, iw);
                // the walk stops in front of: a cell outside the window, START, a cell it enters in a gap state
                const bool stop = !exists || !iw || (yy | xx) == 0u || (c & 8u) == 0u;
                const unsigned long long sm = __builtin_amdgcn_ballot_w64(stop);
                const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane(__ffsll((long long)sm) - 1);   // (lane 63 < kmax <= 64 implies some lane stops only if ...: see below)
                const uint32_t nrun = sm == 0ull ? 64u : p;           // cells entered in state M
                flush(nrun + 1u);
                if ((uint32_t)lane < nrun) { mb1[len - flushed + (uint32_t)lane] = yy; mb2[len - flushed + (uint32_t)lane] = xx; }
                len += nrun;
                if (nrun != 0u) {
                    y -= nrun; x -= nrun;
                    code_cur = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)(nrun - 1u));
                }
                if (sm == 0ull) continue;                              // 64 cells in state M: the run goes on
                // lane p's cell ends the run: outside the window (next iteration stages it), START, or entered in a gap state
                const uint32_t py = (uint32_t)__builtin_amdgcn_readlane((int)yy, (int)p), px = (uint32_t)__builtin_amdgcn_readlane((int)xx, (int)p);
                const bool p_exists = p < kmax, p_in = __builtin_amdgcn_readlane((int)iw, (int)p) != 0;
                if (!p_exists) { status = PGM_ERR_BACKTRACK; break; }  // the diagonal left the matrix without reaching START (cannot happen)
                if (!p_in) continue;
                y = py; x = px;
                if ((y | x) == 0u) break;
                code_cur = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)p);
                state = state_of(code_cur);
                push(state == State_x ? 0xFFFFFFFFu : y, state == State_y ? 0xFFFFFFFFu : x);
                continue;
            }
            // ---- a gap state: one cell at a time ----
            uint32_t ny = y, nx = x;
            bool resolve;
            if (state == State_y) {
                if (y == 0u) { status = PGM_ERR_BACKTRACK; break; }
                ny = y - 1; resolve = (code_cur & 1u) != 0u;
            } else {
                if (x == 0u) { status = PGM_ERR_BACKTRACK; break; }
                nx = x - 1; resolve = (code_cur & 2u) != 0u;
            }
            uint32_t code_n = 0u;
            if ((ny | nx) != 0u) {
                bool iw;
                need_window(ny, nx);
                code_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)code_of(ny, nx, iw));
                if (resolve) state = state_of(code_n);
            }
            y = ny; x = nx; code_cur = code_n;
            if ((x | y) != 0u) push(state == State_x ? 0xFFFFFFFFu : y, state == State_y ? 0xFFFFFFFFu : x);
        }
        push(0u, 0u);
        flush(MB + 1u);
        if (len > n1 + n2) { status = PGM_ERR_BACKTRACK; len = n1 + n2; }
        if (lane == 0) {
            J.result->score = Wend;
            J.result->n_tr_indels = 0;
            J.result->len = len;
            J.result->status = status;
            *len_lds = len;
            if (stat) {   // {ticks since the band end of wavefront 0 until the walk began << 40 | window ticks << 20 | walk ticks, windows << 32 | misses << 16 | iterations}
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                stat[0] = (((stat[1] - stat[-2]) & 0xfffffull) << 40) | ((st_win & 0xfffffull) << 20) | ((now - stat[1]) & 0xfffffull);
                stat[1] = (st_nwin << 32) | (st_miss << 16) | st_iter;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    pgm_traceback_publish(J, *len_lds, tid, nthreads);
}

// Helper wavefronts of a MODE 2 sweep (PgmJob::mode2): wavefronts 1-7 of the worker.  The sweeping wavefront keeps the terms
// it can evaluate from its own register windows (chain M, X, Y; X from columns x-2, x-3, Y from row y-2, the M pairs
// (y-1, x-2) and (y-2, x-1)).  Every other term of step t reads the history no later than step t - 3 (all but one no
// later than t - 4), so the helpers evaluate them up to a few steps ahead of the sweep (their latency is
// off its critical path as long as their throughput keeps up) and fold them per lane into ONE set of maxima res[t & 3]
// {M, X, Y} with LDS float-max atomics (exact, order free; the sweep resets the words after reading them):
//   GROUP 0  (wavefronts 1, 2) the near terms the sweep does not hold in registers, one row per lane like the sweep: Y from
//            row y-3 (part 0: the only helper term that reads step t - 3, slack 3), the six M pairs of rows y-1..y-3 x
//            columns x-1..x-3 that read step t - 4 or older (part 1, slack 4)
//   GROUP 1  (wavefronts 3, 7) the far edges of the COLUMNS, one row per lane (a column's far edges pass down the lanes one
//            step at a time): X term, M terms with the three near rows.  On-chip entries (summary slots and the overflow
//            table): every other one each; in a job with long column entries the first one takes all on-chip entries and the
//            second one (LONG) the long ones
//   GROUP 2  (wavefronts 5, 6, 4) the far edges of the ROWS, one (row, far edge) ENTRY per lane — a band has a handful of them,
//            a lane per row would loop to the largest count per row for all 64 rows: Y term, M terms with the three near
//            columns and with the far edges of the entry's current column.  Part k takes the passes k, k + 3, k + 6 of the
//            band's entry list.
// (groups 1 and 2 may run up to far_slack steps ahead of what is recorded.)  The wavefronts only meet through LDS words:
// sw[0] = last step the sweeping wavefront has recorded + 2 (1 = its prologue is done), sw[h] = number of steps whose terms
// helper wavefront h has published, sw[8] = the band's row entry list is built (by part 0 of group 2).
//
// LONG = true (PgmJob::long1 / long2): entries that are not in the LDS history — up to PGM_NLONG column entries farther than
// PGM_DCAP (slots 7, 6, 5 of the column summary) or REMOTE row entries (farther than PGM_DCAP, or above the virtual lanes).
// Their sources are read from the cell storage: they were stored at least 16 steps before the last recorded step, i.e.
// before the sweeping wavefront's last counted wait (same band), or are covered by the progress of the band above.  The
// loads of the regular pairs are issued PGM_PF steps ahead into register FIFOs, unconditionally (offset 0 when there is
// nothing to load), so that the compiler can wait for them with counted waits; only (remote row x far column) and (far
// row x long column) pairs are loaded on demand.  A long column entry travels down the lanes one step at a time like the
// column itself, so a lane only loads the cell of its OWN row; W of the three rows above comes from the lanes above, one
// step later each (the same systolic window as in the sweep; lanes 1-3 load lane 0's three rows of the band above).
// Columns with more on-chip entries than the summary holds keep the rest in an overflow table (LDS copy, any helper).
template <int GROUP, bool LONG, bool MASK = false>   // MASK: the column summaries of this job may hold long entries in their last slots
__device__ __forceinline__ void pgm_terms_helper(const PgmJob &J, const uint32_t b, uint8_t *slot, const int lane, int *sw_generic,
                                                 const int hidx, const uint32_t part, const uint32_t nparts, const bool idle = false, const bool nofold = false, unsigned long long *hst = nullptr,
                                                 const uint32_t tpar = 0u, const uint32_t tstr = 1u, const int flag_idx = 8, float *sblk_own = nullptr, const bool builder = true) {
    // (tpar, tstr: the steps this wavefront evaluates, t % tstr == tpar — pgm_crit_kernel deals the steps of a band to two wavefronts
    // per part; flag_idx: the word "row entry list built"; sblk_own: this wavefront's score block; builder: part 0 builds the list)
    constexpr int BL = PGM_BLOCK, VL = PGM_VL, HS = 64 + PGM_VL, NR = PGM_NRING, KF = PGM_KF8, RS = 5, PF = PGM_PF, NL = PGM_NLONG, KQ = 3;
    static_assert(PF == 4 && BL == 8, "FIFO slots are indexed with i & 3");
    static_assert(PGM_DCAP + 1 - (4 + PF) >= 16 + 3, "a long source must be stored before the sweep's last counted wait");
    static_assert(KQ * PGM_CPARTS >= KF, "every pass of the entry list has an owner");
    typedef __attribute__((address_space(3))) int pgm_lds_int;
    typedef __attribute__((address_space(3))) float pgm_lds_float;
    typedef uint32_t pgm_v4u __attribute__((ext_vector_type(4)));
    typedef uint32_t pgm_v2u __attribute__((ext_vector_type(2)));
    pgm_lds_int *sw = (pgm_lds_int *)sw_generic;
    const uint32_t n1 = J.n1, ncol = J.ncol, tsteps = J.tsteps, nblk = J.nblk;
    const float ge = J.sc.gap_extend, gi = J.sc.gap_init, sg = J.sc.start_gap;
    const uint32_t D = J.hD, Dm = D - 1u, DX = J.hDX, DXm = DX - 1u;
    const float *hW = (const float *)slot, *hY = hW + D * HS, *hX = hY + D * HS;
    const float4 *ring3 = (const float4 *)(hX + DX * 64u);
    uint8_t *aux = slot + J.aux_off;
    float *res = (float *)(aux + PGM_AUX_RES), *sblk = sblk_own ? sblk_own : (float *)(aux + PGM_AUX_SBLK) + part * 512u;
    uint2 *elist = (uint2 *)(aux + PGM_AUX_EL);
    int *ecnt = (int *)(aux + PGM_AUX_CNT);
    uint2 *ovtab = (uint2 *)(slot + J.ov_off);
    float *rh = (float *)(slot + J.rh_off) + part * 2048u;   // GROUP 2, LONG: W of the last 32 columns of this wavefront's remote rows [column & 31][lane]
    const bool has_ov = GROUP != 0 && J.nov2 != 0;
    const int slack = GROUP == 0 ? (part == 0u ? 3 : 4) : (int)J.far_slack;
    const uint32_t y = 64u * b + (uint32_t)lane;
    const bool rowvalid = y + 1 < n1, has_prev = b > 0;
    const uint32_t yc = rowvalid ? y : 0u;
    const float4 *niq = (const float4 *)(J.ni1 + yc);
    const float4 r0 = pgm_gload4(niq);
    const float ccy = r0.x;
    const uint32_t fy = rowvalid ? __float_as_uint(r0.w) : 0u;
    const bool geny = (fy & PGM_NF_GENERIC) != 0;
    const float c2y = !geny ? r0.y : INFINITY, c3y = !geny ? r0.z : INFINITY;
    const float gopen_x = (rowvalid && y == 0) ? sg : gi;
    const uint32_t ncol_row = rowvalid ? ncol : 0u;
    const float4 *S_band = (const float4 *)(J.S + (size_t)b * nblk * 64u * BL);
    const uint32_t lb = (uint32_t)(VL + lane);
    auto fold = [&](float *p, float v) { if (nofold) return; __builtin_amdgcn_ds_fmaxf((pgm_lds_float *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false); };
    // cell (r, c) of the job = byte offset row_off(r) + c * 1024 into the job's cell storage (32-bit: the host only
    // marks entries long / remote when the job's storage is smaller than 4 GiB)
    const __amdgpu_buffer_rsrc_t job_rsrc = pgm_band_rsrc(J.cells, LONG ? J.nb * tsteps * 1024u : 16u);
    auto row_off = [&](uint32_t r) { return (((r >> 6) * tsteps + (r & 63u)) * 64u + (r & 63u)) * 16u; };
    auto load_w = [&](uint32_t off) {   // W of the cell at byte offset off (device-coherent like every read of another wavefront's cells)
        const pgm_v2u v = __builtin_amdgcn_raw_buffer_load_b64(job_rsrc, off + 8u, 0, 16);
        return __uint_as_float(v.x);
    };
    if (has_ov) {   // (every far helper copies the table: same values to the same words, no hand-shake needed)
        for (uint32_t i = (uint32_t)lane; i < J.nov2 * (uint32_t)PGM_OV_ENT; i += 64u) ovtab[i] = make_uint2(pgm_gld(&J.ov2[i].x), pgm_gld(&J.ov2[i].y));
    }
    // GROUP 2: entry list of the band, remote entries first (order irrelevant otherwise), built by part 0.  LDS operations of
    // one wavefront execute in order.
    int ne = 0;
    // a band whose entries fit one pass (the usual case): wavefront part 0 takes the Y term and the near pairs of that pass, parts 1
    // and 2 the pairs with every other far edge of the columns; otherwise part k takes the whole passes k, k + 3, k + 6
    bool single = false;
    auto p_of = [&](int q) { return single ? (q == 0 ? 0u : 64u) : part + nparts * (uint32_t)q; };
    uint32_t e_o[KQ], e_dy[KQ];
    float e_cy[KQ];
    bool e_ok[KQ], e_rem = false;
    uint32_t e_off = 0u;
    if (GROUP == 2) {
        if (part == 0u && builder) {
            __hip_atomic_store(ecnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(ecnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int32_t f0 = (rowvalid && !geny) ? pgm_gld(J.fp1 + yc) : 0, f1 = (rowvalid && !geny) ? pgm_gld(J.fp1 + yc + 1) : 0;
            uint32_t nrem = 0, nloc = 0;
            for (int32_t e = f0; e < f1; ++e) { if (pgm_gld(&J.fe1[e].x) >> 31) ++nrem; else ++nloc; }
            uint32_t baseR = 0, baseL = 0;
            if (nrem) baseR = (uint32_t)__hip_atomic_fetch_add(ecnt, (int)nrem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nloc) baseL = (uint32_t)__hip_atomic_fetch_add(ecnt + 1, (int)nloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int totR = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ecnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            baseL += (uint32_t)totR;
            for (int32_t e = f0; e < f1; ++e) {
                const uint32_t dx = pgm_gld(&J.fe1[e].x), cb = pgm_gld(&J.fe1[e].y);
                const uint32_t pos = (dx >> 31) ? baseR++ : baseL++;
                if (pos < 512u) elist[pos] = make_uint2((uint32_t)lane | ((dx & 0x7fffffu) << 8) | (dx & 0x80000000u), cb);
            }
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + flag_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(sw + flag_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
        }
        const int totR = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ecnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        const int totL = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ecnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        ne = min(totR + totL, 512);   // (the host keeps a band within 512 entries and PGM_REMOTE_MAX remote ones)
        single = ne <= 64;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int idx = (int)p_of(q) * 64 + lane;
            e_ok[q] = idx < ne;
            const uint2 a = e_ok[q] ? elist[idx] : make_uint2((uint32_t)lane | (1u << 8), __float_as_uint(INFINITY));
            e_o[q] = a.x & 255u; e_dy[q] = (a.x >> 8) & 0x7fffffu; e_cy[q] = __uint_as_float(a.y);
            if (LONG && q == 0) {
                e_rem = (a.x >> 31) != 0;
                if (e_rem) e_off = row_off(64u * b + e_o[q] - e_dy[q]);
            }
        }
    }
    const bool do_near = !single || part == 0u, do_pairs = !single || part >= 1u;
    const uint32_t pj0 = single ? part - 1u : 0u, pjs = single ? 2u : 1u;   // this wavefront's far column entries: j = pj0, pj0 + pjs, ...
    if (GROUP == 2 && (int)p_of(0) * 64 >= ne) {   // no pass of the entry list for this wavefront: nothing to publish but "done"
        __hip_atomic_store(sw + hidx, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    // GROUP 2, LONG: {W, Y} of the source row of the remote entry of the part's first pass (passes 0 and 1 hold the remote
    // entries) at the columns x .. x + PF - 1 (FIFO) and W at x-1 .. x-3 (window)
    float2 rf[PF];
    float rw[4];
    auto issue2 = [&](int col, int slot_ix) {   // {W, Y} of the remote source at column col
        const bool ok = e_rem && (uint32_t)col < ncol;
        const pgm_v2u v = __builtin_amdgcn_raw_buffer_load_b64(job_rsrc, (ok ? e_off + (uint32_t)col * 1024u : 0u) + 8u, 0, 16);
        rf[slot_ix] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    };
    // GROUP 1, LONG: per long slot s, X and W of the lane's row at the source column of the long entry of the columns
    // x .. x + PF - 1 (FIFO), lane 0's three rows above in lanes 1-3 (FIFO), and the systolic window of the rows above
    float fX[NL][PF], fW[NL][PF], fA[NL][PF], cW0[NL], cW1[NL], cW2[NL];
    const uint32_t own_off = row_off(y);
    const bool aux_ok = has_prev && lane >= 1 && lane <= 3;
    const uint32_t aux_off = aux_ok ? row_off(64u * b - (uint32_t)lane) : 0u;
    auto issue1 = [&](int xq, int slot_ix) {           // sources of the long entries of column xq, if it has any
        const bool act = (uint32_t)xq < ncol_row;
        const uint32_t rq = (uint32_t)xq & (uint32_t)(NR - 1);
        const uint32_t nl = act ? PGM_NF_NLONG(__float_as_uint(ring3[rq].w)) : 0u;
        float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (__builtin_amdgcn_ballot_w64(nl != 0u) != 0ull) dq = ring3[rq + 2u * (uint32_t)NR];
#pragma unroll
        for (int s = 0; s < NL; ++s) {
            const bool has = nl > (uint32_t)s;
            uint32_t off = 0u, aoff = 0u;
            if (__builtin_amdgcn_ballot_w64(has) != 0ull) {
                const uint32_t col = (uint32_t)xq - __float_as_uint(s == 0 ? dq.w : (s == 1 ? dq.z : dq.y));
                off = has ? own_off + col * 1024u : 0u;
                const uint32_t col0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)col);      // lane 0's request
                const bool has0 = __builtin_amdgcn_readfirstlane((int)has) != 0;
                aoff = (has0 && aux_ok) ? aux_off + col0 * 1024u : 0u;
            }
            const pgm_v4u v = __builtin_amdgcn_raw_buffer_load_b128(job_rsrc, off, 0, 16);
            fX[s][slot_ix] = __uint_as_float(v.y); fW[s][slot_ix] = __uint_as_float(v.z);
            fA[s][slot_ix] = load_w(aoff);
        }
    };
    if (LONG && GROUP == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rw[k] = PGM_NEG_INF;
#pragma unroll
        for (int k = 0; k < 32; ++k) rh[k * 64 + lane] = PGM_NEG_INF;
    }
    if (LONG && GROUP == 1) {
#pragma unroll
        for (int s = 0; s < NL; ++s) { cW0[s] = PGM_NEG_INF; cW1[s] = PGM_NEG_INF; cW2[s] = PGM_NEG_INF; }
    }
    float4 pfs[BL / 4];
    auto load_s_block = [&](uint32_t s0) {
        const uint32_t tb = min(s0 / BL, nblk - 1u);
#pragma unroll
        for (int q = 0; q < BL / 4; ++q) pfs[q] = pgm_gload4(S_band + ((size_t)tb * 64u + (uint32_t)lane) * (BL / 4) + q);
    };
    load_s_block(0);
    int seen = 0;
    unsigned long long hwait = 0;
    const unsigned long long ht0 = hst ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (LONG && GROUP != 0) {   // the FIFOs are primed once the sweep's prologue is done (column ring staged)
        while (seen < 1) {
            seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (seen < 1) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            if (GROUP == 1) issue1(s - lane, s);
            else issue2(s - (int)e_o[0], s);
        }
    }
    for (uint32_t t0 = 0; t0 < tsteps; t0 += BL) {
        float Sc[BL];
#pragma unroll
        for (int q = 0; q < BL / 4; ++q) { Sc[4 * q] = pfs[q].x; Sc[4 * q + 1] = pfs[q].y; Sc[4 * q + 2] = pfs[q].z; Sc[4 * q + 3] = pfs[q].w; }
        if (GROUP == 2) {
#pragma unroll
            for (int i = 0; i < BL; ++i) sblk[i * 64 + lane] = Sc[i];
        }
        load_s_block(t0 + BL);
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            if (!LONG && tstr == 2u && ((uint32_t)i & 1u) != tpar) continue;   // (the other wavefront of this part takes that step)
            const uint32_t t = t0 + (uint32_t)i;
            const int need = max(1, (int)t - slack + 2);
            while (seen < need) {   // (a sleeping poll: a tight one would keep the CU's LDS pipeline and this SIMD's issue slots busy;
                                    //  the far helpers have three steps of lead, the wavefront of the step t - 3 term has none to give away)
                const unsigned long long w0 = hst ? __builtin_amdgcn_s_memrealtime() : 0ull;
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (seen < need) { if (GROUP == 0 || tstr == 2u) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4); }   // (pgm_crit_kernel's steps are short: a finer poll)
                if (hst) hwait += __builtin_amdgcn_s_memrealtime() - w0;
            }
            asm volatile("" ::: "memory");
            const int xs = (int)t - lane;
            const uint32_t rslot = ((uint32_t)xs) & (uint32_t)(NR - 1);
            const float4 cn = ring3[rslot];
            float *rs = res + (t & 3u) * 192u;
            const int s0 = i & 3, sm1 = (i + 3) & 3, sm2 = (i + 2) & 3, sm3 = (i + 1) & 3;
            if (idle) {   // (experiments: hand-shake only)
            } else if (GROUP == 0) {
                const float S = Sc[i];
                auto w_at = [&](int dy, int dx) { return hW[((t - (uint32_t)(dy + dx)) & Dm) * HS + lb - (uint32_t)dy]; };
                auto mt = [&](float w, float cy, float cx) { return __fsub_rn(__fsub_rn(__fadd_rn(w, S), cy), cx); };
                const uint32_t s3 = t - 3u;
                if (part == 0u) {   // Y from row y-3: the only helper term that reads step t - 3 (this wavefront follows the sweep most closely)
                    const float Y3 = hY[(s3 & Dm) * HS + lb - 3u], Wy3 = hW[(s3 & Dm) * HS + lb - 3u];
                    const float gopen_y = (xs == 0) ? sg : gi;
                    fold(rs + 128 + lane, __fsub_rn(fmaxf(__fadd_rn(Y3, ge), __fadd_rn(Wy3, gopen_y)), c3y));
                } else {            // the six near pairs that read step t - 4 and older: (1,3) (2,2) (3,1) (2,3) (3,2) (3,3)
                    const float w13 = w_at(1, 3), w22 = w_at(2, 2), w31 = w_at(3, 1), w23 = w_at(2, 3), w32 = w_at(3, 2), w33 = w_at(3, 3);
                    fold(rs + lane, fmaxf(fmaxf(fmaxf(mt(w13, ccy, cn.z), mt(w22, c2y, cn.y)), fmaxf(mt(w31, c3y, cn.x), mt(w23, c2y, cn.z))), fmaxf(mt(w32, c3y, cn.y), mt(w33, c3y, cn.z))));
                }
            } else if (GROUP == 1) {
                const float S = Sc[i];
                auto xterm = [&](float xp, float wp, float cx) { return __fsub_rn(fmaxf(__fadd_rn(xp, ge), __fadd_rn(wp, gopen_x)), cx); };
                auto mterm3 = [&](float W1, float W2, float W3, float cj) {
                    return fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W1, S), ccy), cj), fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W2, S), c2y), cj), __fsub_rn(__fsub_rn(__fadd_rn(W3, S), c3y), cj)));
                };
                auto from_history = [&](uint32_t d, float cj, float &Mf, float &Xf) {   // entry at distance d (in the LDS history), cost cj
                    const uint32_t s1 = t - d;
                    const float Xh = hX[(s1 & DXm) * 64u + (uint32_t)lane], Wh = hW[(s1 & Dm) * HS + lb];
                    const float W1 = hW[((s1 - 1u) & Dm) * HS + lb - 1u], W2 = hW[((s1 - 2u) & Dm) * HS + lb - 2u], W3 = hW[((s1 - 3u) & Dm) * HS + lb - 3u];
                    Xf = fmaxf(Xf, xterm(Xh, Wh, cj));
                    Mf = fmaxf(Mf, mterm3(W1, W2, W3, cj));
                };
                const uint32_t fxw = (uint32_t)xs < ncol_row ? __float_as_uint(cn.w) : 0u;
                float Mf = PGM_NEG_INF, Xf = PGM_NEG_INF;
                if (nparts != 0u) {
                    // ---- on-chip column entries, one row per lane; this wavefront's share: entries j = part, part + nparts, ... ----
                    const uint32_t nfx = fxw & PGM_NF_COUNT;
                    if (__builtin_amdgcn_ballot_w64(nfx > part) != 0ull) {
                        const int nfxw = pgm_wave_max8(nfx);
                        const float4 f1 = ring3[rslot + (uint32_t)NR], f2 = ring3[rslot + 3u * (uint32_t)NR];
                        float4 f1b = make_float4(0.f, 0.f, 0.f, 0.f), f2b = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
                        if (nfxw > 4) { f1b = ring3[rslot + 2u * (uint32_t)NR]; f2b = ring3[rslot + 4u * (uint32_t)NR]; }
                        const uint32_t fdx[KF] = {__float_as_uint(f1.x), __float_as_uint(f1.y), __float_as_uint(f1.z), __float_as_uint(f1.w),
                                                  __float_as_uint(f1b.x), __float_as_uint(f1b.y), __float_as_uint(f1b.z), __float_as_uint(f1b.w)};
                        const float fcx[KF] = {f2.x, f2.y, f2.z, f2.w, f2b.x, f2b.y, f2b.z, f2b.w};
#pragma unroll
                        for (int j = 0; j < KF; ++j) {
                            if (j < nfxw && ((uint32_t)j % nparts) == part) {
                                const bool on = !(MASK || LONG) || (uint32_t)j < nfx;   // (MASK: the last slots may hold long entries, not in the history; else absent slots have cost +inf)
                                from_history(on ? fdx[j] : 1u, on ? fcx[j] : INFINITY, Mf, Xf);
                            }
                        }
                    }
                    if (has_ov) {
                        const uint32_t nov = PGM_NF_NOV(fxw), ovi = PGM_NF_OVI(fxw) * (uint32_t)PGM_OV_ENT;
                        for (uint32_t j = part; __builtin_amdgcn_ballot_w64(j < nov) != 0ull; j += nparts) {
                            const bool on = j < nov;
                            const uint2 e = ovtab[on ? ovi + j : 0u];
                            from_history(on ? e.x : 1u, on ? __uint_as_float(e.y) : INFINITY, Mf, Xf);
                        }
                    }
                }
                if (LONG) {
                    // ---- long column entries ----
                    const uint32_t nlx = PGM_NF_NLONG(fxw);
                    float4 cq = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
                    if (__builtin_amdgcn_ballot_w64(nlx != 0u) != 0ull) cq = ring3[rslot + 4u * (uint32_t)NR];
#pragma unroll
                    for (int s = 0; s < NL; ++s) {
                        const bool has = nlx > (uint32_t)s;
                        if (__builtin_amdgcn_ballot_w64(has) != 0ull) {
                            const float c7 = s == 0 ? cq.w : (s == 1 ? cq.z : cq.y);
                            const float Xo = fX[s][s0], Wo = fW[s][s0], av = fA[s][s0];
                            const float i1 = has_prev ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av), 1)) : PGM_NEG_INF;
                            const float i2 = has_prev ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av), 2)) : PGM_NEG_INF;
                            const float i3 = has_prev ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av), 3)) : PGM_NEG_INF;
                            const float W1 = pgm_dpp_wave_shr1(cW0[s], i1), W2 = pgm_dpp_wave_shr1(cW1[s], i2), W3 = pgm_dpp_wave_shr1(cW2[s], i3);
                            if (has) {
                                Xf = fmaxf(Xf, xterm(Xo, Wo, c7));
                                Mf = fmaxf(Mf, mterm3(W1, W2, W3, c7));
                            }
                            cW0[s] = Wo; cW1[s] = W1; cW2[s] = W2;
                        }
                    }
                    issue1(xs + PF, s0);
                }
                if (__builtin_amdgcn_ballot_w64(Mf > PGM_NEG_INF || Xf > PGM_NEG_INF) != 0ull) { fold(rs + lane, Mf); fold(rs + 64 + lane, Xf); }
            } else {
                // ---- row entries, one entry per lane; this wavefront's passes: p = part + nparts q ----
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    const bool fifo = LONG && q == 0;   // (the first pass keeps its FIFO moving even without remote entries or near part: counted waits)
                    if ((int)p_of(q) * 64 < ne || fifo) {
                        const uint32_t o = e_o[q], dy = e_dy[q];
                        const float cy = e_cy[q];
                        const int xo = (int)t - (int)o;
                        const uint32_t rso = (uint32_t)xo & (uint32_t)(NR - 1);
                        const float4 cno = ring3[rso];
                        const float So = sblk[i * 64 + (int)o];
                        const float gopen_y = (xo == 0) ? sg : gi;
                        const bool rem = LONG && q == 0 && e_rem;
                        const uint32_t s1 = t - dy, lp = rem ? 0u : (uint32_t)VL + o - dy;
                        float Yh = PGM_NEG_INF, Wh = PGM_NEG_INF, W1 = PGM_NEG_INF, W2 = PGM_NEG_INF, W3 = PGM_NEG_INF;
                        if (do_near) {
                            Yh = hY[(s1 & Dm) * HS + lp]; Wh = hW[(s1 & Dm) * HS + lp];
                            W1 = hW[((s1 - 1u) & Dm) * HS + lp]; W2 = hW[((s1 - 2u) & Dm) * HS + lp]; W3 = hW[((s1 - 3u) & Dm) * HS + lp];
                        }
                        if (fifo) {
                            const float2 cur = rf[s0];
                            const bool cv = rem && (uint32_t)xo < ncol;
                            const float Wr = cv ? cur.x : PGM_NEG_INF;
                            rw[s0] = Wr;
                            if (rem) rh[((uint32_t)xo & 31u) * 64u + (uint32_t)lane] = Wr;   // (pairs with the far edges of later columns read it back)
                            if (rem && do_near) { Yh = cv ? cur.y : PGM_NEG_INF; Wh = Wr; W1 = rw[sm1]; W2 = rw[sm2]; W3 = rw[sm3]; }
                            issue2(xo + PF, s0);
                        }
                        const float Yt = __fsub_rn(fmaxf(__fadd_rn(Yh, ge), __fadd_rn(Wh, gopen_y)), cy);   // (-inf without the near part)
                        float Mt = fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W1, So), cy), cno.x),
                                         fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(W2, So), cy), cno.y), __fsub_rn(__fsub_rn(__fadd_rn(W3, So), cy), cno.z)));
                        // ---- pairs with the far entries of the entry's current column ----
                        const uint32_t fow = (e_ok[q] && (uint32_t)xo < ncol) ? __float_as_uint(cno.w) : 0u;
                        const uint32_t nfo = fow & PGM_NF_COUNT, nlo = LONG ? PGM_NF_NLONG(fow) : 0u, novo = has_ov ? PGM_NF_NOV(fow) : 0u;
                        if (do_pairs && __builtin_amdgcn_ballot_w64(nfo > pj0 || novo > pj0 || nlo != 0u) != 0ull) {
                            const int nw = pgm_wave_max8(nfo);
                            const float4 g1 = ring3[rso + (uint32_t)NR], g2 = ring3[rso + 3u * (uint32_t)NR];
                            float4 g1b = make_float4(0.f, 0.f, 0.f, 0.f), g2b = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
                            if (nw > 4 || (LONG && __builtin_amdgcn_ballot_w64(nlo != 0u) != 0ull)) { g1b = ring3[rso + 2u * (uint32_t)NR]; g2b = ring3[rso + 4u * (uint32_t)NR]; }
                            const uint32_t gdx[KF] = {__float_as_uint(g1.x), __float_as_uint(g1.y), __float_as_uint(g1.z), __float_as_uint(g1.w),
                                                      __float_as_uint(g1b.x), __float_as_uint(g1b.y), __float_as_uint(g1b.z), __float_as_uint(g1b.w)};
                            const float gcx[KF] = {g2.x, g2.y, g2.z, g2.w, g2b.x, g2b.y, g2b.z, g2b.w};
#pragma unroll
                            for (int j = 0; j < KF; ++j) {
                                if (j < nw && (pjs == 1u || ((uint32_t)j & 1u) == pj0)) {
                                    const bool on = !LONG || (uint32_t)j < nfo;
                                    float Wp = hW[((s1 - (on ? gdx[j] : 1u)) & Dm) * HS + lp];
                                    if (LONG) {   // a remote row: W of its source row at that column comes from the ring of its own walk
                                        const int colp = xo - (int)(on ? gdx[j] : 1u);
                                        const float Wr2 = rh[((uint32_t)colp & 31u) * 64u + (uint32_t)lane];
                                        if (rem) Wp = colp >= 0 ? Wr2 : PGM_NEG_INF;
                                    }
                                    Mt = fmaxf(Mt, __fsub_rn(__fsub_rn(__fadd_rn(Wp, So), cy), on ? gcx[j] : INFINITY));
                                }
                            }
                            const uint32_t ovi = PGM_NF_OVI(fow) * (uint32_t)PGM_OV_ENT;
                            if (has_ov) {
                                for (uint32_t j = pj0; __builtin_amdgcn_ballot_w64(j < novo) != 0ull; j += pjs) {
                                    const bool on = j < novo;
                                    const uint2 e = ovtab[on ? ovi + j : 0u];
                                    float Wp = hW[((s1 - (on ? e.x : 1u)) & Dm) * HS + lp];
                                    if (LONG) {
                                        const int colp = xo - (int)(on ? e.x : 1u);
                                        const float Wr2 = rh[((uint32_t)colp & 31u) * 64u + (uint32_t)lane];
                                        if (rem) Wp = colp >= 0 ? Wr2 : PGM_NEG_INF;
                                    }
                                    Mt = fmaxf(Mt, __fsub_rn(__fsub_rn(__fadd_rn(Wp, So), cy), on ? __uint_as_float(e.y) : INFINITY));
                                }
                            }
                            if (LONG) {
                                // pairs neither the history nor the remote rows' rings hold — (any far row, long column) — on demand: the
                                // loads of the slots together
                                const bool dem = nlo != 0u && pj0 == 0u;   // (one of the wavefronts that share a pass)
                                if (__builtin_amdgcn_ballot_w64(dem) != 0ull) {
                                    const uint32_t sb = rem ? e_off : row_off(64u * b + o - dy);
                                    float Wq[NL];
                                    bool onq[NL];
#pragma unroll
                                    for (int u = 0; u < NL; ++u) {
                                        const int j = KF - 1 - u, col = xo - (int)gdx[j];
                                        onq[u] = dem && (uint32_t)u < nlo && col >= 0;
                                        Wq[u] = load_w(onq[u] ? sb + (uint32_t)col * 1024u : 0u);
                                    }
#pragma unroll
                                    for (int u = 0; u < NL; ++u) Mt = fmaxf(Mt, __fsub_rn(__fsub_rn(__fadd_rn(onq[u] ? Wq[u] : PGM_NEG_INF, So), cy), gcx[KF - 1 - u]));
                                }
                            }
                        }
                        if (e_ok[q]) { fold(rs + o, Mt); fold(rs + 128 + o, Yt); }
                    }
                }
            }
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + hidx, (int)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (hst && lane == 0) { hst[hidx] = hwait; hst[8 + hidx] = __builtin_amdgcn_s_memrealtime() - ht0; }   // timeline: ticks in the poll loop, ticks in all
}

// NOTRACEBACK = true: timing build for tools (the fill alone, no traceback); DBG = true: timeline (PGM_FILL_TRACE) and the
// experiment switches of PGM_TEST_NOSTORE
template <bool NOTRACEBACK, bool DBG>
__global__ void __launch_bounds__(64 * PGM_WAVES, 1) pgm_fill_kernel(const PgmJob *__restrict__ jobs, const PgmItem *__restrict__ items, uint32_t nitems,
                                                      int *__restrict__ sync, unsigned long long *__restrict__ trace,
                                                      uint32_t spin_limit, uint32_t stall_job, uint32_t stall_band, uint32_t dbg_flags_, uint32_t ticket_off, uint32_t tbq_off) {
    const uint32_t dbg_flags = DBG ? dbg_flags_ : 0u;
    if (!DBG) trace = nullptr;
    int *abort_flag = sync;        // [0] abort flag, [1] ticket counter of the band list, [2] lean list; the traceback kernel's words from [32] on (PGM_SY_*)
    // LDS of the band sweeps (one slot per sweeping wavefront)
    __shared__ __attribute__((aligned(16))) struct { uint8_t pool[PGM_POOL]; } L;
    __shared__ int item_lds;
    __shared__ __attribute__((aligned(16))) int fsync[12];   // MODE 2 item: [0] last recorded step + 2, [1..7] steps published by helper wavefront h, [8] row entry list built
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // role is wave-uniform: keep its branches scalar
    bool aborted = false;
    for (;;) {
        // next item of the list (all wavefronts of the worker take the same one)
        __syncthreads();
        if (threadIdx.x == 0) {
            int it = -1;
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                it = __hip_atomic_fetch_add(sync + ticket_off, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ([1], or [PGM_SY_CRIT_TICKET]: the launch of the batch's longest chains)
            item_lds = (it >= 0 && (uint32_t)it < nitems) ? it : -1;
            for (int k = 0; k < 12; ++k) fsync[k] = 0;
        }
        __syncthreads();
        const int it = item_lds;
        if (it < 0) break;
        const PgmItem item = items[it];
        {   // s_setprio takes an immediate
            const uint32_t pr = __builtin_amdgcn_readfirstlane(item.prio);
            if (pr >= 3u) __builtin_amdgcn_s_setprio(3); else if (pr == 2u) __builtin_amdgcn_s_setprio(2); else if (pr == 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        // optional timeline (tools/probe_trace.py): per item {worker, start, end of band, end of traceback} in 100 MHz ticks
        if (trace && threadIdx.x == 0) { trace[6 * it] = blockIdx.x; trace[6 * it + 1] = __builtin_amdgcn_s_memrealtime(); trace[6 * it + 2] = 0; trace[6 * it + 3] = 0; trace[6 * it + 4] = 0; trace[6 * it + 5] = 0; }
        const PgmJob &J = jobs[item.job];
        const uint32_t b = item.band + (uint32_t)role;
        const bool last_band = (item.band + item.count == J.nb);
        const unsigned long long clk0 = (DBG && trace) ? __builtin_readcyclecounter() : 0ull;
        if ((uint32_t)role < item.count) {
            unsigned long long wait_ticks[2] = {0, 0};   // timeline only: waiting for band b-1, waiting for the helpers
            const bool stall = item.job == stall_job && b == stall_band;
            uint8_t *slot = L.pool + (size_t)role * J.slot_bytes;
            if (J.mode2) pgm_sweep_band<2, DBG>(J, b, slot, lane, abort_flag, aborted, spin_limit, stall, trace ? wait_ticks : nullptr, fsync, dbg_flags);
            else if (J.has_extras) pgm_sweep_band<1, DBG>(J, b, slot, lane, abort_flag, aborted, spin_limit, stall, trace ? wait_ticks : nullptr, nullptr, dbg_flags);
            else pgm_sweep_band<0, DBG>(J, b, slot, lane, abort_flag, aborted, spin_limit, stall, trace ? wait_ticks : nullptr, nullptr, dbg_flags);
            if (trace && threadIdx.x == 0) { trace[6 * it] |= wait_ticks[0] << 16; if (!last_band) { trace[6 * it + 4] = wait_ticks[1]; trace[6 * it + 5] = __builtin_readcyclecounter() - clk0; } }   // (wavefront 0 of the worker; worker id in the low 16 bits)
        } else if (J.mode2 && !(dbg_flags & 4u)) {
            // helpers of the sweeping wavefront 0 (a MODE 2 item is one band); they yield issue slots to sweeping wavefronts
            __builtin_amdgcn_s_setprio(0);
            unsigned long long *hst = trace ? trace + 6 * (size_t)nitems + 16 * (size_t)it : nullptr;
            if (role <= 2) { if (!(dbg_flags & 32u)) pgm_terms_helper<0, false>(J, item.band, L.pool, lane, fsync, role, (uint32_t)(role - 1), 2u, false, false, hst); }
            else if (J.has_far && !(dbg_flags & 16u)) {
                // wavefront 4 shares its SIMD with the sweeping wavefront 0: it gets the part that usually has the least to do
                // (the last third of the row entry passes); column helpers: wavefronts 3 and 7, row helpers: 5, 6, 4
                if (role == 3 || role == 7) {
                    const uint32_t part = role == 3 ? 0u : 1u;
                    if (!J.long2) pgm_terms_helper<1, false>(J, item.band, L.pool, lane, fsync, role, part, 2u, (dbg_flags & 256u) != 0, (dbg_flags & 2048u) != 0, hst);
                    else if (part == 0u) pgm_terms_helper<1, false, true>(J, item.band, L.pool, lane, fsync, role, 0u, 1u, (dbg_flags & 256u) != 0, false, hst);
                    else pgm_terms_helper<1, true>(J, item.band, L.pool, lane, fsync, role, 0u, 0u, (dbg_flags & 256u) != 0, false, hst);   // (nparts = 0: the long entries only)
                } else {
                    const uint32_t part = role == 4 ? 2u : (uint32_t)(role - 5);
                    if (J.long1 | J.long2) pgm_terms_helper<2, true>(J, item.band, L.pool, lane, fsync, role, part, (uint32_t)PGM_CPARTS, (dbg_flags & 512u) != 0, false, hst);
                    else pgm_terms_helper<2, false>(J, item.band, L.pool, lane, fsync, role, part, (uint32_t)PGM_CPARTS, (dbg_flags & 512u) != 0, (dbg_flags & 2048u) != 0, hst);
                }
            }
        }
        if (trace && threadIdx.x == 0) trace[6 * it + 2] = __builtin_amdgcn_s_memrealtime();
        if (last_band) {
            // The last band of a job is the last one to finish.  Its traceback is not this kernel's business (the walker's code in
            // here cost the sweeps 40 VGPRs, and a worker that walks for 0.6 ms sweeps nothing): pgm_tb_kernel follows on the stream.
            // Only an aborted batch leaves its records here.
            __syncthreads();
            if (threadIdx.x == 0) J.times[0] = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                J.result->score = 0.f; J.result->n_tr_indels = 0; J.result->len = 0; J.result->status = PGM_ERR_DEVICE; J.hresult->score = 0.f; J.hresult->n_tr_indels = 0; J.hresult->len = 0; __threadfence_system(); __hip_atomic_store(&J.hresult->status, (int32_t)PGM_ERR_DEVICE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (threadIdx.x == 0) pgm_tbq_push(sync, tbq_off, item.job);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The bands of the jobs that need no helper wavefronts (MODE 0 / 1: one wavefront sweeps one band on its own), scheduled per
// WAVEFRONT.  In pgm_fill_kernel a worker's eight wavefronts take eight consecutive bands of one job together and meet again
// when the last of them is through: band k of the item starts 78 k steps after band 0 and ends as much later, so for the
// 1000-column jobs of the lower tree levels a third of the worker's wavefront-time is ramp, and a job of 17 bands leaves a
// worker with ONE busy wavefront for its last item.  Here every wavefront of a worker takes its own next band from the list
// (ordered by the host like the items: longest remaining path first, a job's bands ascending) and sweeps it in its own
// eighth of the CU's LDS; nothing in a MODE 0 / 1 sweep involves another wavefront.  Runs beside pgm_fill_kernel (which keeps
// the MODE 2 jobs) and pgm_lean_kernel on a stream of its own, on its share of the CUs; pgm_tb_kernel follows all three.
// The list has two parts: [0, nnarrow) bands whose sweep fits an eighth of the CU's LDS — workers 0 .. nworkers_narrow - 1, eight
// sweeping wavefronts each — and [nnarrow, nbands) WIDE bands that need up to a quarter (a 32-step history: the middle levels of a
// guide tree) — the remaining workers, PGM_WIDE_WAVES sweeping wavefronts each (the others leave).
__global__ void __launch_bounds__(64 * PGM_WAVES, 1) pgm_band_kernel(const PgmJob *__restrict__ jobs, const PgmItem *__restrict__ bands, uint32_t nnarrow, uint32_t nbands,
                                                                  uint32_t nworkers_narrow, int *__restrict__ sync, uint32_t spin_limit, uint32_t stall_job, uint32_t stall_band, uint32_t tbq_off) {
    __shared__ __attribute__((aligned(16))) uint8_t pool[PGM_POOL];
    int *abort_flag = sync;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool wide = blockIdx.x >= nworkers_narrow;
    if (wide && role >= PGM_WIDE_WAVES) return;
    uint8_t *slot = pool + (size_t)role * (wide ? (PGM_POOL / PGM_WIDE_WAVES / 16 * 16) : (PGM_POOL / PGM_WAVES / 16 * 16));
    const uint32_t qbase = wide ? nnarrow : 0u, qend = wide ? nbands : nnarrow;
    int *ticket = sync + (wide ? PGM_SY_WIDE_TICKET : PGM_SY_BAND_TICKET);
    bool aborted = false;
    for (;;) {
        // every lane takes part in the dequeue (lane 0 adds 1, the others 0), see pgm_nw_kernel
        uint32_t q = (uint32_t)__hip_atomic_fetch_add(ticket, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q = qbase + (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
        if (q >= qend || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        const PgmItem item = bands[q];
        {   // s_setprio takes an immediate
            const uint32_t pr = __builtin_amdgcn_readfirstlane(item.prio);
            if (pr >= 3u) __builtin_amdgcn_s_setprio(3); else if (pr == 2u) __builtin_amdgcn_s_setprio(2); else if (pr == 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        const PgmJob &J = jobs[item.job];
        const uint32_t b = item.band;
        const bool stall = item.job == stall_job && b == stall_band;
        if (J.has_extras) pgm_sweep_band<1, false>(J, b, slot, lane, abort_flag, aborted, spin_limit, stall, nullptr, nullptr, 0u);
        else pgm_sweep_band<0, false>(J, b, slot, lane, abort_flag, aborted, spin_limit, stall, nullptr, nullptr, 0u);
        if (b + 1u == J.nb && lane == 0) J.times[0] = __builtin_amdgcn_s_memrealtime();
        if (b + 1u == J.nb && lane == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {   // an aborted batch leaves its records here
            J.result->score = 0.f; J.result->n_tr_indels = 0; J.result->len = 0; J.result->status = PGM_ERR_DEVICE; J.hresult->score = 0.f; J.hresult->n_tr_indels = 0; J.hresult->len = 0; __threadfence_system(); __hip_atomic_store(&J.hresult->status, (int32_t)PGM_ERR_DEVICE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (b + 1u == J.nb && lane == 0) pgm_tbq_push(sync, tbq_off, item.job);
    }
}

// ---------------------------------------------------------------------------------------------
// Tracebacks of the jobs the sweep kernels (pgm_fill_kernel, pgm_crit_kernel, pgm_band_kernel) have swept, and the pre-linking that
// makes the long ones short.  One worker of 512 threads per CU.  A worker takes jobs until there is none left; from then on (and
// while the job it has claimed is still being swept) it pre-links grid tiles of the corridors of the tracebacks under way
// (pgm_prelink_tile), in the order their walkers will reach them, and leaves when the last traceback has finished.
//
// Two ways of getting jobs:
//   tbq_off == 0  the kernel FOLLOWS a sweep kernel on its stream (every cell final and visible): entry k of `list`, largest job first
//   tbq_off != 0  the kernel runs BESIDE the sweep kernels from the start of the stage, on CUs of its own (round 4: with the sweeps
//                 fast, a traceback that only starts when its whole launch is through was the longest tail of the stage): the worker
//                 that completes a job's last band appends the job to the ready queue sync_[tbq_off ..] (pgm_tbq_push); a worker
//                 here claims queue position k, waits for its entry (agent-scope acquire: the job's cells were written through by
//                 wavefronts that waited for their stores before they published their progress) and walks that job.
//   walk == 0     pre-link only: instances that follow a sweep kernel on its CUs and lend them to the tracebacks still under way.
// All grids of a stage are resident together (their workers add up to the CUs of the device).  Round 3 ran this kernel beside the
// sweeps with its grid WAITING for CUs and saw the sweeps stop for a minute in one launch of a few hundred: every sweep worker was
// resident (counters), ticket 250 of 787 taken, bands with a finished predecessor not started, the traceback grid with zero CUs.
// Nothing in the kernels waits for the traceback grid, so the waves were not waiting for each other: they were not running.  With
// more runnable queues than the device can hold, the hardware scheduler time-slices the queues — it saves the resident waves of one
// (they keep their place in the list but make no progress) to let the other in, which cannot make progress either while the first
// still owns work it needs; every switch costs milliseconds, and a poll budget counted in polls, not in time, stretched the
// round trips to a minute.  Hence the rule, enforced by the host (cu_shares): no grid of a stage ever waits for a CU.
template <bool DBG>
__global__ void __launch_bounds__(64 * PGM_WAVES, 1) pgm_tb_kernel(const PgmJob *__restrict__ jobs, const int2 *__restrict__ list, uint32_t ntb, int *__restrict__ sync_,
                                                                unsigned long long *__restrict__ trace, uint32_t spin_limit, uint32_t lq_off, uint32_t sybase,
                                                                uint32_t tbq_off, uint32_t walk) {
    if (!DBG) trace = nullptr;
    __shared__ __attribute__((aligned(16))) union { PgmTbLds t; PgmLkLds g; } L;
    __shared__ int cmd_lds, arg_lds;
    // (a batch has up to two instances of this kernel — behind pgm_band_kernel and behind pgm_fill_kernel — each with its own
    // list, counters (sybase) and announcements)
    int *abort_flag = sync_, *sync = sync_ + sybase, *lq = sync + PGM_SY_LQ_N, *lq_ids = sync_ + lq_off;
    bool walking = walk != 0u;   // there may be a job left to take
    int claimed = -1;            // (thread 0, tbq_off != 0) position of the ready queue this worker has claimed and is waiting for
    uint32_t lk_backoff = 0u, lk_first = 0u;   // (thread 0: first announcement that may still have tiles)
    // the worker gives up — and says so: abort flag — after spin_limit idle polls (the hand-off test's knob) or, by default, when
    // nothing has happened for PGM_IDLE_LIMIT_TICKS of the real-time counter (not after a number of polls: a poll's length varies)
    unsigned long long idle_since = __builtin_amdgcn_s_memrealtime();
    for (uint32_t polls = 0;; ++polls) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int cmd = -2;   // -3: traceback of job arg (list entry arg when following a kernel); >= 0: pre-link tile cmd of job arg; -2: nothing right now; -1: leave
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                (uint32_t)__hip_atomic_load(sync + PGM_SY_TB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ntb) cmd = -1;
            else if ((spin_limit && polls > spin_limit) || (!spin_limit && (polls & 255u) == 255u && __builtin_amdgcn_s_memrealtime() - idle_since > PGM_IDLE_LIMIT_TICKS)) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cmd = -1;
            } else if (walking && tbq_off == 0u) {
                const uint32_t k = (uint32_t)__hip_atomic_fetch_add(sync + PGM_SY_TBQ_N, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k < ntb) { cmd = -3; arg_lds = list[k].x; } else cmd = -5;
            } else if (walking) {
                if (claimed < 0) {
                    const uint32_t k = (uint32_t)__hip_atomic_fetch_add(sync + PGM_SY_TBQ_N, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k < ntb) claimed = (int)k; else cmd = -5;
                }
                if (claimed >= 0) {
                    const int id = __hip_atomic_load(sync_ + tbq_off + claimed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (id != 0) { cmd = -3; arg_lds = id - 1; claimed = -1; }
                }
            }
            if (cmd == -2 && (polls & ((1u << lk_backoff) - 1u)) == 0u) {
                // pre-link: the first announcements first (the jobs are taken largest first, so these are the longest walks), at most
                // eight with tiles left looked at per poll; the grid rows a walker has left are skipped in one step
                const uint32_t na = (uint32_t)__hip_atomic_load(lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t looked = 0;
                for (uint32_t t = lk_first; t < na && looked < 8u && cmd == -2; ++t) {
                    const int id = __hip_atomic_load(lq_ids + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (id == 0) continue;       // announced, not written yet
                    ++looked;
                    const PgmJob &Jq = jobs[id - 1];
                    const uint32_t ntiles = Jq.lrows * PGM_LK_W;
                    int *next = Jq.lready + Jq.lrows;
                    const uint32_t cur = (uint32_t)__hip_atomic_load(next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur >= ntiles) { if (t == lk_first) ++lk_first; --looked; continue; }   // (nothing left there, for good)
                    // tiles are handed out from the END corner back (ticket k -> tile ntiles - 1 - k)
                    const uint32_t wrow = (uint32_t)__hip_atomic_load(next + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t skip_to = ntiles - min(ntiles, (wrow + 1u) * PGM_LK_W);
                    if (cur < skip_to) __hip_atomic_fetch_max(next, (int)skip_to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t k = (uint32_t)__hip_atomic_fetch_add(next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k < ntiles && k >= skip_to) { cmd = (int)(ntiles - 1u - k); arg_lds = id - 1; }
                }
            }
            cmd_lds = cmd;
        }
        __syncthreads();
        const int cmd = cmd_lds;
        if (cmd == -1) break;
        if (cmd == -5) { walking = false; continue; }
        if (cmd == -2) { lk_backoff = min(lk_backoff + 1u, 3u); if (walking) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(127); continue; }   // ~1 / ~3 us; the announcements are looked at every 2, 4, 8 polls while they have nothing
        lk_backoff = 0u;
        if (threadIdx.x == 0) idle_since = __builtin_amdgcn_s_memrealtime();
        if (cmd == -3) {
            const int jid = arg_lds;
            int it = 0;   // (the job's last item of the work list: the timeline's slot; tools build, following a kernel)
            if (trace) for (uint32_t k = 0; k < ntb; ++k) if (list[k].x == jid) it = list[k].y;
            if (tbq_off != 0u) {   // the job's cells come from other CUs of this very stage: nothing stale in this CU's cache
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            pgm_traceback_job(jobs[jid], L.t, (int)threadIdx.x, 64 * PGM_WAVES, trace ? trace + 6 * it + 4 : nullptr, lq, lq_ids, (uint32_t)jid);
            if (threadIdx.x == 0) jobs[jid].times[1] = __builtin_amdgcn_s_memrealtime();
            if (trace && threadIdx.x == 0) trace[6 * it + 3] = __builtin_amdgcn_s_memrealtime();
        } else {
            if (tbq_off != 0u) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            pgm_prelink_tile(jobs[arg_lds], L.g, (uint32_t)cmd, (int)threadIdx.x);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The lean jobs of a batch (PgmJob::lean) have a kernel of their own: they share nothing with the other jobs, and compiled
// into pgm_fill_kernel their code changed the register allocation of that kernel's hot loops (the band sweeps and the helper
// wavefronts ran 6-30 % slower).  Here it gets its own budget (one worker of eight wavefronts per CU, like the fill kernel).
// Persistent workers take whole jobs (largest first) through an atomic ticket; a job's eight wavefronts sweep its bands
// (pgm_sweep_chain), then wavefront 0 walks the decision bits (pgm_traceback_chain).  The host launches it on a second
// stream beside pgm_fill_kernel, whose grid leaves the CUs for it free.
#define PGM_LEAN_LDS (8 * PGM_LEAN_RING * 8 > 49152 ? 8 * PGM_LEAN_RING * 8 : 49152)
template <int R>
__global__ void __launch_bounds__(64 * PGM_WAVES, 2) pgm_lean_kernel(const PgmJob *__restrict__ jobs, const uint32_t *__restrict__ list, uint32_t nlist,
                                                                  int *__restrict__ sync, unsigned long long *__restrict__ trace, uint32_t spin_limit) {
    __shared__ __attribute__((aligned(16))) uint8_t pool[PGM_LEAN_LDS];   // rings of the sweep; windows and mapping block of the walk
    __shared__ __attribute__((aligned(16))) int fsync[16];                // [0..7] columns produced, [8..15] columns consumed by wavefront w
    __shared__ int item_lds, tb_go;
    __shared__ uint32_t tb_len;
    int *abort_flag = sync;        // [0] abort flag of the batch, [2] ticket counter of the lean list
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    bool aborted = false;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int it = -1;
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                it = __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item_lds = (it >= 0 && (uint32_t)it < nlist) ? it : -1;
            for (int k = 0; k < 16; ++k) fsync[k] = 0;
        }
        __syncthreads();
        const int it = item_lds;
        if (it < 0) break;
        const PgmJob &J = jobs[list[it]];
        if (trace && threadIdx.x == 0) { trace[6 * it] = blockIdx.x; trace[6 * it + 1] = __builtin_amdgcn_s_memrealtime(); trace[6 * it + 2] = 0; trace[6 * it + 3] = 0; trace[6 * it + 4] = 0; trace[6 * it + 5] = 0; }
        const bool tabmode = J.tabhdr != nullptr && pgm_gld(J.tabhdr) == 0;   // (wave-uniform: every node of both graphs has a class)
        if (tabmode) {
            // the job's score table: entry (class of the row, class of the column) from one node of either class, with the operations of
            // pgm_emission_skew_kernel (terms in ascending order, one multiply and one add each, then pgm_emission_finish)
            float *tabw = (float *)(pool + PGM_LEAN_TAB_OFF);
            const uint32_t NC = J.dim + 2u, DP = J.dp;
            const float mi = J.sc.match_init;
            for (uint32_t e = threadIdx.x; e < NC * NC; e += 64u * PGM_WAVES) {
                const int r1 = pgm_gld(J.tabhdr + 1 + e / NC), r2 = pgm_gld(J.tabhdr + 1 + NC + e % NC);
                float v = 0.f;
                if (r1 != 0 && r2 != 0) {
                    const float *g = J.g1f + (size_t)DP * (uint32_t)(r1 - 1), *tc = J.t2 + (size_t)DP * (uint32_t)(r2 - 1);
                    float acc = 0.0f;
                    for (uint32_t k = 0; k < DP; ++k) acc = __fadd_rn(acc, __fmul_rn(pgm_gld(g + k), pgm_gld(tc + k)));
                    v = pgm_emission_finish(acc, pgm_gld(J.a1 + (r1 - 1)), pgm_gld(J.b2 + (r2 - 1)), mi);
                }
                tabw[e] = v;
            }
            __syncthreads();
            pgm_sweep_chain<R, false, true>(J, role, lane, pool, fsync, abort_flag, aborted, spin_limit);
        }
        else if (J.keep_cells) pgm_sweep_chain<R, true>(J, role, lane, pool, fsync, abort_flag, aborted, spin_limit);
        else pgm_sweep_chain<R, false>(J, role, lane, pool, fsync, abort_flag, aborted, spin_limit);
        if (trace && threadIdx.x == 0) trace[6 * it + 2] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (threadIdx.x == 0) {
            J.times[0] = __builtin_amdgcn_s_memrealtime();
            const bool ok = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
            if (!ok) { J.result->score = 0.f; J.result->n_tr_indels = 0; J.result->len = 0; J.result->status = PGM_ERR_DEVICE; J.hresult->score = 0.f; J.hresult->n_tr_indels = 0; J.hresult->len = 0; __threadfence_system(); __hip_atomic_store(&J.hresult->status, (int32_t)PGM_ERR_DEVICE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
            tb_go = ok ? 1 : 0;
        }
        __syncthreads();
        if (tb_go != 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (trace && threadIdx.x == 0) trace[6 * it + 5] = __builtin_amdgcn_s_memrealtime();   // (the walker's statistics are relative to this)
            pgm_traceback_chain<R>(J, pool, &tb_len, (int)threadIdx.x, 64 * PGM_WAVES, trace ? trace + 6 * it + 4 : nullptr);
            if (threadIdx.x == 0) J.times[1] = __builtin_amdgcn_s_memrealtime();
        }
        if (trace && threadIdx.x == 0) trace[6 * it + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

#endif
