// Micro-model of a MODE 2 band sweep (one 512-thread workgroup per CU): what does a step of the critical wavefront cost when
// it only carries the chain terms (the t-1 dependencies) and everything else is evaluated by sibling wavefronts from the LDS
// history?  Timing only (synthetic costs and scores); the arithmetic shape, the LDS traffic and the hand-shakes are those of
// pgm_sweep_band<2> / pgm_terms_helper.
//
//   wavefront 0  (A)  chain: X from column x-1, Y from row y-1 (two DPP shifts), merge with the pre-folded maxima {M, X, Y} of
//                     the step, W, cell store, record {W, Y, X} in the history, counter
//   wavefronts 1, 2 (B, one for the even and one for the odd steps): every near term the chain wavefront does not hold — the
//                     terms that read step t-3 and older first, then (once step t-2 is recorded) the three that read step t-2 —
//                     folded with the far helpers' maxima and written as the step's pre-folded maxima
//   wavefronts 3..7 (F) stand-ins for the far helpers: K history reads, V vector operations and two LDS float-max atomics per step,
//                     four steps of lead
// variants (argv): see main().  Prints cycles per step of wavefront 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define HS 80
#define VL 16
#define DD 64
#define DM (DD - 1)
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) float lds_float;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct Args {
    int steps;
    int mode;      // 0: A alone (never waits, reads pre), 1: A + B, 2: A + B + F, 3: OLD layout (A carries the t-2 terms, merges 3 words + resets, B = light near helpers, F)
    int nfar;      // far stand-ins (0..5)
    int farK, farV;
    int store;     // cell stores + progress publication
    int xv;        // extra dependent vector ops in A per step (calibration)
    float *cells;  // [steps][64] float4
    int *prog;
    float *out;
    unsigned long long *clk;   // [0] cycles of A, [1] realtime ticks, [2] A's wait cycles, [3] aborted
};

__device__ __forceinline__ float dpp_shr1(float src, float lane0) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0), __float_as_int(src), 0x138, 0xf, 0xf, false));
}

template <int MODE, bool STORE>
__global__ void __launch_bounds__(512, 1) model(Args a) {
    __shared__ __attribute__((aligned(16))) float hW[DD * HS], hY[DD * HS], hX[DD * 64];
    __shared__ __attribute__((aligned(16))) v4f pre[4 * 64];
    __shared__ __attribute__((aligned(16))) float res[8 * 192];
    __shared__ __attribute__((aligned(16))) v4f colq[128];
    __shared__ __attribute__((aligned(16))) v2f colA[128];   // what the chain wavefront needs of a column: {cost of the chain edge, flags}
    __shared__ __attribute__((aligned(16))) float preM[4 * 64], preX[4 * 64], preY[4 * 64];
    __shared__ __attribute__((aligned(16))) int swg[16];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    lds_int *sw = (lds_int *)swg;
    for (int i = threadIdx.x; i < DD * HS; i += 512) { hW[i] = -1e30f; hY[i] = -1e30f; }
    for (int i = threadIdx.x; i < DD * 64; i += 512) hX[i] = -1e30f;
    for (int i = threadIdx.x; i < 4 * 64; i += 512) pre[i] = v4f{-1.f, -2.f, -3.f, 0.f};
    for (int i = threadIdx.x; i < 8 * 192; i += 512) res[i] = -INFINITY;
    for (int i = threadIdx.x; i < 128; i += 512) colq[i] = v4f{0.5f + 0.01f * i, (i % 5 == 0) ? 1.5f : INFINITY, (i % 7 == 0) ? 2.5f : INFINITY, 0.f};
    for (int i = threadIdx.x; i < 128; i += 512) colA[i] = v2f{0.5f + 0.01f * i, 0.f};
    for (int i = threadIdx.x; i < 4 * 64; i += 512) { preM[i] = -1.f; preX[i] = -2.f; preY[i] = -3.f; }
    if (threadIdx.x < 16) swg[threadIdx.x] = 0;
    __syncthreads();
    const int steps = a.steps;
    const float ge = -0.7f, gi = -3.1f;
    const float ccy = 0.3f + 0.001f * lane, c2y = (lane % 3 == 0) ? 1.1f : INFINITY, c3y = (lane % 5 == 0) ? 2.1f : INFINITY;
    const uint32_t lb = VL + lane;
    const int SPIN = 1 << 18;
    if (wave == 0) {
        // ---------------- A ----------------
        const bool old = MODE == 3;
        float W_o = 0.f, Y_o = -1.f, X_o = -2.f, ow2 = 0.f, ox2 = 0.f, ow3 = 0.f, ox3 = 0.f, u1Wp = 0.f, u1Wpp = 0.f, u2W = 0.f, u2Wp = 0.f, u1Yp = 0.f, u2Y = 0.f;
        v4f cn_n = colq[(uint32_t)(-lane) & 127u];
        if (!old) { const v2f c = colA[(uint32_t)(-lane) & 127u]; cn_n = v4f{c.x, c.y, 0.f, 0.f}; }
        float inW1 = hW[(0xFFFFFFFFu & DM) * HS + VL - 1], inY1 = hY[(0xFFFFFFFFu & DM) * HS + VL - 1];
        float inW2 = inW1, inY2 = inY1;
        int seenB0 = 0, seenB1 = 0, seenF = 0;
        v4i cnt_a = {0, 0, 0, 0}, cnt_b = {0, 0, 0, 0};
        unsigned long long waitc = 0;
        bool aborted = false;
        __builtin_amdgcn_s_setprio(3);
        __hip_atomic_store(sw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // prologue done: the history is initialised
        const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cells + (size_t)blockIdx.x * steps * 256), 0, (uint32_t)steps * 1024u, 0x00020000);
        for (int t0 = 0; t0 < steps && !aborted; t0 += 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = t0 + i;
            const int x = t - lane;
            const v4f cn = cn_n;
            const float iW1 = inW1, iY1 = inY1, iW2 = inW2, iY2 = inY2;
            if (MODE != 0) {
                const int want = t + 1;
                int m;
                if (!old) m = (i & 1) ? cnt_a.y : cnt_a.x;                                   // B of this step's parity
                else { m = min(min(cnt_a.y, cnt_a.z), min(cnt_a.w, cnt_b.x)); m = min(m, min(min(cnt_b.y, cnt_b.z), cnt_b.w)); }
                int seen = __builtin_amdgcn_readfirstlane(m);
                if (__builtin_expect(seen < want, 0)) {
                    const unsigned long long w0 = __builtin_readcyclecounter();
                    int spins = 0;
                    for (;;) {
                        asm volatile("" ::: "memory");
                        const v4i wa = *(const __attribute__((address_space(3))) v4i *)sw, wb = *(const __attribute__((address_space(3))) v4i *)(sw + 4);
                        if (!old) m = (i & 1) ? wa.z : wa.y;
                        else { m = min(min(wa.y, wa.z), min(wa.w, wb.x)); m = min(m, min(min(wb.y, wb.z), wb.w)); }
                        seen = __builtin_amdgcn_readfirstlane(m);
                        if (seen >= want) break;
                        if (++spins > SPIN) { aborted = true; break; }
                    }
                    waitc += __builtin_readcyclecounter() - w0;
                }
                asm volatile("" ::: "memory");
                if (old) { cnt_a = *(const __attribute__((address_space(3))) v4i *)sw; cnt_b = *(const __attribute__((address_space(3))) v4i *)(sw + 4); }
                else { const v2i c2 = *(const __attribute__((address_space(3))) v2i *)(sw + 1); cnt_a.x = c2.x; cnt_a.y = c2.y; }
            }
            // operands of step t + 1
            if (old) cn_n = colq[(uint32_t)(x + 1) & 127u];
            else { const v2f c = colA[(uint32_t)(x + 1) & 127u]; cn_n.x = c.x; cn_n.y = c.y; }
            inW1 = hW[((uint32_t)t & DM) * HS + VL - 1]; inY1 = hY[((uint32_t)t & DM) * HS + VL - 1];
            if (old) { inW2 = hW[((uint32_t)(t - 1) & DM) * HS + VL - 2]; inY2 = hY[((uint32_t)(t - 1) & DM) * HS + VL - 2]; }
            float rM, rX, rY;
            if (old) {
                const uint32_t ro = ((uint32_t)i & 3u) * 192u + (uint32_t)lane;
                rM = res[ro]; rX = res[ro + 64]; rY = res[ro + 128];
                res[ro] = -INFINITY; res[ro + 64] = -INFINITY; res[ro + 128] = -INFINITY;
            } else {
                const uint32_t po = ((uint32_t)i & 3u) * 64u + (uint32_t)lane;
                rM = preM[po]; rX = preX[po]; rY = preY[po];
            }
            const float S = 0.25f;
            const float gopen_y = (x == 0) ? -2.f : gi;
            float n2W = 0.f, n2Y = 0.f;
            if (old) { n2W = dpp_shr1(u1Wp, iW2); n2Y = dpp_shr1(u1Yp, iY2); }
            const float u1W = dpp_shr1(W_o, iW1), u1Y = dpp_shr1(Y_o, iY1);
            float Xv = __fsub_rn(fmaxf(__fadd_rn(X_o, ge), __fadd_rn(W_o, gi)), cn.x);
            float Yv = __fsub_rn(fmaxf(__fadd_rn(u1Y, ge), __fadd_rn(u1W, gopen_y)), ccy);
            float Mv = rM;
            if (old) {
                Mv = fmaxf(Mv, __fsub_rn(__fsub_rn(__fadd_rn(u1Wp, S), ccy), cn.x));
                Xv = fmaxf(Xv, fmaxf(__fsub_rn(fmaxf(__fadd_rn(ox2, ge), __fadd_rn(ow2, gi)), cn.y), __fsub_rn(fmaxf(__fadd_rn(ox3, ge), __fadd_rn(ow3, gi)), cn.z)));
                Yv = fmaxf(Yv, __fsub_rn(fmaxf(__fadd_rn(n2Y, ge), __fadd_rn(n2W, gopen_y)), c2y));
                Mv = fmaxf(Mv, fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(u1Wpp, S), ccy), cn.y), __fsub_rn(__fsub_rn(__fadd_rn(u2Wp, S), c2y), cn.x)));
            }
            Xv = fmaxf(Xv, rX); Yv = fmaxf(Yv, rY);
            if (a.xv) {
#pragma unroll
                for (int k = 0; k < 25; ++k) Xv = __fadd_rn(__fmul_rn(Xv, 0.99999f), 1e-6f);
            }
            float Wv = fmaxf(Mv, fmaxf(Xv, Yv));
            if (x == -0x40000000) Wv = 0.f;
            if (!old && __float_as_uint(cn.y) != 0u) Yv = -INFINITY;   // (column flags: kill)
            // keep the values bounded (timing model: the numbers mean nothing)
            Wv = fminf(Wv, 100.f);
            const bool active = (uint32_t)x < (uint32_t)steps;
            if (STORE) {
                v4u v; v.x = __float_as_uint(Mv); v.y = __float_as_uint(Xv); v.z = __float_as_uint(Wv); v.w = __float_as_uint(Yv);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (active ? (uint32_t)lane * 16u : 0x80000000u) + (uint32_t)t * 1024u, 0, 16);
            }
            const uint32_t ho = ((uint32_t)t & DM) * HS + lb;
            hW[ho] = Wv; hY[ho] = Yv; hX[((uint32_t)t & DM) * 64u + (uint32_t)lane] = Xv;
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw, t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ow3 = ow2; ox3 = ox2; ow2 = W_o; ox2 = X_o; u1Wpp = u1Wp; u1Wp = u1W; u2Wp = u2W; u2W = n2W; u1Yp = u1Y; u2Y = n2Y;
            W_o = Wv; Y_o = Yv; X_o = Xv;
            if (STORE && i == 3) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (lane == 0) __hip_atomic_store((__attribute__((address_space(1))) int *)(uintptr_t)(a.prog + 16 * blockIdx.x), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_store(sw, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a.out[lane] = W_o + u2Y + u2W;
        if (lane == 0 && blockIdx.x == 0) { a.clk[0] = c1 - c0; a.clk[1] = r1 - r0; a.clk[2] = waitc; a.clk[3] = aborted ? 1 : 0; }
    } else if (wave <= 2) {
        // ---------------- B ----------------
        if (MODE == 0) return;
        const bool old = MODE == 3;
        const int q = wave - 1;
        int seen = 0;
        float acc = 0.f;
        auto wait_rec = [&](int need) {
            int spins = 0;
            while (seen < need) {
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (seen < need && ++spins > SPIN) { seen = 0x7fffffff; break; }
            }
            asm volatile("" ::: "memory");
        };
        auto w_at = [&](int t, int dy, int dx) { return hW[((uint32_t)(t - dy - dx) & DM) * HS + lb - (uint32_t)dy]; };
        const float S = 0.25f;
        auto mt = [&](float w, float cy, float cx) { return __fsub_rn(__fsub_rn(__fadd_rn(w, S), cy), cx); };
        if (old) {
            // light near helpers as today: part 0 = Y from row y-3 (slack 3), part 1 = six M pairs (slack 4); every step
            for (int t = 0; t < steps; ++t) {
                wait_rec(max(1, t - (q == 0 ? 3 : 4) + 2));
                if (seen == 0x7fffffff) break;
                const int x = t - lane;
                const v4f cn = colq[(uint32_t)x & 127u];
                float *rs = res + ((uint32_t)t & 3u) * 192u;
                if (q == 0) {
                    const float Y3 = hY[((uint32_t)(t - 3) & DM) * HS + lb - 3u], W3 = hW[((uint32_t)(t - 3) & DM) * HS + lb - 3u];
                    __builtin_amdgcn_ds_fmaxf((lds_float *)(rs + 128 + lane), __fsub_rn(fmaxf(__fadd_rn(Y3, ge), __fadd_rn(W3, gi)), c3y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
                } else {
                    const float w13 = w_at(t, 1, 3), w22 = w_at(t, 2, 2), w31 = w_at(t, 3, 1), w23 = w_at(t, 2, 3), w32 = w_at(t, 3, 2), w33 = w_at(t, 3, 3);
                    __builtin_amdgcn_ds_fmaxf((lds_float *)(rs + lane), fmaxf(fmaxf(fmaxf(mt(w13, ccy, cn.z), mt(w22, c2y, cn.y)), fmaxf(mt(w31, c3y, cn.x), mt(w23, c2y, cn.z))), fmaxf(mt(w32, c3y, cn.y), mt(w33, c3y, cn.z))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
                }
                asm volatile("" ::: "memory");
                __hip_atomic_store(sw + wave, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __hip_atomic_store(sw + wave, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            a.out[64 * wave + lane] = acc;
            return;
        }
        int seenF = MODE == 2 && a.nfar > 0 ? 0 : 0x7fffffff;
        for (int t = q; t < steps; t += 2) {
            const int x = t - lane;
            // ---- phase 1: everything that reads step t-3 and older (needs t-3 recorded: sw[0] >= t-1)
            wait_rec(max(1, t - 1));
            if (seen == 0x7fffffff) break;
            const v4f cn = colq[(uint32_t)x & 127u];
            const float gopen_y = (x == 0) ? -2.f : gi;
            const float w12 = w_at(t, 1, 2), w21 = w_at(t, 2, 1), w13 = w_at(t, 1, 3), w22 = w_at(t, 2, 2), w31 = w_at(t, 3, 1), w23 = w_at(t, 2, 3), w32 = w_at(t, 3, 2), w33 = w_at(t, 3, 3);
            const float x3 = hX[((uint32_t)(t - 3) & DM) * 64u + (uint32_t)lane], wx3 = w_at(t, 0, 3);
            const float y3 = hY[((uint32_t)(t - 3) & DM) * HS + lb - 3u], wy3 = w_at(t, 3, 0);
            float Mv = fmaxf(fmaxf(fmaxf(mt(w13, ccy, cn.z), mt(w22, c2y, cn.y)), fmaxf(mt(w31, c3y, cn.x), mt(w23, c2y, cn.z))), fmaxf(mt(w32, c3y, cn.y), mt(w33, c3y, cn.z)));
            Mv = fmaxf(Mv, fmaxf(mt(w12, ccy, cn.y), mt(w21, c2y, cn.x)));
            float Xv = __fsub_rn(fmaxf(__fadd_rn(x3, ge), __fadd_rn(wx3, gi)), cn.z);
            float Yv = __fsub_rn(fmaxf(__fadd_rn(y3, ge), __fadd_rn(wy3, gopen_y)), c3y);
            // far helpers' maxima of this step (they are far ahead normally)
            if (seenF < t + 1) {
                int spins = 0;
                for (;;) {
                    asm volatile("" ::: "memory");
                    const v4i wa = *(const __attribute__((address_space(3))) v4i *)(sw + 4), wb = *(const __attribute__((address_space(3))) v4i *)(sw + 8);
                    int m = min(min(wa.x, wa.y), min(wa.z, wa.w)); m = min(m, wb.x);
                    seenF = __builtin_amdgcn_readfirstlane(m);
                    if (seenF >= t + 1 || ++spins > SPIN) break;
                }
            }
            {
                const uint32_t ro = ((uint32_t)t & 7u) * 192u + (uint32_t)lane;
                const float rM = res[ro], rX = res[ro + 64], rY = res[ro + 128];
                res[ro] = -INFINITY; res[ro + 64] = -INFINITY; res[ro + 128] = -INFINITY;
                Mv = fmaxf(Mv, rM); Xv = fmaxf(Xv, rX); Yv = fmaxf(Yv, rY);
            }
            // ---- phase 2: the three terms that read step t-2 (sw[0] >= t)
            wait_rec(max(1, t));
            if (seen == 0x7fffffff) break;
            const float w11 = w_at(t, 1, 1), x2 = hX[((uint32_t)(t - 2) & DM) * 64u + (uint32_t)lane], wx2 = w_at(t, 0, 2);
            const float y2 = hY[((uint32_t)(t - 2) & DM) * HS + lb - 2u], wy2 = w_at(t, 2, 0);
            Mv = fmaxf(Mv, mt(w11, ccy, cn.x));
            Xv = fmaxf(Xv, __fsub_rn(fmaxf(__fadd_rn(x2, ge), __fadd_rn(wx2, gi)), cn.y));
            Yv = fmaxf(Yv, __fsub_rn(fmaxf(__fadd_rn(y2, ge), __fadd_rn(wy2, gopen_y)), c2y));
            { const uint32_t po = ((uint32_t)t & 3u) * 64u + (uint32_t)lane; preM[po] = fminf(Mv, 50.f); preX[po] = fminf(Xv, 50.f); preY[po] = fminf(Yv, 50.f); }
            asm volatile("" ::: "memory");
            __hip_atomic_store(sw + wave, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __hip_atomic_store(sw + wave, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a.out[64 * wave + lane] = acc;
    } else {
        // ---------------- F ----------------
        const int h = wave - 3;
        const bool old = MODE == 3;
        lds_int *mine = old ? sw + wave : sw + 4 + h;
        if ((MODE != 2 && MODE != 3) || h >= a.nfar) { __hip_atomic_store(mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return; }
        int seen = 0;
        float acc = 0.f;
        const uint32_t ring = old ? 3u : 7u;
        for (int t = 0; t < steps; ++t) {
            const int need = max(1, t - 4 + 2);
            int spins = 0;
            while (seen < need) {
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (seen < need) { __builtin_amdgcn_s_sleep(4); if (++spins > SPIN) { seen = 0x7fffffff; break; } }
            }
            if (seen == 0x7fffffff) break;
            asm volatile("" ::: "memory");
            float m1 = -INFINITY, m2 = -INFINITY;
            for (int k = 0; k < a.farK; ++k) {
                const uint32_t d = 4u + (uint32_t)((k * 5 + h) & 15);
                const float w = hW[((uint32_t)(t - (int)d) & DM) * HS + lb - (uint32_t)(k & 3)];
                m1 = fmaxf(m1, __fsub_rn(__fadd_rn(w, 0.25f), ccy));
            }
            for (int k = 0; k < a.farV; ++k) m2 = __fadd_rn(__fmul_rn(m2 == -INFINITY ? m1 : m2, 0.999f), -0.001f);
            float *rs = res + ((uint32_t)t & ring) * 192u;
            __builtin_amdgcn_ds_fmaxf((lds_float *)(rs + lane), fminf(m1, -5.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            __builtin_amdgcn_ds_fmaxf((lds_float *)(rs + 64 + lane), fminf(m2, -5.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            asm volatile("" ::: "memory");
            __hip_atomic_store(mine, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __hip_atomic_store(mine, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a.out[64 * wave + lane] = acc;
    }
}

int main(int argc, char **argv) {
    const int steps = 6000;   // (a multiple of 4)
    Args a; memset(&a, 0, sizeof a);
    a.steps = steps;
    if (hipMalloc(&a.cells, (size_t)steps * 1024 * 64) != hipSuccess || hipMalloc(&a.prog, 64 * 64) != hipSuccess || hipMalloc(&a.out, 512 * 4) != hipSuccess || hipMalloc(&a.clk, 64) != hipSuccess) return 1;
    struct Cfg { const char *name; int mode, nfar, farK, farV, store, xv; };
    const Cfg cfgs[] = {
        {"A alone, no store", 0, 0, 0, 0, 0, 0},
        {"A alone, store", 0, 0, 0, 0, 1, 0},
        {"A alone, store, +50 dependent ops", 0, 0, 0, 0, 1, 1},
        {"A + B, no store", 1, 0, 0, 0, 0, 0},
        {"A + B, store", 1, 0, 0, 0, 1, 0},
        {"A + B + 5 F (K 6, V 20), store", 2, 5, 6, 20, 1, 0},
        {"A + B + 5 F (K 12, V 40), store", 2, 5, 12, 40, 1, 0},
        {"A + B + 5 F (K 12, V 80), store", 2, 5, 12, 80, 1, 0},
        {"A + B + 3 F (K 12, V 40), store", 2, 3, 12, 40, 1, 0},
        {"OLD: A with t-2 terms + 2 near + 5 F (K 6, V 20), store", 3, 5, 6, 20, 1, 0},
        {"OLD: A with t-2 terms + 2 near + 5 F (K 12, V 40), store", 3, 5, 12, 40, 1, 0},
        {"OLD: A with t-2 terms + 2 near + 5 F (K 12, V 80), store", 3, 5, 12, 80, 1, 0},
        {"OLD: A with t-2 terms + 2 near, no F, store", 3, 0, 0, 0, 1, 0},
    };
    const int ngrid = argc > 1 ? std::min(64, atoi(argv[1])) : 1;   // workgroups (every one runs the same model; timing of block 0)
    for (const Cfg &c : cfgs) {
        a.nfar = c.nfar; a.farK = c.farK; a.farV = c.farV; a.xv = c.xv;
        double best = 1e30, bestw = 0, mhz = 0; int ab = 0;
        for (int rep = 0; rep < 3; ++rep) {
            if (c.mode == 0 && !c.store) hipLaunchKernelGGL((model<0, false>), dim3(ngrid), dim3(512), 0, 0, a);
            else if (c.mode == 0) hipLaunchKernelGGL((model<0, true>), dim3(ngrid), dim3(512), 0, 0, a);
            else if (c.mode == 1 && !c.store) hipLaunchKernelGGL((model<1, false>), dim3(ngrid), dim3(512), 0, 0, a);
            else if (c.mode == 1) hipLaunchKernelGGL((model<1, true>), dim3(ngrid), dim3(512), 0, 0, a);
            else if (c.mode == 2) hipLaunchKernelGGL((model<2, true>), dim3(ngrid), dim3(512), 0, 0, a);
            else hipLaunchKernelGGL((model<3, true>), dim3(ngrid), dim3(512), 0, 0, a);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            unsigned long long h[4]; if (hipMemcpy(h, a.clk, 32, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            const double cyc = (double)h[0] / steps;
            if (cyc < best) { best = cyc; bestw = (double)h[2] / steps; mhz = (double)h[0] / h[1] * 100.0; }
            ab |= (int)h[3];
        }
        printf("%-62s %7.1f cycles/step (%6.1f waiting) = %.3f us at %.0f MHz%s\n", c.name, best, bestw, best / mhz, mhz, ab ? "  ABORTED" : "");
        fflush(stdout);
    }
    return 0;
}
