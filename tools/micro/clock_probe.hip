// Micro-probe: does a lone wavefront's instruction rate depend on how many other single-wavefront workgroups run?
// N workgroups of 64 threads (32 KB LDS each, like the fill workers) run a dependent chain of VALU ops (+ optional LDS
// round trips); prints wall time per iteration and the shader-clock / real-time-clock ratio.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(64) probe(float *out, unsigned long long *clk, int iters, int use_lds) {
    __shared__ float buf[8192];
    float v = threadIdx.x * 1e-3f;
    buf[threadIdx.x] = v;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v = __fadd_rn(__fmul_rn(v, 0.999f), 0.001f);
        if (use_lds) { buf[(threadIdx.x + i) & 8191] = v; v += buf[(threadIdx.x + i + 1) & 8191]; }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    float *out; unsigned long long *clk;
    hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&clk, 4096 * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 200000;
    for (int lds = 0; lds < 2; ++lds)
        for (int n : {1, 4, 16, 24, 32, 64, 128, 256, 512, 1024}) {
            hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, out, clk, 1000, lds);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, out, clk, iters, lds);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("lds=%d n=%4d  %.3f ms  %.2f ns/iter  cyclecounter/realtime(100MHz)=%.3f -> %.0f MHz\n", lds, n, ms, ms * 1e6 / iters,
                   (double)h[0] / h[1], (double)h[0] / h[1] * 100.0);
            fflush(stdout);
        }
    return 0;
}
