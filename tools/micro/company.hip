// Company for a latency-bound kernel: occupies N CUs (one 1024-thread workgroup with 100 KB of LDS each) for a given time with one kind
// of work, so that another process's kernel on the remaining CUs shows which shared resource slows it down.
//   company MODE NBLOCKS SECONDS      MODE: alu (dependent FMAs, no memory), mem (streaming reads + writes of a 2 GB buffer),
//                                           poll (agent-scope atomic loads of one word, s_sleep in between), lds (LDS traffic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
__global__ void __launch_bounds__(1024, 1) company(int mode, float4 *buf, size_t n4, int *word, unsigned long long ticks, float *out) {
    __shared__ float lds[25600];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float v = threadIdx.x * 1e-3f;
    lds[threadIdx.x] = v;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 64; ++k) v = __fmaf_rn(v, 0.999f, 0.001f);
        } else if (mode == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 x = buf[i % n4];
                acc.x += x.x; acc.y += x.y;
                buf[(i + n4 / 2) % n4] = make_float4(v, v, v, v);
                i += (size_t)gridDim.x * 1024;
            }
        } else if (mode == 2) {
            v += (float)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(2);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) { lds[(threadIdx.x * 7 + k * 64) % 25600] = v; v += lds[(threadIdx.x + k * 33) % 25600]; }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = v + acc.x + acc.y;
}
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: company alu|mem|poll|lds NBLOCKS SECONDS\n"); return 2; }
    const int mode = !strcmp(argv[1], "alu") ? 0 : !strcmp(argv[1], "mem") ? 1 : !strcmp(argv[1], "poll") ? 2 : 3;
    const int nb = atoi(argv[2]);
    const double secs = atof(argv[3]);
    float4 *buf; int *word; float *out;
    const size_t n4 = (size_t)1 << 27;   // 2 GB
    if (hipMalloc(&buf, n4 * 16) != hipSuccess || hipMalloc(&word, 64) != hipSuccess || hipMalloc(&out, (size_t)nb * 4096) != hipSuccess) return 1;
    hipMemset(buf, 0, n4 * 16); hipMemset(word, 0, 64);
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipLaunchKernelGGL(company, dim3(nb), dim3(1024), 0, 0, mode, buf, n4, word, 2000000ull /* 20 ms */, out);
        hipDeviceSynchronize();
        ++launches;
    }
    printf("company %s: %d launches of 20 ms on %d CUs\n", argv[1], launches, nb);
    return 0;
}
