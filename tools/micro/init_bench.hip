// What pgm_ctx_create's ~80-250 ms are made of (GPU box): each HIP start-up step timed in a fresh process.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/init_bench tools/micro/init_bench.hip && tools/micro/bin/init_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k() {}
int main() {
    double t0 = now(), t;
    int n = 0; hipGetDeviceCount(&n);            t = now(); printf("hipGetDeviceCount (runtime start-up) %.1f ms\n", t - t0); t0 = t;
    hipSetDevice(0);                              t = now(); printf("hipSetDevice %.1f ms\n", t - t0); t0 = t;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); t = now(); printf("hipGetDeviceProperties %.1f ms\n", t - t0); t0 = t;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); t = now(); printf("hipStreamCreate %.1f ms\n", t - t0); t0 = t;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); t = now(); printf("first launch (code object load of this tiny binary) %.1f ms\n", t - t0); t0 = t;
    void *h, *d; hipHostMalloc(&h, 1 << 20, 0); hipMalloc(&d, 1 << 20); t = now(); printf("hipHostMalloc + hipMalloc 1 MB %.1f ms\n", t - t0); t0 = t;
    hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t = now(); printf("first H2D copy %.1f ms\n", t - t0); t0 = t;
    hipMemcpyAsync(h, d, 1 << 20, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t = now(); printf("first D2H copy %.1f ms\n", t - t0); t0 = t;
    return 0;
}
