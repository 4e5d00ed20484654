// Cost of getting 76 MB of host data to the device from a cold process: hipHostMalloc + copy, against a huge-page block
// touched by the writer threads and registered afterwards, against a plain pageable copy.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pin_bench tools/micro/pin_bench.hip -lpthread && /tmp/pin_bench
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t n, int nt) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([=]() { size_t a = n / nt * t, b = t == nt - 1 ? n : n / nt * (t + 1); memset(p + a, t + 1, b - a); });
    for (auto &t : th) t.join();
}
int main() {
    const size_t N = 76u << 20;
    void *d = nullptr;
    hipStream_t s;
    hipSetDevice(0); hipStreamCreate(&s); hipMalloc(&d, N);
    { void *h; hipHostMalloc(&h, 1 << 20, 0); hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); hipHostFree(h); }   // warm copy path
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        void *h; hipHostMalloc(&h, N, 0);
        double t1 = now();
        touch((char *)h, N, 8);
        double t2 = now();
        hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        double t3 = now();
        hipHostFree(h);
        printf("hipHostMalloc %.2f ms, fill (8 threads) %.2f, copy %.2f, free %.2f\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
        for (int huge = 0; huge < 2; ++huge) {
            t0 = now();
            const size_t len = (N + (2u << 20) - 1) / (2u << 20) * (2u << 20);
            void *p = aligned_alloc(2u << 20, len);
            if (huge) madvise(p, len, MADV_HUGEPAGE);
            t1 = now();
            touch((char *)p, N, 8);
            t2 = now();
            hipError_t e = hipHostRegister(p, len, hipHostRegisterDefault);
            t3 = now();
            hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
            double t4 = now();
            hipHostUnregister(p);
            double t5 = now();
            free(p);
            printf("aligned_alloc%s %.2f ms, fill %.2f, register %.2f (%s), copy %.2f, unregister %.2f\n", huge ? "+MADV_HUGEPAGE" : "", t1 - t0, t2 - t1, t3 - t2, hipGetErrorString(e), t4 - t3, t5 - t4);
        }
        t0 = now();
        void *p = malloc(N);
        touch((char *)p, N, 8);
        t1 = now();
        hipMemcpyAsync(d, p, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        t2 = now();
        free(p);
        printf("malloc + fill %.2f ms, pageable copy %.2f\n", t1 - t0, t2 - t1);
    }
    return 0;
}
