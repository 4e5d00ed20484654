// hwid_probe.hip — where do the four wavefronts of a 256-thread workgroup land?  Prints, for a launch shaped like the fill
// kernel (2 workgroups per CU through the LDS footprint), the SIMD and wave-slot ids (HW_REG_HW_ID) of every wavefront.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_probe tools/micro/hwid_probe.hip && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

__global__ __launch_bounds__(256, 2) void probe(unsigned *out, int spin) {
    __shared__ float pad[18000];   // 72 KB: two workgroups per CU
    const int w = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float acc = 0;
    for (int i = 0; i < spin; ++i) { pad[(threadIdx.x * 7 + i) % 18000] = acc; acc += pad[(threadIdx.x + i * 13) % 18000]; }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + w) * 2] = hw; out[(blockIdx.x * 4 + w) * 2 + 1] = xcc + (acc == 12345.f); }
}

int main() {
    const int nb = 512;
    unsigned *d; hipMalloc(&d, nb * 8 * sizeof(unsigned));
    std::vector<unsigned> h(nb * 8);
    probe<<<nb, 256>>>(d, 20000);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::string, int> pat, slots;
    std::map<unsigned, int> per_cu;
    for (int b = 0; b < nb; ++b) {
        char s[64], t[64]; int o = 0, p = 0;
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 2];
            o += snprintf(s + o, sizeof s - o, "%u", (hw >> 4) & 3);
            p += snprintf(t + p, sizeof t - p, "%u", hw & 15);
        }
        pat[s]++; slots[t]++;
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        per_cu[(xcc << 16) | (hw & 0xff00)]++;
        if (b < 24) printf("block %3d  xcc %u se %u sh %u cu %2u  simd %s  slot %s\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, s, t);
    }
    printf("simd patterns (wave 0..3):"); for (auto &kv : pat) printf("  %s x%d", kv.first.c_str(), kv.second); printf("\n");
    printf("slot patterns (wave 0..3):"); for (auto &kv : slots) printf("  %s x%d", kv.first.c_str(), kv.second); printf("\n");
    std::map<int, int> hist; for (auto &kv : per_cu) hist[kv.second]++;
    printf("workgroups per (xcc,se,sh,cu):"); for (auto &kv : hist) printf("  %d blocks on %d CUs", kv.first, kv.second); printf("\n");
    return 0;
}
