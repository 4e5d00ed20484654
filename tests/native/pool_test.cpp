#include "pgm_pool.h"
#include <cstdio>
#include <chrono>
int main() {
    pgm_pool::Pool pool(16);
    std::vector<int> hits(100000);
    long total = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 4000; ++rep) {
        size_t n = 1 + (rep * 7919) % 300;
        std::fill(hits.begin(), hits.begin() + n, 0);
        std::string e = pool.run(n, 1 + rep % 16, [&](size_t i) { hits[i]++; if (rep % 1000 == 3 && i == 2) throw std::runtime_error("x"); });
        for (size_t i = 0; i < n; ++i) if (hits[i] != 1) { printf("BAD rep %d i %zu hits %d\n", rep, i, hits[i]); return 1; }
        if ((rep % 1000 == 3 && n > 2) != !e.empty()) { printf("BAD err rep %d\n", rep); return 1; }
        total += n;
    }
    // nested
    pool.run(8, 16, [&](size_t) { pool.run(4, 16, [&](size_t) {}); });
    printf("ok %ld indices, %.1f us per section\n", total, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 4000);
}
