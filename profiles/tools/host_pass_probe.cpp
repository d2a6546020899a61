// Where does hf_create's first pass (14 B read + 8 B written per window, 16 threads, 1.1-1.3 ms for 1.5 M windows = ~27 GB/s) lose its time?
// hipcc -O2 -std=c++17 profiles/tools/host_pass_probe.cpp -o /tmp/host_pass_probe && /tmp/host_pass_probe
// Variants of the same loop over arrays allocated the way numpy allocates the caller's (malloc, touched by the main thread):
//   read    sum of the four input arrays                      pageable  + two 4-byte outputs into malloc'd memory
//   pinned  the same, outputs into hipHostMalloc'd memory      table     + the two random byte-table accesses of the key marking
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main() {
    const size_t N = 1527428; const int T = 16;
    uint16_t* cov = (uint16_t*) malloc(N * 2); uint16_t* mq = (uint16_t*) malloc(N * 2); uint16_t* cp = (uint16_t*) malloc(N * 2);
    uint64_t* an = (uint64_t*) malloc(N * 8);
    for (size_t i = 0; i < N; i++) { cov[i] = 15 + (i * 2654435761u >> 28); mq[i] = cov[i]; cp[i] = 0; an[i] = 0; }
    uint32_t *o1 = (uint32_t*) malloc(N * 4), *o2 = (uint32_t*) malloc(N * 4), *p1 = nullptr, *p2 = nullptr;
    memset(o1, 0, N * 4); memset(o2, 0, N * 4);
    hipHostMalloc((void**) &p1, N * 4); hipHostMalloc((void**) &p2, N * 4);
    memset(p1, 0, N * 4); memset(p2, 0, N * 4);
    std::vector<uint8_t> tab(65536 * 9, 0);
    volatile uint64_t sink = 0;
    auto run = [&](const char* name, int mode) {
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int k = 0; k < T; k++) th.emplace_back([&, k] {
                const size_t a = N * k / T, b = N * (k + 1) / T;
                uint64_t s = 0; unsigned xp = 0;
                uint32_t* w1 = mode == 2 || mode == 3 ? p1 : o1; uint32_t* w2 = mode == 2 || mode == 3 ? p2 : o2;
                for (size_t t = a; t < b; t++) {
                    const unsigned cv = cov[t], m = mq[t], c = cp[t], r = (unsigned) (an[t] >> 58);
                    if (mode == 0) { s += cv + m + c + r; continue; }
                    w1[t] = cv | (m << 8) | (c << 16) | (r << 24); w2[t] = cv | (r << 8);
                    if (mode == 3) { uint8_t* cell = tab.data() + ((size_t) ((cv & 255) << 8 | xp)) * 9 + (m & 7); if (!*cell) *cell = 1; xp = cv & 255; }
                }
                sink += s;
            });
            for (auto& t : th) t.join();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        const double bytes = mode == 0 ? N * 14.0 : N * 22.0;
        printf("%-8s best of 5: %.3f ms = %.1f GB/s (thread start-up included)\n", name, best, bytes / best / 1e6);
    };
    run("read", 0); run("pageable", 1); run("pinned", 2); run("table", 3);
    { const auto t0 = std::chrono::steady_clock::now(); std::vector<std::thread> th; for (int k = 0; k < T; k++) th.emplace_back([] {}); for (auto& t : th) t.join();
      printf("16 empty threads: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
    return 0;
}
