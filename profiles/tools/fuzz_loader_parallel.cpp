// ASan / UBSan fuzz of the loader with the parallel decoders forced on (hf_io.cpp spec_*): a Huffman-only .cov.gz with one to three flipped
// bits or a random truncation must load to the SAME windows as the intact file or fail with an error — never crash, hang or load something else.
//   flagger_amd/csrc/dense_cov /tmp/fz.cov.gz 9 20 120 -1 1200000 300000 && \
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -I include -o /tmp/fzl profiles/tools/fuzz_loader_parallel.cpp \
//       flagger_amd/csrc/hf_io.cpp flagger_amd/csrc/hf_summary.cpp -lz -lpthread && /tmp/fzl /tmp/fz.cov.gz 400
#include "hmm_flagger_io.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
static std::vector<unsigned char> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<unsigned char> v; unsigned char b[65536]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v; }
static unsigned long long digest(hfio_table* t) {
    hf_windows w; hfio_windows(t, &w);
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*) p; for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; } };
    mix(w.cov, (size_t) w.n_windows * 2); mix(w.mapq, (size_t) w.n_windows * 2); mix(w.clip, (size_t) w.n_windows * 2); mix(w.annot, (size_t) w.n_windows * 8);
    mix(w.chunk_off, ((size_t) w.n_chunks + 1) * 8);
    return h;
}
int main(int argc, char** argv) {
    const std::vector<unsigned char> base = slurp(argv[1]);
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    setenv("HF_IO_PARALLEL", "4", 1); setenv("HF_IO_PARALLEL_MIN", "0", 1); setenv("HF_IO_PIECE", "30000", 1); setenv("HF_IO_PROBE", "50000", 1);
    hfio_table* t0 = hfio_load(argv[1], 1000000, 4000);
    if (!t0) { printf("the intact file does not load: %s\n", hfio_last_error()); return 1; }
    const unsigned long long want = digest(t0);
    hfio_destroy(t0);
    std::mt19937_64 g(99);
    int loaded_same = 0, failed = 0, wrong = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<unsigned char> v = base;
        if (it % 3 == 2) v.resize(64 + g() % (v.size() - 64));
        else for (int k = 0, n = 1 + (int) (g() % 3); k < n; k++) v[20 + g() % (v.size() - 20)] ^= (unsigned char) (1u << (g() % 8));
        const std::string path = "/tmp/fzl_case.cov.gz";
        FILE* f = fopen(path.c_str(), "wb"); fwrite(v.data(), 1, v.size(), f); fclose(f);
        hfio_table* t = hfio_load(path.c_str(), 1000000, 4000);
        if (!t) { failed++; continue; }
        if (digest(t) == want) loaded_same++; else { wrong++; printf("case %d loaded DIFFERENT windows\n", it); }
        hfio_destroy(t);
    }
    printf("%d cases: %d failed with an error, %d loaded the intact file's windows, %d loaded something else\n", iters, failed, loaded_same, wrong);
    return wrong != 0;
}
