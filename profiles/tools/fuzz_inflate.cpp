// ASan / UBSan fuzz of the loader's DEFLATE decoder (flagger_amd/csrc/hf_inflate.h): garbage behind a gzip header, and zlib streams of four levels
// with one to four flipped bits and random truncation; exact-size input buffers so that any over-read is seen.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I flagger_amd/csrc -o /tmp/fz profiles/tools/fuzz_inflate.cpp -lz && /tmp/fz
#include "hf_inflate.h"
#include <cstdio>
#include <vector>
#include <random>
#include <zlib.h>
int main() {
    std::mt19937_64 rng(12345);
    std::vector<uint8_t> out(1 << 20);
    static hfz::Inflater z;
    long ok = 0, err = 0;
    // 1) pure garbage after a gzip header   2) valid streams with random mutations
    std::vector<uint8_t> base(200000);
    for (auto& b : base) b = (uint8_t) ("ACGT\t\n0123456789"[rng() % 16]);
    std::vector<std::vector<uint8_t>> streams;
    for (int lv : {0, 1, 6, 9}) {
        uLongf cl = compressBound(base.size()) + 64; std::vector<uint8_t> c(cl);
        z_stream s{}; deflateInit2(&s, lv, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
        s.next_in = base.data(); s.avail_in = base.size(); s.next_out = c.data(); s.avail_out = cl; deflate(&s, Z_FINISH); c.resize(s.total_out); deflateEnd(&s);
        streams.push_back(c);
    }
    for (int it = 0; it < 200000; it++) {
        std::vector<uint8_t> in;
        if (it % 2 == 0) {
            size_t n = 1 + rng() % 300;
            in = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
            for (size_t i = 0; i < n; i++) in.push_back((uint8_t) rng());
        } else {
            in = streams[rng() % streams.size()];
            int nm = 1 + rng() % 4;
            for (int k = 0; k < nm; k++) in[rng() % in.size()] ^= (uint8_t) (1u << (rng() % 8));
            if (rng() % 4 == 0) in.resize(rng() % in.size());
        }
        // exact-size allocation so that ASan sees any over-read
        std::vector<uint8_t> exact(in);
        z.in = exact.data(); z.in_len = exact.size(); z.pos = 0;
        int rc = z.read_gzip_header();
        size_t tot = 0;
        while (rc == hfz::OK) {
            size_t got = 0;
            std::vector<uint8_t> o(tot + 70000);
            memcpy(o.data(), out.data(), tot);
            rc = z.run(o.data() + tot, o.size() - tot, tot, &got);
            if (tot + got > out.size()) out.resize((tot + got) * 2);
            memcpy(out.data() + tot, o.data() + tot, got);
            tot += got;
            if (tot > (8u << 20)) break;
        }
        if (rc == hfz::END_OF_MEMBER) { uint32_t c, s; z.read_gzip_trailer(&c, &s); ok++; } else err++;
    }
    printf("ended %ld, errors %ld\n", ok, err);
}
