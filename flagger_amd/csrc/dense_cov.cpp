// dense_cov — a bam2cov-like synthetic coverage track at REAL row density: where flagger_amd/synth.py writes one run per
// window, a coverage track computed from alignments (bam2cov -> .cov.gz, the input of hmm_flagger: track_reader.c:751-818, chunk.c:393-483)
// changes value every few hundred bases — tens of millions of rows and gigabytes of text for a human diploid assembly, in ONE DEFLATE
// stream.  Test / bench infrastructure (like synth.py): the loader under test is hf_io.cpp.
//
//   dense_cov <out.cov|out.cov.gz> <seed> <min_run> <max_run> <only: -1 = all contigs | index> <len_0> <len_1> ...
//
// Every contig draws from its own generator (seed, contig index): a file with `only = i` holds exactly the rows contig i has in the full
// file — the slow per-base oracle loader is run on such single-contig files and compared with the product loader's windows of that contig.
// Rows: 1-based inclusive `start end cov mapq clip annotation region`; runs of min_run..max_run bases (the last one cut at the contig's
// end), so rows straddle window and chunk boundaries everywhere.  Coverage: a mean-reverting walk around 20 (40x diploid HiFi, one
// haplotype), with stretches of another regime every few Mb (near-zero = erroneous, half = duplicated, double = collapsed) so that the
// HMM has something to label; mapq = the part of the coverage with a high mapping quality, clip = clipped alignments (both <= cov).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

namespace {
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) { next(); next(); }
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    uint32_t below(uint32_t n) { return (uint32_t) ((next() >> 33) % n); }       // (bias of < 2^-31 n: irrelevant here)
    double unit() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); }
};
struct Out {
    gzFile gz = nullptr; FILE* fp = nullptr;
    std::vector<char> buf; size_t n = 0; uint64_t bytes = 0;
    bool open(const std::string& path) {
        buf.resize(8u << 20);
        if (path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0) { gz = gzopen(path.c_str(), "w6h")      /* as the reference writes a .cov.gz: level 6, Z_HUFFMAN_ONLY (ptBlock.c:2271) */; if (gz) gzbuffer(gz, 1 << 20); return gz != nullptr; }
        fp = std::fopen(path.c_str(), "wb"); return fp != nullptr;
    }
    void flush() { if (!n) return; if (gz) gzwrite(gz, buf.data(), (unsigned) n); else std::fwrite(buf.data(), 1, n, fp); bytes += n; n = 0; }
    char* room(size_t want) { if (n + want > buf.size()) flush(); return buf.data() + n; }
    void close() { flush(); if (gz) gzclose(gz); if (fp) std::fclose(fp); }
};
inline char* put_uint(char* p, uint32_t v) {
    char tmp[12]; int k = 0;
    do { tmp[k++] = (char) ('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}
}

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: dense_cov <out.cov[.gz]> <seed> <min_run> <max_run> <only|-1> <contig length> ...\n"); return 2; }
    const std::string path = argv[1];
    const uint64_t seed = std::strtoull(argv[2], nullptr, 10);
    const int min_run = std::atoi(argv[3]), max_run = std::atoi(argv[4]), only = std::atoi(argv[5]);
    if (min_run < 1 || max_run < min_run) { std::fprintf(stderr, "dense_cov: bad run lengths\n"); return 2; }
    Out out;
    if (!out.open(path)) { std::fprintf(stderr, "dense_cov: cannot open %s\n", path.c_str()); return 1; }
    {   // the header layout of the reference's simulator (programs/src/simulate_coverage_data.py:146-168), as synth.py writes it
        const char* h = "#annotation:len:2\n#annotation:name:0:no_annotation\n#annotation:name:1:whole_genome\n#region:len:1\n#region:coverage:0:20\n"
                        "#avg_alignment_len:15000\n#start-only:false\n";
        char* p = out.room(std::strlen(h)); std::memcpy(p, h, std::strlen(h)); out.n += std::strlen(h);
    }
    uint64_t rows = 0, bases = 0;
    for (int ci = 0; ci + 6 < argc; ci++) {
        if (only >= 0 && ci != only) continue;
        const long len = std::atol(argv[6 + ci]);
        if (len <= 0 || len > 2000000000L) { std::fprintf(stderr, "dense_cov: bad contig length\n"); return 2; }
        Rng rng(seed * 1000003ull + (uint64_t) ci);
        {
            char* p = out.room(64);
            const int k = std::snprintf(p, 64, ">hap_ctg%d %ld\n", ci, len);
            out.n += (size_t) k;
        }
        long pos = 1;                       // 1-based start of the next run
        int cov = 20, regime_mean = 20;
        long regime_left = 2000000 + (long) rng.below(6000000);      // bases until the regime changes
        while (pos <= len) {
            long run = min_run + (long) rng.below((uint32_t) (max_run - min_run + 1));
            if (pos + run - 1 > len) run = len - pos + 1;
            // the regime: mostly haploid coverage; now and then a stretch of something else
            regime_left -= run;
            if (regime_left <= 0) {
                if (regime_mean != 20) { regime_mean = 20; regime_left = 1500000 + (long) rng.below(8000000); }
                else {
                    const uint32_t k = rng.below(10);
                    regime_mean = k < 3 ? 1 : (k < 6 ? 10 : (k < 9 ? 40 : 62));
                    regime_left = 20000 + (long) rng.below(250000);
                }
            }
            // mean-reverting walk: a step of -2..2 pulled towards the regime's mean
            int step = (int) rng.below(5) - 2;
            if (cov < regime_mean - 3) step += 2; else if (cov > regime_mean + 3) step -= 2;
            else if (cov < regime_mean && rng.below(3) == 0) step += 1; else if (cov > regime_mean && rng.below(3) == 0) step -= 1;
            cov += step;
            if (cov < 0) cov = 0;
            if (cov > 250) cov = 250;
            int mapq = cov;
            if (rng.below(16) == 0 && cov > 0) mapq = (int) rng.below((uint32_t) cov + 1);     // a low-mapq stretch
            int clip = 0;
            if (rng.below(40) == 0 && cov > 0) clip = 1 + (int) rng.below((uint32_t) (cov < 6 ? cov : 6));
            char* p = out.room(96);
            char* q = p;
            q = put_uint(q, (uint32_t) pos); *q++ = '\t';
            q = put_uint(q, (uint32_t) (pos + run - 1)); *q++ = '\t';
            q = put_uint(q, (uint32_t) cov); *q++ = '\t';
            q = put_uint(q, (uint32_t) mapq); *q++ = '\t';
            q = put_uint(q, (uint32_t) clip); *q++ = '\t';
            *q++ = '2'; *q++ = '\t'; *q++ = '0'; *q++ = '\n';                   // annotation "whole_genome", region 0
            out.n += (size_t) (q - p);
            pos += run; rows++; bases += (uint64_t) run;
        }
    }
    out.close();
    std::printf("{\"rows\": %llu, \"bases\": %llu, \"text_bytes\": %llu}\n", (unsigned long long) rows, (unsigned long long) bases, (unsigned long long) out.bytes);
    return 0;
}
