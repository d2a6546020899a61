// hf_io.cpp — window table (SoA) + the file formats either side of the hot path
// (include/hmm_flagger_io.h).  Citations: mobinasri/flagger programs/submodules/.
#include "../../include/hmm_flagger_io.h"
#include "../../include/hmm_flagger_summary.h"
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include "hf_inflate.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
thread_local std::string g_io_err;
constexpr int kMaxCoverage = 250;   // chunk.c:8

struct ChunkMeta { std::string ctg; int32_t ctg_len, s, e; };

// window accumulator of one chunk being filled (Chunk.windowSum*, chunk.h:24-33)
// value histogram for the mode rules (common.c:407-427) that remembers which entries it touched: a window usually sees
// one value, and clearing / scanning 125 counters per window was a third of the loader's time
template <int NV>
struct SmallHist {
    int h[NV] = {0};
    int touched[NV];
    int nt = 0;
    void add(int v, int n) { if (h[v] == 0) touched[nt++] = v; h[v] += n; }
    void reset() { for (int i = 0; i < nt; i++) h[touched[i]] = 0; nt = 0; }
    int mode(int minv) const {                // the largest count, the lowest value on ties; minv when nothing was added
        int best = -1, maxc = 0;
        for (int i = 0; i < nt; i++) {
            const int v = touched[i], c = h[v];
            if (c > maxc || (c == maxc && c > 0 && v < best)) { best = v; maxc = c; }
        }
        return best < 0 ? minv : minv + best;
    }
};
struct WindowAcc {
    int n = 0;                      // bases in the open window (windowItr + 1)
    double cov = 0, mapq = 0, clip = 0;
    uint64_t flag = 0;
    SmallHist<101> reg; SmallHist<12> tru, pre;
    void reset() { n = 0; cov = mapq = clip = 0; flag = 0; reg.reset(); tru.reset(); pre.reset(); }
};

// n repeated additions of v onto sum, as the reference does per base (chunk.c:459-461); when both are
// integers below 2^53 every partial sum is exact, so one multiply-add gives the identical double
// (whole(): v == floor(v) without the libm call a baseline x86-64 build makes of floor — a conversion there and back; only asked of |v| < 2^52)
inline bool whole(double v) { return std::fabs(v) < 4503599627370496.0 && (double) (long long) v == v; }
inline void add_run(double& sum, double v, int n) {
    const double lim = 4503599627370496.0;  // 2^52
    if (whole(v) && whole(sum) && std::fabs(v) * n + std::fabs(sum) < lim) { sum += v * n; return; }
    for (int i = 0; i < n; i++) sum += v;
}

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

struct hfio_table {
    std::vector<std::string> annotation_names;
    std::vector<int32_t> region_coverages;
    int32_t n_labels = 0, avg_alignment_len = 0, chunk_len = 0, window_len = 0;
    bool truth_available = false, prediction_available = false, start_only = false;
    std::vector<ChunkMeta> chunks;
    std::vector<int64_t> chunk_off{0};
    std::vector<int32_t> chunk_s, chunk_e, chunk_ctg_len;
    std::vector<uint16_t> cov, mapq, clip;
    std::vector<uint64_t> annot;
    std::vector<int8_t> truth, prediction;

    void push_window(WindowAcc& a) {                      // chunk.c:393-441
        if (a.n == 0) return;
        double c, m, k;
        if (start_only) {
            c = a.cov * window_len / a.n; m = a.mapq * window_len / a.n; k = a.clip * window_len / a.n;
        } else {
            c = a.cov / a.n; m = a.mapq / a.n; k = a.clip / a.n;
        }
        cov.push_back((uint16_t) (kMaxCoverage < std::round(c) ? kMaxCoverage : std::round(c)));
        mapq.push_back((uint16_t) (kMaxCoverage < std::round(m) ? kMaxCoverage : std::round(m)));
        clip.push_back((uint16_t) (kMaxCoverage < std::round(k) ? kMaxCoverage : std::round(k)));
        const int region = a.reg.mode(0);
        annot.push_back((a.flag & 0x03FFFFFFFFFFFFFFULL) | ((uint64_t) region << 58));   // ptBlock.c:300-304
        truth.push_back((int8_t) a.tru.mode(-1));
        prediction.push_back((int8_t) a.pre.mode(-1));
        a.reset();
    }
    void close_chunk(const ChunkMeta& cm) {
        if ((int64_t) cov.size() == chunk_off.back()) return;     // no window: the reference never makes such a chunk
        chunks.push_back(cm);
        chunk_s.push_back(cm.s); chunk_e.push_back(cm.e); chunk_ctg_len.push_back(cm.ctg_len);
        chunk_off.push_back((int64_t) cov.size());
    }
};

namespace {

std::string file_ext(const std::string& p) {                      // common.c:51-66
    int len = (int) p.size(), i = len - 1;
    for (; 0 <= i; i--)
        if (p[i] == '.') {
            const char* t = p.c_str() + i;
            if (std::strcmp(t, ".gz") != 0 && std::strcmp(t, ".tar") != 0 && std::strcmp(t, ".tar.gz") != 0 &&
                std::strcmp(t, ".zip") != 0) break;
        }
    return p.substr(i + 1);
}

uint64_t annot_flag_of(const char* s) {                           // ptBlock.c:225-236
    uint64_t flag = 0;
    const char* p = s;
    while (*p) {
        const int idx = std::atoi(p);
        if (0 < idx && idx <= 64) flag |= 1ULL << (idx - 1);
        const char* q = std::strchr(p, ',');
        if (!q) break;
        p = q + 1;
    }
    return flag;
}

bool starts_with(const char* s, const char* pre) { return std::strncmp(s, pre, std::strlen(pre)) == 0; }
const char* field_after(const char* line, int n_colons) {         // pointer just after the n-th ':'
    const char* p = line;
    for (int i = 0; i < n_colons; i++) { p = std::strchr(p, ':'); if (!p) return nullptr; p++; }
    return p;
}


// ---- reading: one thread inflates line-aligned blocks of ~4 MiB, a few threads turn the blocks' lines into records
// (everything that is text work: splitting, numbers, annotation flags), the caller consumes the blocks in file order and does
// what is order-dependent (contigs, chunks, windows).  Text -> numbers was 3/4 of a single-threaded loader's time. ----
struct CovRec {
    enum Kind : uint8_t { ROW = 0, CONTIG = 1, HEADER = 2, SHORT_ROW = 3 };
    int32_t s, e;                       // ROW: 0-based inclusive; CONTIG / HEADER: offset and length of the line in the block's text
    double cov, mapq, clip;
    uint64_t flag;
    int16_t region, truth, pred;        // clamped / shifted as the consumer stores them
    uint8_t kind;
};
void parse_block(char* text, size_t len, std::vector<CovRec>& out);      // below (needs the numeric helpers)

class BlockReader {
  public:
    struct Block {
        std::vector<char> text;         // [carry of the block before | this block's bytes]
        size_t off = 0, len = 0;        // the lines to parse: text[off .. off + len), whole lines ('\n'-terminated except the file's last)
        size_t data0 = 0, data_len = 0; // this block's own bytes (CRC-32 of a gzip member is over these)
        std::vector<CovRec> recs;       // CovRec offsets are relative to base()
        uint32_t crc = 0;               // crc32 of the block's own bytes (computed by the thread that parses it)
        bool member_end = false;        // a gzip member ends with this block: its CRC-32 and ISIZE
        uint32_t want_crc = 0, want_isize = 0;
        int state = 0;                  // 0 empty, 1 filled (text), 2 being parsed, 3 parsed, 4 being consumed
        bool last = false;              // no data: the end marker
        char* base() { return text.data() + off; }
    };
    // own decoder over the mapped file (gzip members or plain text); zlib's gzread when the file cannot be mapped (or HF_IO_ZLIB=1)
    explicit BlockReader(const char* path) {
        const char* force = std::getenv("HF_IO_ZLIB");
        if (!(force && force[0] == '1')) map_file(path);
        if (!map_) {
            gz_ = gzopen(path, "r");                                 // also reads uncompressed text
            if (!gz_) { open_failed_ = true; return; }
            gzbuffer(gz_, 1 << 20);
        }
        unsigned hw = std::thread::hardware_concurrency();
        {   // decoders inside one stream (spec_* below): as many as half the threads, at most eight
            if (const char* e = std::getenv("HF_IO_PARALLEL")) { if (e[0] == '0') spec_off_ = true; else { const int v = std::atoi(e); if (v >= 2 && v <= 32) spec_k_ = (size_t) v; } }
            else spec_k_ = hw >= 16 ? 8 : (hw >= 8 ? 4 : (hw >= 4 ? 2 : 1));
            if (spec_k_ < 2) spec_off_ = true;
            if (const char* e = std::getenv("HF_IO_PARALLEL_MIN")) spec_min_ = (size_t) std::atoll(e);
            if (const char* e = std::getenv("HF_IO_PIECE")) { const long long v = std::atoll(e); if (v >= 4096) spec_piece_ = (size_t) v; }
            if (const char* e = std::getenv("HF_IO_PROBE")) { const long long v = std::atoll(e); if (v >= 1) kSpecProbe = (size_t) v; }
        }
        int nw = hw >= 16 ? 6 : (hw >= 8 ? 4 : (hw >= 4 ? 2 : 1));   // (six: with the decoders running ahead, four parsers were the next stage to fill up)
        if (const char* e = std::getenv("HF_IO_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 32) nw = v; }
        prod_ = std::thread([this] { produce(); });
        for (int i = 0; i < nw; i++) work_.emplace_back([this] { parse_loop(); });
    }
    ~BlockReader() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        if (prod_.joinable()) prod_.join();
        for (auto& t : work_) if (t.joinable()) t.join();
        spec_abort_.store(true);
        for (auto& r : spec_round_) if (r && r->th.joinable()) r->th.join();
        if (gz_) gzclose(gz_);
        if (map_ && !map_is_static_) munmap(const_cast<uint8_t*>(map_), map_len_);
        if (fd_ >= 0) close(fd_);
    }
    bool open_failed() const { return open_failed_; }
    // the next block in file order, parsed; nullptr at the end of the file.  The block handed out before is recycled.
    Block* next() {
        std::unique_lock<std::mutex> g(m_);
        if (held_ >= 0) { blk_[held_].state = 0; held_ = -1; cv_.notify_all(); }
        const auto w0 = std::chrono::steady_clock::now();
        cv_.wait(g, [this] { return blk_[cons_].state == 3; });
        wait_cons_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        Block* b = &blk_[cons_];
        if (b->last) return nullptr;
        // gzip members: CRC-32 and length of what was decoded, block by block (crc32_combine)
        run_crc_ = run_len_ ? (uint32_t) crc32_combine(run_crc_, b->crc, (z_off_t) b->data_len) : b->crc;
        run_len_ += b->data_len;
        if (b->member_end) {
            if (run_crc_ != b->want_crc || (uint32_t) run_len_ != b->want_isize) { failed_ = true; return nullptr; }
            run_crc_ = 0; run_len_ = 0;
        }
        b->state = 4; held_ = cons_; cons_ = (cons_ + 1) % kN;
        return b;
    }
    // true when the stream ended on an error (truncated or corrupt .cov.gz, CRC mismatch) instead of its end
    bool failed() const { return failed_; }
    // seconds the consumer waited for a parsed block / the producer (inflate) waited for a free one: which stage bounds the pipeline (HF_IO_TRACE)
    double consumer_wait() const { return wait_cons_; }
    double producer_wait() const { return wait_prod_; }

  private:
    double wait_cons_ = 0.0, wait_prod_ = 0.0;
    size_t released_ = 0;           // bytes of the mapped file already handed back (madvise)
    static constexpr size_t kCap = 4u << 20;
    static constexpr size_t kHist = 32768;
    static constexpr int kN = 8;
    void map_file(const char* path) {
        fd_ = open(path, O_RDONLY);
        if (fd_ < 0) return;
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd_); fd_ = -1; return; }
        map_len_ = (size_t) st.st_size;
        if (map_len_ == 0) { map_ = reinterpret_cast<const uint8_t*>(""); map_is_static_ = true; return; }
        void* p = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (p == MAP_FAILED) { close(fd_); fd_ = -1; map_len_ = 0; return; }
        madvise(p, map_len_, MADV_SEQUENTIAL);
        map_ = static_cast<const uint8_t*>(p);
    }
    // up to `want` more bytes of the file's text behind b.text[have): returns what was produced; sets eof / failed_ / member_end
    size_t fetch(Block& b, size_t have, size_t hist, size_t want, bool& eof) {
        if (gz_) {
            size_t got = 0;
            while (got < want) {                                    // gzread returns short counts only at the end of the stream
                const int r = gzread(gz_, b.text.data() + have + got, (unsigned) (want - got));
                if (r < 0) { failed_ = true; eof = true; break; }   // corrupt deflate data, I/O error
                if (r == 0) {                                       // end of data: a stream cut before its trailer is an error, not EOF
                    int errnum = Z_OK;
                    (void) gzerror(gz_, &errnum);
                    if (errnum != Z_OK && errnum != Z_STREAM_END) failed_ = true;
                    eof = true; break;
                }
                got += (size_t) r;
            }
            return got;
        }
        if (!gzip_) {                                               // plain text
            size_t n = map_len_ - inf_.pos;
            if (n > want) n = want;
            std::memcpy(b.text.data() + have, map_ + inf_.pos, n);
            inf_.pos += n;
            if (inf_.pos == map_len_) eof = true;
            return n;
        }
        if (spec_on_) {                                             // several decoders inside the member (below): their pieces, in order
            const size_t n = spec_fetch(reinterpret_cast<uint8_t*>(b.text.data()) + have, want);
            if (n || spec_on_) return n;
            // (the speculative section is over — the member's last block, or a fall-back: the one decoder goes on where it was left)
        }
        size_t got = 0;
        int rc = inf_.run(reinterpret_cast<uint8_t*>(b.text.data()) + have, want, hist, &got);
        if (rc == hfz::AT_BOUNDARY) {                               // the probe: a block boundary behind the member's first megabytes
            inf_.stop_out = (uint64_t) -1;
            if (!inf_.saw_match) spec_begin(inf_.bit_pos(), inf_.total_out);   // no match so far: go parallel from here (total_out counts this call's bytes)
            if (got) return got;
            const size_t n = spec_on_ ? spec_fetch(reinterpret_cast<uint8_t*>(b.text.data()) + have, want) : 0;
            if (n || spec_on_) return n;
            rc = inf_.run(reinterpret_cast<uint8_t*>(b.text.data()) + have, want, hist, &got);   // (matches, or it did not start: the one decoder)
        }
        // the mapped file behind the decoder is not needed again: hand its pages back every 32 MiB, or a multi-GB .cov.gz of a real
        // bam2cov track sits in this process's resident set to the end (round 6: 450 MB of a 577 MB peak on the dense synthetic track)
        if (!map_is_static_ && inf_.pos > released_ + ((size_t) 32 << 20)) {
            const size_t upto = (inf_.pos - ((size_t) 1 << 20)) & ~(size_t) 4095;      // (whole pages, a margin behind the bit buffer)
            if (upto > released_) { madvise(const_cast<uint8_t*>(map_) + released_, upto - released_, MADV_DONTNEED); released_ = upto; }
        }
        if (rc == hfz::END_OF_MEMBER) {
            uint32_t crc = 0, isz = 0;
            if (inf_.read_gzip_trailer(&crc, &isz) != hfz::OK) { failed_ = true; eof = true; return got; }
            b.member_end = true; b.want_crc = crc; b.want_isize = isz;
            // another member (concatenated gzip files), or the end; anything else after a member is ignored, as gzread does
            const int nm = inf_.next_member();
            if (nm == hfz::ERR_TRUNCATED) { failed_ = true; eof = true; }      // the file ends inside the next member's header
            else if (nm != hfz::OK) eof = true;
            else spec_member_start();
        } else if (rc != hfz::OK) { failed_ = true; eof = true; }
        return got;
    }
    // ------------------------------------------------------------------------------------------
    // Several decoders inside ONE DEFLATE stream (round 6).  The loader is bound by the one thread that inflates.  A
    // .cov.gz as the reference writes it is Z_HUFFMAN_ONLY (ptBlock.c:2271): literals only, no history to resolve — a decoder can start at
    // any block boundary.  After a member's first megabytes have shown no match, the stream is cut into pieces of kSpecPiece compressed bytes:
    // the first piece of a round starts at a KNOWN boundary, every other one searches the first bit position behind its nominal start at
    // which a complete dynamic block header parses AND whose block decodes to printable text; each decodes up to the start of the next.
    // Nothing is taken on trust: a round counts only if every piece ended exactly where the next began; a match symbol, a piece that
    // cannot find or reach its neighbour, any decode error — the one decoder takes over at the last verified boundary (the text before it
    // is its history); and CRC-32 and length of the member are checked over everything as before.
    // HF_IO_PARALLEL=0 switches it off, HF_IO_PARALLEL_MIN=<bytes> sets the smallest member it is tried on (default 32 MiB), HF_IO_PIECE=<bytes>.
    // ------------------------------------------------------------------------------------------
    size_t kSpecProbe = 2u << 20;                                   // text a member must have produced without a match (HF_IO_PROBE: tests)
    struct SpecPiece {
        std::atomic<size_t> start_bit{(size_t) -1};                 // -1: not searched yet, -2: none found
        size_t end_bit = 0;
        std::vector<uint8_t> out; size_t n = 0;
        int rc = hfz::ERR_DATA;
    };
    struct SpecRound {
        std::vector<std::unique_ptr<SpecPiece>> pc;
        std::thread th;
        size_t start_bit = 0, end_bit = 0, good = 0;               // good: pieces that form a verified chain from start_bit
        bool member_end = false, failed = false;
    };
    static bool text_like(const uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; i++) { const uint8_t c = p[i]; if (!((c >= 0x20 && c < 0x7f) || c == '\t' || c == '\n' || c == '\r')) return false; }
        return true;
    }
    void spec_member_start() {
        spec_want_ = false; inf_.saw_match = false; inf_.stop_bit = (size_t) -1; inf_.stop_out = (uint64_t) -1;
        if (spec_off_ || !map_) return;
        if (map_len_ - inf_.pos >= spec_min_) { spec_want_ = true; inf_.stop_out = kSpecProbe; }
    }
    void spec_decode_piece(SpecRound* R, size_t k, size_t nominal_bit, size_t round_end_bit) {
        SpecPiece& P = *R->pc[k];
        hfz::Inflater z;
        z.in = map_; z.in_len = map_len_; z.literal_only = true;
        P.out.resize(spec_piece_ * 3 + (1u << 16));
        size_t start = nominal_bit;
        if (k == 0) z.seek_bit(start, 0);
        else {
            // the first bit position at which a whole dynamic header parses and the block behind it is text
            const size_t limit = std::min(map_len_ * 8, nominal_bit + spec_piece_ * 8);
            bool found = false;
            for (size_t b = nominal_bit; b < limit && !spec_abort_.load(std::memory_order_relaxed); b++) {
                if (!z.try_dynamic_header_at(b)) continue;
                z.stop_bit = b + 1;                                 // (this block only)
                size_t got = 0;
                const int rc = z.run(P.out.data(), P.out.size(), 0, &got);
                if (rc == hfz::AT_BOUNDARY && got > 0 && text_like(P.out.data(), got)) { P.n = got; start = b; found = true; break; }
            }
            if (!found) { P.start_bit.store((size_t) -2, std::memory_order_release); P.rc = hfz::ERR_DATA; return; }
        }
        P.start_bit.store(start, std::memory_order_release);
        // up to the next piece's start (the last piece: the first boundary behind the round's nominal end)
        size_t stop = round_end_bit;
        bool open_end = false;
        if (k + 1 < R->pc.size()) {
            size_t nb;
            while ((nb = R->pc[k + 1]->start_bit.load(std::memory_order_acquire)) == (size_t) -1) {
                if (spec_abort_.load(std::memory_order_relaxed)) { P.rc = hfz::ERR_DATA; return; }
                std::this_thread::yield();
            }
            if (nb != (size_t) -2) stop = nb;
            else {
                // no block start behind this piece.  Near the end of the file that is the member's tail: run to its last block.  Anywhere else the
                // search has failed: stop at the first boundary behind the neighbour's nominal start — the chain breaks there and the one decoder goes on.
                const size_t nominal_next = (nominal_bit >> 3) + spec_piece_;
                if (map_len_ - std::min(map_len_, nominal_next) <= 2 * spec_piece_) open_end = true; else stop = nominal_next * 8;
            }
        }
        z.stop_bit = open_end ? (size_t) -1 : stop;
        for (;;) {
            size_t got = 0;
            const int rc = z.run(P.out.data() + P.n, P.out.size() - P.n, 0, &got);
            P.n += got;
            if (rc == hfz::OK) {                                    // output full
                if (P.out.size() > spec_piece_ * 64 + ((size_t) 64 << 20)) { P.rc = hfz::ERR_DATA; break; }   // (nothing inflates like that here)
                P.out.resize(P.out.size() + P.out.size() / 2);
                continue;
            }
            P.rc = rc;
            break;
        }
        P.end_bit = z.bit_pos();
    }
    void spec_run_round(SpecRound* R) {
        const size_t K = R->pc.size();
        const size_t start_byte = R->start_bit >> 3;
        const size_t round_end_bit = std::min(map_len_, start_byte + K * spec_piece_) * 8;
        std::vector<std::thread> th;
        for (size_t k = 1; k < K; k++) th.emplace_back([this, R, k, start_byte, round_end_bit] { spec_decode_piece(R, k, (start_byte + k * spec_piece_) * 8, round_end_bit); });
        spec_decode_piece(R, 0, R->start_bit, round_end_bit);
        for (auto& t : th) t.join();
        // the verified chain: piece k counts if it decoded cleanly and ended exactly where piece k + 1 began (or at the member's last block)
        size_t at = R->start_bit;
        for (size_t k = 0; k < K; k++) {
            SpecPiece& P = *R->pc[k];
            if (P.start_bit.load() != at) break;
            if (P.rc == hfz::END_OF_MEMBER) { R->good = k + 1; R->member_end = true; at = P.end_bit; break; }
            if (P.rc != hfz::AT_BOUNDARY) break;
            R->good = k + 1; at = P.end_bit;
        }
        R->end_bit = at;
        R->failed = R->good == 0 || (!R->member_end && R->good < K);   // (a partial chain is used as far as it goes, then the one decoder)
    }
    std::unique_ptr<SpecRound> spec_launch(size_t start_bit) {
        std::unique_ptr<SpecRound> R(new SpecRound());
        R->start_bit = start_bit;
        const size_t left = map_len_ - (start_bit >> 3);
        size_t K = (left + spec_piece_ - 1) / spec_piece_;
        if (K > spec_k_) K = spec_k_;
        if (K < 1) K = 1;
        for (size_t k = 0; k < K; k++) R->pc.emplace_back(new SpecPiece());
        SpecRound* r = R.get();
        R->th = std::thread([this, r] { spec_run_round(r); });
        return R;
    }
    void spec_begin(size_t bit, uint64_t produced) {
        spec_on_ = true; spec_cur_ = 0; spec_piece_i_ = 0; spec_off_in_piece_ = 0; spec_produced_ = produced;
        spec_round_[0] = spec_launch(bit);
        spec_round_[0]->th.join();
        spec_rounds_++;
        if (!spec_round_[0]->member_end && !spec_round_[0]->failed) spec_round_[1] = spec_launch(spec_round_[0]->end_bit);
    }
    // the one decoder again, at bit `bit` of the input (a verified block boundary), with `produced` bytes of the member before it
    void spec_leave(size_t bit, uint64_t produced, bool member_end) {
        spec_abort_.store(true);
        for (auto& r : spec_round_) if (r) { if (r->th.joinable()) r->th.join(); r.reset(); }
        spec_abort_.store(false);
        spec_on_ = false; spec_want_ = false;
        inf_.seek_bit(bit, produced);
        inf_.literal_only = false; inf_.stop_bit = (size_t) -1; inf_.stop_out = (uint64_t) -1;
        if (member_end) inf_.final_block = true;                    // (run() answers END_OF_MEMBER at once: the trailer follows)
        else spec_fallbacks_++;
    }
    size_t spec_fetch(uint8_t* dst, size_t want) {
        size_t got = 0;
        while (got < want && spec_on_) {
            SpecRound* R = spec_round_[spec_cur_].get();
            if (spec_piece_i_ < R->good) {
                SpecPiece& P = *R->pc[spec_piece_i_];
                const size_t n = std::min(want - got, P.n - spec_off_in_piece_);
                std::memcpy(dst + got, P.out.data() + spec_off_in_piece_, n);
                got += n; spec_off_in_piece_ += n; spec_produced_ += n;
                if (spec_off_in_piece_ == P.n) { std::vector<uint8_t>().swap(P.out); spec_piece_i_++; spec_off_in_piece_ = 0; }
                continue;
            }
            // the round is used up
            if (!map_is_static_) {                                  // its part of the mapped file is not needed again
                const size_t upto = ((R->end_bit >> 3) > ((size_t) 1 << 20) ? (R->end_bit >> 3) - ((size_t) 1 << 20) : 0) & ~(size_t) 4095;
                if (upto > released_) { madvise(const_cast<uint8_t*>(map_) + released_, upto - released_, MADV_DONTNEED); released_ = upto; }
            }
            if (R->member_end || R->failed) { spec_leave(R->end_bit, spec_produced_, R->member_end); break; }
            const int nxt = spec_cur_ ^ 1;
            SpecRound* N = spec_round_[nxt].get();
            if (N->th.joinable()) N->th.join();
            spec_rounds_++;
            spec_round_[spec_cur_].reset();
            spec_cur_ = nxt; spec_piece_i_ = 0; spec_off_in_piece_ = 0;
            if (N->good == 0) { spec_leave(N->start_bit, spec_produced_, false); break; }
            if (!N->member_end && !N->failed) spec_round_[nxt ^ 1] = spec_launch(N->end_bit);
        }
        return got;
    }
  public:
    size_t spec_rounds() const { return spec_rounds_; }
    size_t spec_fallbacks() const { return spec_fallbacks_; }
  private:
    bool spec_on_ = false, spec_want_ = false, spec_off_ = false;
    std::atomic<bool> spec_abort_{false};
    size_t spec_min_ = (size_t) 32 << 20, spec_piece_ = (size_t) 2 << 20, spec_k_ = 4;
    std::unique_ptr<SpecRound> spec_round_[2];
    int spec_cur_ = 0;
    size_t spec_piece_i_ = 0, spec_off_in_piece_ = 0, spec_rounds_ = 0, spec_fallbacks_ = 0;
    uint64_t spec_produced_ = 0;

    void produce() {
        if (map_) {
            inf_.in = map_; inf_.in_len = map_len_; inf_.pos = 0;
            gzip_ = map_len_ >= 2 && map_[0] == 0x1f && map_[1] == 0x8b;
            if (gzip_ && inf_.read_gzip_header() != hfz::OK) failed_ = true;
            if (gzip_) spec_member_start();
        }
        std::vector<char> carry;                                    // the end of the block before: >= the decoder's 32 KiB of history and
        size_t tail = 0;                                            // ... the unfinished line (its last `tail` bytes)
        bool eof = failed_;
        for (int i = 0;; i = (i + 1) % kN) {
            {
                std::unique_lock<std::mutex> g(m_);
                const auto w0 = std::chrono::steady_clock::now();
                cv_.wait(g, [&] { return stop_ || blk_[i].state == 0; });
                wait_prod_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
                if (stop_) return;
            }
            Block& b = blk_[i];
            const size_t c = carry.size();
            if (b.text.size() < c + kCap + 1) b.text.resize(c + kCap + 1);
            if (c) std::memcpy(b.text.data(), carry.data(), c);
            b.member_end = false; b.data0 = c;
            size_t have = c, end_of_lines = 0;
            bool cut = false;
            while (!eof && !cut) {                                  // until the block holds a line end (or the file / a member ends)
                const size_t want = b.text.size() - 1 - have;
                have += fetch(b, have, have, want, eof);
                if (b.member_end || eof) break;
                size_t k = have;
                while (k > c - tail && b.text[k - 1] != '\n') k--;
                if (k > c - tail) { end_of_lines = k; cut = true; break; }
                if (b.text.size() - 1 - have < 4096) b.text.resize(b.text.size() * 2);   // one line longer than the block: keep reading
            }
            size_t new_tail = 0;
            if (!cut) {                                             // the end of a member or of the file
                if (eof) end_of_lines = have;                       // (a last line without '\n' is a line)
                else {
                    size_t k = have;
                    while (k > c - tail && b.text[k - 1] != '\n') k--;
                    end_of_lines = k;
                }
            }
            new_tail = have - end_of_lines;
            b.off = c - tail; b.len = end_of_lines - b.off; b.data_len = have - c;
            const size_t keep = std::min(have, std::max(kHist, new_tail));
            carry.assign(b.text.data() + have - keep, b.text.data() + have);
            tail = new_tail;
            const bool nothing = failed_ || (b.data_len == 0 && b.len == 0 && eof && !b.member_end);
            {
                std::lock_guard<std::mutex> g(m_);
                b.last = nothing; b.recs.clear();
                b.state = nothing ? 3 : 1;
            }
            cv_.notify_all();
            if (nothing) return;
        }
    }
    void parse_loop() {
        for (;;) {
            int i;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || blk_[parse_].state == 1 || (blk_[parse_].state == 3 && blk_[parse_].last); });
                if (stop_ || blk_[parse_].last) { cv_.notify_all(); return; }
                i = parse_; blk_[i].state = 2; parse_ = (parse_ + 1) % kN;
            }
            Block& b = blk_[i];
            b.crc = gzip_ ? (uint32_t) crc32(0L, reinterpret_cast<const Bytef*>(b.text.data() + b.data0), (uInt) b.data_len) : 0u;
            parse_block(b.base(), b.len, b.recs);
            { std::lock_guard<std::mutex> g(m_); b.state = 3; }
            cv_.notify_all();
        }
    }
    gzFile gz_ = nullptr;
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t map_len_ = 0;
    bool map_is_static_ = false, gzip_ = false, open_failed_ = false;
    hfz::Inflater inf_;
    Block blk_[kN];
    std::thread prod_;
    std::vector<std::thread> work_;
    std::mutex m_;
    std::condition_variable cv_;
    int parse_ = 0, cons_ = 0, held_ = -1;
    uint32_t run_crc_ = 0;
    uint64_t run_len_ = 0;
    bool stop_ = false;
    std::atomic<bool> failed_{false};
};

// atoi / atof of a field for the shapes that occur in coverage files, with the library calls as the fallback.
// An integer of at most 15 digits divided by an exact power of ten is correctly rounded, i.e. it is strtod's result.
inline int fast_atoi(const char* p) {
    const char* q = p;
    bool neg = false;
    if (*q == '-') { neg = true; q++; } else if (*q == '+') q++;
    if (*q < '0' || *q > '9') return std::atoi(p);
    long v = 0;
    int nd = 0;
    while (*q >= '0' && *q <= '9' && nd < 10) { v = v * 10 + (*q - '0'); q++; nd++; }
    if (*q >= '0' && *q <= '9') return std::atoi(p);     // very long: let the library decide
    if (v > 2147483647L) return std::atoi(p);
    return (int) (neg ? -v : v);
}
inline double fast_atof(const char* p) {
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const char* q = p;
    bool neg = false;
    if (*q == '-') { neg = true; q++; } else if (*q == '+') q++;
    if (*q < '0' || *q > '9') return std::atof(p);
    unsigned long long m = 0;
    int nd = 0, frac = 0;
    while (*q >= '0' && *q <= '9') { if (++nd > 15) return std::atof(p); m = m * 10 + (unsigned) (*q - '0'); q++; }
    if (*q == '.') {
        q++;
        while (*q >= '0' && *q <= '9') { if (++nd > 15) return std::atof(p); m = m * 10 + (unsigned) (*q - '0'); q++; frac++; }
    }
    if (*q != '\0') return std::atof(p);                 // exponent, inf, nan, trailing text: the library's rules
    const double v = frac ? (double) m / p10[frac] : (double) m;
    return neg ? -v : v;
}

// a block's lines -> records.  The text is cut in place (fields and lines become NUL-terminated strings).
void parse_block(char* text, size_t len, std::vector<CovRec>& out) {
    out.clear();
    out.reserve(len / 24 + 16);
    char* p = text;
    char* const end = text + len;
    while (p < end) {
        char* nl = (char*) std::memchr(p, '\n', (size_t) (end - p));
        char* le = nl ? nl : end;                                   // (the file's last line may lack its '\n': text has a spare byte)
        char* next = nl ? nl + 1 : end;
        *le = '\0';
        size_t L = (size_t) (le - p);
        if (L && p[L - 1] == '\r') p[--L] = '\0';
        if (L == 0) { p = next; continue; }
        CovRec r;
        std::memset(&r, 0, sizeof r);
        if (p[0] == '#' || p[0] == '>') {
            r.kind = p[0] == '#' ? CovRec::HEADER : CovRec::CONTIG;
            r.s = (int32_t) (p - text); r.e = (int32_t) L;
            out.push_back(r);
            p = next;
            continue;
        }
        // start end cov mapq clip annots region [truth [prediction]]
        // Fast path (round 6): the shape every row of a bam2cov track has — non-negative integers in every column, annotation indices separated by
        // commas — read in ONE scan of the line, digits accumulated as they pass.  Anything else (a sign, a decimal point, an exponent, a field of
        // ten or more digits, a missing column) leaves `ok` false and the row takes the general path below, which accepts what strtod / atoi accept.
        {
            const char* q = p;
            uint32_t v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            uint64_t flag = 0;
            int f = 0;
            bool ok = true;
            while (ok && f < 9) {
                if (f == 5) {                                       // annotation indices: i[,j...]   (ptBlock.c:225-236: 1-based bits, 0 = none)
                    for (;;) {
                        uint32_t a = 0; int nd = 0;
                        while ((unsigned) (*q - '0') <= 9u && nd < 4) { a = a * 10 + (uint32_t) (*q - '0'); q++; nd++; }
                        if (nd == 0 || nd >= 4) { ok = false; break; }
                        if (a > 0 && a <= 64) flag |= 1ULL << (a - 1);
                        if (*q == ',') { q++; continue; }
                        break;
                    }
                } else {
                    uint32_t a = 0; int nd = 0;
                    while ((unsigned) (*q - '0') <= 9u && nd < 10) { a = a * 10 + (uint32_t) (*q - '0'); q++; nd++; }
                    if (nd == 0 || nd >= 10) ok = false;            // (at most nine digits: no overflow, and the double of it is exact)
                    v[f] = a;
                }
                if (!ok) break;
                f++;
                if (*q == '\t') { q++; continue; }
                if (q == le) break;                                 // (the line's end: *le is the terminator written above)
                ok = false;
            }
            if (ok && q == le && f >= 7) {
                r.kind = CovRec::ROW;
                r.s = (int32_t) v[0] - 1; r.e = (int32_t) v[1] - 1;
                r.cov = (double) v[2]; r.mapq = (double) v[3]; r.clip = (double) v[4];
                r.flag = flag;
                r.region = (int16_t) clampi((int) v[6], 0, 100);
                r.truth = (int16_t) (clampi((f >= 8 ? (int) v[7] : -1), -1, 10) + 1);
                r.pred = (int16_t) (clampi((f >= 9 ? (int) v[8] : -1), -1, 10) + 1);
                out.push_back(r);
                p = next;
                continue;
            }
        }
        char* fld[10]; int nf = 0;
        for (char* q = p; nf < 10;) {
            fld[nf++] = q;
            char* t = (char*) std::memchr(q, '\t', (size_t) (le - q));
            if (!t) break;
            *t = '\0'; q = t + 1;
        }
        if (nf < 7) { r.kind = CovRec::SHORT_ROW; out.push_back(r); p = next; continue; }
        r.kind = CovRec::ROW;
        r.s = fast_atoi(fld[0]) - 1; r.e = fast_atoi(fld[1]) - 1;   // 1-based inclusive -> 0-based
        r.cov = fast_atof(fld[2]); r.mapq = fast_atof(fld[3]); r.clip = fast_atof(fld[4]);
        r.flag = annot_flag_of(fld[5]);
        r.region = (int16_t) clampi(fast_atoi(fld[6]), 0, 100);
        r.truth = (int16_t) (clampi((nf >= 8 ? fast_atoi(fld[7]) : -1), -1, 10) + 1);
        r.pred = (int16_t) (clampi((nf >= 9 ? fast_atoi(fld[8]) : -1), -1, 10) + 1);
        out.push_back(r);
        p = next;
    }
}

// ---- .cov / .cov.gz: header (track_reader.c:48-457), rows (:751-818), chunks (chunk.c:240-294), windows ----
hfio_table* load_cov(const char* path, int chunk_len, int window_len) {
    if (chunk_len <= 0 || window_len <= 0) { g_io_err = "chunkLen/windowLen must be > 0"; return nullptr; }
    BlockReader* reader = new BlockReader(path);                    // .cov.gz (gzip members) and uncompressed text
    if (reader->open_failed()) { delete reader; g_io_err = std::string("[Error] Unable to open ") + path; return nullptr; }
    hfio_table* t = new hfio_table();
    t->chunk_len = chunk_len; t->window_len = window_len;
    bool have_ann = false, have_reg = false, have_lab = false, have_avg = false;
    int n_ann = 0, n_reg = 0, parsed_cov = 0;
    std::string ctg;
    int ctg_len = 0;
    bool in_contig = false;
    ChunkMeta cur{};
    WindowAcc acc;
    int next_pos = 0;                                               // next base expected in the current contig
    int open_ws = 0, open_we = -1, open_cs = -1;                    // the open window: first / last base, and the chunk start it was computed for
    auto fail = [&](const std::string& m) { g_io_err = m; delete reader; delete t; return (hfio_table*) nullptr; };
    auto first_chunk = [&]() {
        cur.ctg = ctg; cur.ctg_len = ctg_len; cur.s = 0;
        cur.e = ctg_len < 2 * chunk_len ? ctg_len - 1 : chunk_len - 1;   // chunk.c:262
    };
    auto next_chunk = [&]() {
        const int pe = cur.e;
        cur.s = pe + 1;
        cur.e = ctg_len < pe + 2 * chunk_len ? ctg_len - 1 : pe + chunk_len;   // chunk.c:274-277
    };
    while (BlockReader::Block* blk = reader->next())
    for (const CovRec& r : blk->recs) {
        if (r.kind == CovRec::HEADER) {
            const char* line = blk->base() + r.s;
            if (starts_with(line, "#annotation:len") && !have_ann) {
                const char* p = field_after(line, 2); n_ann = p ? std::atoi(p) : 0; have_ann = true;
                t->annotation_names.assign((size_t) (n_ann > 0 ? n_ann : 0), "NA");
            } else if (starts_with(line, "#annotation:name:")) {
                const char* p = field_after(line, 2); const char* q = field_after(line, 3);
                const int idx = p ? std::atoi(p) : -1;
                if (q && idx >= 0 && idx < (int) t->annotation_names.size()) {
                    std::string nm(q); const size_t c = nm.find(':'); if (c != std::string::npos) nm.resize(c);
                    t->annotation_names[(size_t) idx] = nm;
                }
            } else if (starts_with(line, "#region:len") && !have_reg) {
                const char* p = field_after(line, 2); n_reg = p ? std::atoi(p) : 0; have_reg = true;
                if (n_reg > 0 && n_reg <= HF_MAXREGIONS) t->region_coverages.assign((size_t) n_reg, 0);
            } else if (starts_with(line, "#region:coverage:")) {
                const char* p = field_after(line, 2); const char* q = field_after(line, 3);
                const int idx = p ? std::atoi(p) : -1;
                if (q && idx >= 0 && idx < (int) t->region_coverages.size()) t->region_coverages[(size_t) idx] = std::atoi(q);
                parsed_cov++;
            } else if (starts_with(line, "#label:len") && !have_lab) {
                const char* p = field_after(line, 2); t->n_labels = p ? std::atoi(p) : 0; have_lab = true;
            } else if (starts_with(line, "#truth:true")) t->truth_available = true;
            else if (starts_with(line, "#prediction:true")) t->prediction_available = true;
            else if (starts_with(line, "#start-only:true")) t->start_only = true;
            else if (starts_with(line, "#avg_alignment_len:") && !have_avg) { t->avg_alignment_len = std::atoi(line + 19); have_avg = true; }
            continue;
        }
        if (!have_ann) return fail("Error: No '#annotation:len:' found in the header. annotation len should be at least 1.");
        if (n_ann <= 0) return fail("Error: The value of '#annotation:len:' in the header should be at least 1.");
        if (!have_reg) return fail("Error: No '#region:len:' found in the header. region len should be at least 1.");
        if (n_reg <= 0 || n_reg > HF_MAXREGIONS) return fail("Error: The value of '#region:len:' in the header should be at least 1 (and at most 64).");
        if (r.kind == CovRec::CONTIG) {
            char* line = blk->base() + r.s;
            if (in_contig) { t->push_window(acc); t->close_chunk(cur); }
            char* sp = std::strchr(line, ' ');
            ctg_len = sp ? std::atoi(sp + 1) : 0;
            if (sp) *sp = '\0';
            ctg = line + 1;
            in_contig = true; next_pos = 0; acc.reset(); open_we = -1; open_cs = -1;
            first_chunk();
            continue;
        }
        if (!in_contig) return fail("Error: coverage row before any '>contig length' line");
        if (r.kind == CovRec::SHORT_ROW) return fail("Error: a coverage row has fewer than 7 columns");
        const int s = r.s, e = r.e;
        const double v_cov = r.cov, v_mapq = r.mapq, v_clip = r.clip;
        const uint64_t flag = r.flag;
        const int region = r.region, truth = r.truth, pred = r.pred;
        if (s != next_pos || e < s) return fail("Error: coverage rows must tile each contig without gaps (chunk.c:451)");
        int pos = s;
        while (pos <= e && pos <= ctg_len - 1) {
            // the open window's last base: kept from row to row (a row of a real track is a few hundred bases, a window thousands: the division
            // by the window length is needed once per window, not once per row)
            if (pos < open_ws || pos > open_we || open_cs != cur.s) {
                const int wi = (pos - cur.s) / window_len;
                open_ws = cur.s + wi * window_len; open_we = open_ws + window_len - 1; open_cs = cur.s;
                if (open_we > cur.e) open_we = cur.e;
            }
            const int wend = open_we;
            const int seg_end = e < wend ? e : wend;
            const int n = seg_end - pos + 1;
            add_run(acc.cov, v_cov, n); add_run(acc.mapq, v_mapq, n); add_run(acc.clip, v_clip, n);
            acc.flag |= flag; acc.reg.add(region, n); acc.tru.add(truth, n); acc.pre.add(pred, n); acc.n += n;
            if (seg_end == wend) t->push_window(acc);              // full window, or the chunk's trailing partial window
            if (seg_end == cur.e) { t->close_chunk(cur); if (cur.e < ctg_len - 1) next_chunk(); }
            pos = seg_end + 1;
        }
        next_pos = e + 1;
    }
    if (std::getenv("HF_IO_TRACE"))
        std::fprintf(stderr, "[hfio] the consumer waited %.3f s for parsed blocks, the inflating thread %.3f s for free blocks; %zu rounds of parallel decoders, %zu fall-backs to one\n",
                     reader->consumer_wait(), reader->producer_wait(), reader->spec_rounds(), reader->spec_fallbacks());
    if (reader->failed()) return fail(std::string("Error: ") + path + " is truncated or corrupt (the deflate stream ended on an error, or a gzip member's CRC-32 / length does not match)");
    delete reader;
    if (in_contig) { t->push_window(acc); t->close_chunk(cur); }
    if (!have_ann || !have_reg) { g_io_err = "Error: missing '#annotation:len:' / '#region:len:' header"; delete t; return nullptr; }
    if (parsed_cov != n_reg) { g_io_err = "Error: Number of parsed region coverages does not match '#region:len:' in the header line."; delete t; return nullptr; }
    if ((t->truth_available || t->prediction_available) && !have_lab) {
        g_io_err = "Error: '#label:len' should be set to a non-zero number if at least one of truth or prediction tags is set to true in the header.";
        delete t; return nullptr;
    }
    if (t->start_only && (!have_avg || t->avg_alignment_len <= 0)) { g_io_err = "Error: '#avg_alignment_len:' > 0 is required for start-only mode"; delete t; return nullptr; }
    return t;
}

// ---- .bin (chunk.c:596-709 write, 713-828 read): little-endian, no magic ----
bool rd(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }

hfio_table* load_bin(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { g_io_err = std::string("Error: The bin file ") + path + " does not exist."; return nullptr; }
    hfio_table* t = new hfio_table();
    auto fail = [&](const char* m) { g_io_err = m; std::fclose(f); delete t; return (hfio_table*) nullptr; };
    int32_t n_ann = 0, n_reg = 0;
    if (!rd(f, &n_ann, 4) || n_ann < 0 || n_ann > 4096) return fail("bad .bin header");
    for (int i = 0; i < n_ann; i++) {
        int32_t len = 0;
        if (!rd(f, &len, 4) || len <= 0 || len > 65536) return fail("bad .bin header");
        std::string s((size_t) len, '\0');
        if (!rd(f, &s[0], (size_t) len)) return fail("bad .bin header");
        s.resize(std::strlen(s.c_str()));
        t->annotation_names.push_back(s);
    }
    if (!rd(f, &n_reg, 4) || n_reg < 0 || n_reg > HF_MAXREGIONS) return fail("bad .bin header");
    t->region_coverages.resize((size_t) n_reg);
    if (n_reg && !rd(f, t->region_coverages.data(), 4 * (size_t) n_reg)) return fail("bad .bin header");
    uint8_t b3[3];
    if (!rd(f, &t->n_labels, 4) || !rd(f, b3, 3) || !rd(f, &t->avg_alignment_len, 4) || !rd(f, &t->chunk_len, 4) ||
        !rd(f, &t->window_len, 4)) return fail("bad .bin header");
    t->truth_available = b3[0]; t->prediction_available = b3[1]; t->start_only = b3[2];
    int32_t name_len;
    while (std::fread(&name_len, 4, 1, f) == 1) {
        if (name_len <= 0 || name_len > 65536) return fail("bad .bin chunk");
        std::string nm((size_t) name_len, '\0');
        ChunkMeta cm{};
        int32_t n = 0;
        if (!rd(f, &nm[0], (size_t) name_len) || !rd(f, &cm.ctg_len, 4) || !rd(f, &cm.s, 4) || !rd(f, &cm.e, 4) || !rd(f, &n, 4) || n < 0)
            return fail("bad .bin chunk");
        nm.resize(std::strlen(nm.c_str()));
        cm.ctg = nm;
        const size_t o = t->cov.size(), N = o + (size_t) n;
        t->cov.resize(N); t->mapq.resize(N); t->clip.resize(N); t->annot.resize(N); t->truth.resize(N); t->prediction.resize(N);
        if (n && (!rd(f, &t->cov[o], 2 * (size_t) n) || !rd(f, &t->mapq[o], 2 * (size_t) n) || !rd(f, &t->clip[o], 2 * (size_t) n) ||
                  !rd(f, &t->annot[o], 8 * (size_t) n) || !rd(f, &t->truth[o], (size_t) n) || !rd(f, &t->prediction[o], (size_t) n)))
            return fail("truncated .bin chunk");
        t->chunks.push_back(cm);
        t->chunk_s.push_back(cm.s); t->chunk_e.push_back(cm.e); t->chunk_ctg_len.push_back(cm.ctg_len);
        t->chunk_off.push_back((int64_t) N);
    }
    std::fclose(f);
    return t;
}

const char* const kLabelColors[] = {"162,0,37", "250,104,0", "0,138,0", "170,0,255", "99, 99, 96", "250,200,0"};  // chunk.c:10-15
const char* const kLabelNames[] = {"Err", "Dup", "Hap", "Col", "Unk", "Msj"};                                        // chunk.c:16-21
const char* const kStateNames[] = {"Err", "Dup", "Hap", "Col"};

struct Run { int s, e, label; };

void emit_contig(FILE* out, const std::string& ctg, const std::vector<Run>& runs) {   // chunk.c:953-983 + 1058-1072
    if (runs.empty()) return;
    int start = 0, end = 0, label = -1;      // the merged block of a contig starts at 0 (preStart = 0 in the reference)
    auto print = [&]() {
        std::fprintf(out, "%s\t%d\t%d\t%s\t0\t.\t%d\t%d\t%s\n", ctg.c_str(), start, end + 1, kLabelNames[label], start, end + 1,
                     kLabelColors[label]);
    };
    for (const Run& r : runs) {
        if (label != -1 && r.label != label) { print(); start = r.s; }
        end = r.e; label = r.label;
    }
    print();
}
}  // namespace

extern "C" {

const char* hfio_last_error(void) { return g_io_err.c_str(); }

hfio_table* hfio_load(const char* path, int chunk_len, int window_len) {
    if (!path) { g_io_err = "Error: Input path cannot be NULL."; return nullptr; }
    const std::string ext = file_ext(path);
    if (ext == "bin") return load_bin(path);
    if (ext == "cov" || ext == "cov.gz") return load_cov(path, chunk_len, window_len);
    g_io_err = "Error: input file should either cov/cov.gz or a binary file made with create_bin_chunks.";
    return nullptr;
}

void hfio_destroy(hfio_table* t) { delete t; }

int hfio_gunzip(const char* path, unsigned char** out, size_t* n) {
    if (!path || !out || !n) return -5;
    *out = nullptr; *n = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return -5;
    std::vector<uint8_t> in;
    {
        uint8_t buf[1 << 16];
        size_t r;
        while ((r = std::fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + r);
        std::fclose(f);
    }
    static thread_local hfz::Inflater z;
    z.in = in.data(); z.in_len = in.size(); z.pos = 0;
    int rc = z.read_gzip_header();
    if (rc != hfz::OK) return rc == hfz::ERR_TRUNCATED ? -2 : -3;
    size_t cap = 1 << 20, len = 0, member0 = 0;
    uint8_t* o = (uint8_t*) std::malloc(cap);
    if (!o) return -5;
    for (;;) {
        if (cap - len < (1 << 16)) { cap *= 2; uint8_t* p2 = (uint8_t*) std::realloc(o, cap); if (!p2) { std::free(o); return -5; } o = p2; }
        size_t got = 0;
        rc = z.run(o + len, cap - len, len - member0, &got);
        len += got;
        if (rc == hfz::OK) continue;
        if (rc != hfz::END_OF_MEMBER) { std::free(o); return rc == hfz::ERR_TRUNCATED ? -2 : -1; }
        uint32_t crc = 0, isz = 0;
        if (z.read_gzip_trailer(&crc, &isz) != hfz::OK) { std::free(o); return -2; }
        if ((uint32_t) crc32(0L, o + member0, (uInt) (len - member0)) != crc || (uint32_t) (len - member0) != isz) { std::free(o); return -4; }
        member0 = len;
        const int nm = z.next_member();                                      // the next member, or the end (trailing bytes that are no gzip member are ignored)
        if (nm == hfz::ERR_TRUNCATED) { std::free(o); return -2; }           // cut inside the next member's header
        if (nm != hfz::OK) break;
    }
    *out = o; *n = len;
    return 0;
}
void hfio_free(void* p) { std::free(p); }
int64_t hfio_n_windows(const hfio_table* t) { return (int64_t) t->cov.size(); }
int32_t hfio_n_chunks(const hfio_table* t) { return (int32_t) t->chunks.size(); }
int32_t hfio_n_regions(const hfio_table* t) { return (int32_t) t->region_coverages.size(); }
const int32_t* hfio_region_coverages(const hfio_table* t) { return t->region_coverages.data(); }
int32_t hfio_window_len(const hfio_table* t) { return t->window_len; }
int32_t hfio_chunk_len(const hfio_table* t) { return t->chunk_len; }
int32_t hfio_avg_alignment_len(const hfio_table* t) { return t->avg_alignment_len; }
int32_t hfio_start_only(const hfio_table* t) { return t->start_only ? 1 : 0; }
int32_t hfio_n_annotations(const hfio_table* t) { return (int32_t) t->annotation_names.size(); }
const char* hfio_annotation_name(const hfio_table* t, int i) { return t->annotation_names[(size_t) i].c_str(); }
const char* hfio_chunk_ctg(const hfio_table* t, int c) { return t->chunks[(size_t) c].ctg.c_str(); }
int32_t hfio_subset_contigs(hfio_table* t, const char* const* names, int n_names) {
    std::vector<ChunkMeta> chunks;
    std::vector<int64_t> off{0};
    std::vector<int32_t> cs, ce, cl;
    std::vector<uint16_t> cov, mapq, clip;
    std::vector<uint64_t> annot;
    std::vector<int8_t> truth, pred;
    for (size_t c = 0; c < t->chunks.size(); c++) {
        bool keep = false;
        for (int k = 0; k < n_names && !keep; k++) keep = t->chunks[c].ctg == names[k];
        if (!keep) continue;
        const size_t a = (size_t) t->chunk_off[c], b = (size_t) t->chunk_off[c + 1];
        chunks.push_back(t->chunks[c]);
        cs.push_back(t->chunk_s[c]); ce.push_back(t->chunk_e[c]); cl.push_back(t->chunk_ctg_len[c]);
        cov.insert(cov.end(), t->cov.begin() + a, t->cov.begin() + b);
        mapq.insert(mapq.end(), t->mapq.begin() + a, t->mapq.begin() + b);
        clip.insert(clip.end(), t->clip.begin() + a, t->clip.begin() + b);
        annot.insert(annot.end(), t->annot.begin() + a, t->annot.begin() + b);
        if (!t->truth.empty()) truth.insert(truth.end(), t->truth.begin() + a, t->truth.begin() + b);
        if (!t->prediction.empty()) pred.insert(pred.end(), t->prediction.begin() + a, t->prediction.begin() + b);
        off.push_back((int64_t) cov.size());
    }
    t->chunks.swap(chunks); t->chunk_off.swap(off); t->chunk_s.swap(cs); t->chunk_e.swap(ce); t->chunk_ctg_len.swap(cl);
    t->cov.swap(cov); t->mapq.swap(mapq); t->clip.swap(clip); t->annot.swap(annot); t->truth.swap(truth); t->prediction.swap(pred);
    return (int32_t) t->chunks.size();
}

char** hfio_read_name_list(const char* path, int* n_names) {
    *n_names = 0;
    FILE* f = std::fopen(path, "r");
    if (!f) { g_io_err = std::string("Error: Unable to open ") + path; return nullptr; }
    std::vector<std::string> names;
    char line[4096];
    while (std::fgets(line, sizeof line, f)) {
        size_t l = std::strlen(line);
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = '\0';
        if (char* sp = std::strchr(line, ' ')) *sp = '\0';
        if (line[0]) names.push_back(line);
    }
    std::fclose(f);
    char** out = (char**) std::calloc(names.size() + 1, sizeof(char*));
    for (size_t i = 0; i < names.size(); i++) out[i] = strdup(names[i].c_str());
    *n_names = (int) names.size();
    return out;
}

int8_t* hfio_truth(hfio_table* t) { return t->truth.data(); }
int32_t hfio_truth_available(const hfio_table* t) { return t->truth_available ? 1 : 0; }
int32_t hfio_n_labels(const hfio_table* t) { return t->n_labels; }

int hfio_write_summary(hfio_table* t, const int8_t* labels, const char* output_path, const char* bin_array_path,
                       const char* const* label_names_with_unknown, int n_label_names, double overlap_ratio_threshold,
                       int threads) {
    {   // hmm_flagger.c:353-354.  (Several table sets may be written at once, each on a thread of its own: one writer at a time; whoever
        // reads the two fields afterwards has joined those threads.)
        static std::mutex mark_m;
        std::lock_guard<std::mutex> g(mark_m);
        t->prediction_available = true;
        t->n_labels = 4;
    }
    std::vector<const char*> ctg(t->chunks.size()), ann(t->annotation_names.size());
    for (size_t c = 0; c < ctg.size(); c++) ctg[c] = t->chunks[c].ctg.c_str();
    for (size_t a = 0; a < ann.size(); a++) ann[a] = t->annotation_names[a].c_str();
    hfs_input in;
    in.n_windows = (int64_t) t->cov.size(); in.n_chunks = (int32_t) t->chunks.size();
    in.chunk_off = t->chunk_off.data(); in.chunk_s = t->chunk_s.data(); in.chunk_e = t->chunk_e.data();
    in.chunk_ctg = ctg.data(); in.window_len = t->window_len;
    in.annot = t->annot.data(); in.truth = t->truth.empty() ? nullptr : t->truth.data(); in.prediction = labels;
    in.truth_available = t->truth_available; in.prediction_available = 1; in.n_labels = 4;
    in.n_regions = (int32_t) t->region_coverages.size(); in.n_annotations = (int32_t) ann.size();
    in.annotation_names = ann.data();
    const int rc = hfs_write_all_tables(&in, output_path, bin_array_path, label_names_with_unknown, n_label_names,
                                        overlap_ratio_threshold, threads);
    if (rc) g_io_err = hfs_last_error();
    return rc;
}
int8_t* hfio_prediction(hfio_table* t) { return t->prediction.data(); }

void hfio_windows(const hfio_table* t, hf_windows* w) {
    w->n_windows = (int64_t) t->cov.size(); w->n_chunks = (int32_t) t->chunks.size();
    w->chunk_off = t->chunk_off.data(); w->cov = t->cov.data(); w->mapq = t->mapq.data(); w->clip = t->clip.data();
    w->annot = t->annot.data(); w->chunk_s = t->chunk_s.data(); w->chunk_e = t->chunk_e.data();
    w->chunk_ctg_len = t->chunk_ctg_len.data();
    w->window_len = t->window_len; w->mean_read_len = t->avg_alignment_len;
}

int hfio_write_bin(const hfio_table* t, const char* path) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return -1;
    const int32_t n_ann = (int32_t) t->annotation_names.size(), n_reg = (int32_t) t->region_coverages.size();
    std::fwrite(&n_ann, 4, 1, f);
    for (const std::string& s : t->annotation_names) { const int32_t len = (int32_t) s.size() + 1; std::fwrite(&len, 4, 1, f); std::fwrite(s.c_str(), 1, (size_t) len, f); }
    std::fwrite(&n_reg, 4, 1, f);
    std::fwrite(t->region_coverages.data(), 4, (size_t) n_reg, f);
    std::fwrite(&t->n_labels, 4, 1, f);
    const uint8_t b3[3] = {(uint8_t) t->truth_available, (uint8_t) t->prediction_available, (uint8_t) t->start_only};
    std::fwrite(b3, 1, 3, f);
    std::fwrite(&t->avg_alignment_len, 4, 1, f); std::fwrite(&t->chunk_len, 4, 1, f); std::fwrite(&t->window_len, 4, 1, f);
    for (size_t c = 0; c < t->chunks.size(); c++) {
        const ChunkMeta& cm = t->chunks[c];
        const size_t o = (size_t) t->chunk_off[c]; const int32_t n = (int32_t) (t->chunk_off[c + 1] - t->chunk_off[c]);
        const int32_t nl = (int32_t) cm.ctg.size() + 1;
        std::fwrite(&nl, 4, 1, f); std::fwrite(cm.ctg.c_str(), 1, (size_t) nl, f);
        std::fwrite(&cm.ctg_len, 4, 1, f); std::fwrite(&cm.s, 4, 1, f); std::fwrite(&cm.e, 4, 1, f); std::fwrite(&n, 4, 1, f);
        std::fwrite(&t->cov[o], 2, (size_t) n, f); std::fwrite(&t->mapq[o], 2, (size_t) n, f); std::fwrite(&t->clip[o], 2, (size_t) n, f);
        std::fwrite(&t->annot[o], 8, (size_t) n, f); std::fwrite(&t->truth[o], 1, (size_t) n, f); std::fwrite(&t->prediction[o], 1, (size_t) n, f);
    }
    return std::fclose(f) == 0 ? 0 : -1;
}

// chunk.c:985-1124: per contig, maximal runs of equal labels; a run shorter than its state's minimum length
// becomes Hap; equal neighbours are then merged; rows are 0-based half-open
int hfio_write_final_bed(const hfio_table* t, const int8_t* labels, const char* path, const char* track_name,
                         const int32_t* min_len_per_state) {
    FILE* out = std::fopen(path, "w");
    if (!out) return -1;
    std::fprintf(out, "track name=%s visibility=1 itemRgb=\"On\"\n", track_name);
    const int hap = 2;
    std::vector<Run> runs;
    const std::string* pre_ctg_p = nullptr;   // (the contig of the window before: compared once per chunk, not per window)
    int run_start = 0, pre_end = 0, pre_label = -1;
    bool have = false;
    auto close_run = [&]() {
        const int len = pre_end + 1 - run_start;
        const int minlen = (pre_label >= 0 && pre_label < 4 && min_len_per_state) ? min_len_per_state[pre_label] : 0;
        runs.push_back(Run{run_start, pre_end, len < minlen ? hap : pre_label});
    };
    for (size_t c = 0; c < t->chunks.size(); c++) {
        const ChunkMeta& cm = t->chunks[c];
        const int64_t a = t->chunk_off[c], b = t->chunk_off[c + 1];
        const bool new_ctg = have && *pre_ctg_p != cm.ctg;                    // only a chunk's first window can start a contig
        for (int64_t i = a; i < b; i++) {
            const int k = (int) (i - a);
            const int start = cm.s + k * t->window_len;                       // chunk.c:934-935
            int end = cm.s + (k + 1) * t->window_len - 1;
            if (cm.e < end) end = cm.e;
            const int label = labels[i] != -1 ? labels[i] : 4;                // 4 = "Unk"
            if (!have) run_start = start;
            const bool label_changed = have && label != pre_label;
            const bool ctg_changed = new_ctg && i == a;
            if (label_changed || ctg_changed) { close_run(); run_start = start; }
            if (ctg_changed) { emit_contig(out, *pre_ctg_p, runs); runs.clear(); }
            pre_end = end; pre_label = label; have = true;
            if (i == a) pre_ctg_p = &cm.ctg;
        }
    }
    if (have) { close_run(); emit_contig(out, *pre_ctg_p, runs); }
    return std::fclose(out) == 0 ? 0 : -1;
}

int hfio_write_posterior_bed(const hfio_table* t, const double* posterior, const int8_t* labels, const char* path) {  // hmm_flagger.c:240-282
    FILE* out = std::fopen(path, "w");
    if (!out) return -1;
    std::fprintf(out, "#ctg\tstart\tend\t");
    for (int s = 0; s < 4; s++) std::fprintf(out, "posterior_%s_%d\t", kStateNames[s], s);
    std::fprintf(out, "prediction\n");
    for (size_t c = 0; c < t->chunks.size(); c++) {
        const ChunkMeta& cm = t->chunks[c];
        const int64_t a = t->chunk_off[c], b = t->chunk_off[c + 1];
        for (int64_t i = a; i < b; i++) {
            const int k = (int) (i - a);
            const int start = cm.s + k * t->window_len;
            int end = cm.s + (k + 1) * t->window_len - 1;
            if (cm.e < end) end = cm.e;
            std::fprintf(out, "%s\t%d\t%d\t", cm.ctg.c_str(), start, end + 1);
            for (int s = 0; s < 4; s++) std::fprintf(out, "%.2f\t", posterior[i * 4 + s]);
            std::fprintf(out, "%s\n", kStateNames[labels[i] >= 0 && labels[i] < 4 ? labels[i] : 0]);
        }
    }
    return std::fclose(out) == 0 ? 0 : -1;
}

}  // extern "C"
