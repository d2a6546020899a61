// hf_inflate.h — a DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for the loader (hf_io.cpp): reading a .cov.gz was bound by zlib's
// inflate on one thread (~350-400 MB/s of text) once the text -> numbers work had moved to other threads.  Written from the RFCs:
// a 64-bit bit buffer refilled with one unaligned load, one table look-up per symbol (11-bit primary table + second-level tables for
// longer codes, 8-bit primary table for distances), matches copied eight bytes at a time.  The decoder is resumable between symbols:
// the caller asks for "up to n bytes" and gets a little less when the next symbol might not fit.
//
// Safety net: the caller checks CRC-32 and ISIZE of every gzip member (hf_io.cpp; zlib's crc32 / crc32_combine); any structural error
// here returns an error code — there is no path on which wrong bytes are accepted silently.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace hfz {

enum { OK = 0, END_OF_MEMBER = 1, NEED_ROOM = 2, AT_BOUNDARY = 3, ERR_DATA = -1, ERR_TRUNCATED = -2, ERR_HEADER = -3, ERR_MATCH = -4 };

constexpr int kLitBits = 11, kDistBits = 8;
// table entry: bits 0-7 code length to consume (primary: whole code or, for a sub-table pointer, the primary bits); bits 8-15 kind:
//   0x80 literal, 0x40 end of block, 0x20 sub-table pointer (bits 16-31 = first entry, bits 8-12 = sub-table bits), else a length /
//   distance symbol with bits 8-12 = number of extra bits; bits 16-31: literal value or base
constexpr uint32_t F_LIT = 0x8000u, F_EOB = 0x4000u, F_SUB = 0x2000u;

// Round 6: PAIRS of literals.  The reference writes its .cov.gz with gzopen(path, "w6h") — Z_HUFFMAN_ONLY (ptBlock.c:2271): a real bam2cov
// track is literals only, a dozen symbols (digits, tab, newline) with codes of 3-5 bits.  A second table indexed by the next kPairBits bits
// resolves TWO literals at once where both codes fit: bits 0-7 = bits of both codes, bits 8-15 / 16-23 = the two bytes, F_PAIR marks a
// valid entry.  One look-up and one 2-byte store per pair; up to five pairs per refill of the bit buffer.
constexpr int kPairBits = 10;
constexpr uint32_t F_PAIR = 0x80000000u;

struct Tables {
    uint32_t lit[(1 << kLitBits) + 4096];     // primary + second-level entries (far above what 288 symbols of <= 15 bits can need)
    uint32_t dist[(1 << kDistBits) + 2048];
    uint32_t pair[1 << kPairBits];
};

// the pair table out of the primary literal table: index i decodes to literal A (code length la <= kPairBits) followed by literal B when
// B's code fits the bits that are left (the primary entry of the remaining bits, zero-extended, is then decided by B's own bits alone)
inline void build_pairs(const uint32_t* lit, uint32_t* pair) {
    static_assert(kPairBits <= kLitBits, "pairs are read out of the primary table");
    for (uint32_t i = 0; i < (1u << kPairBits); i++) {
        uint32_t out = 0;
        const uint32_t a = lit[i];                      // (the primary table's index is kLitBits wide: the bits above kPairBits are zero here, and
        const uint32_t la = a & 0xffu;                  //  an entry with a code of <= kPairBits bits does not depend on them)
        if ((a & F_LIT) && !(a & F_SUB) && la >= 1 && la <= (uint32_t) kPairBits) {
            const uint32_t left = (uint32_t) kPairBits - la;
            const uint32_t b = lit[i >> la];
            const uint32_t lb = b & 0xffu;
            if ((b & F_LIT) && !(b & F_SUB) && lb >= 1 && lb <= left)
                out = F_PAIR | (la + lb) | (((a >> 16) & 0xffu) << 8) | (((b >> 16) & 0xffu) << 16);
        }
        pair[i] = out;
    }
}

inline uint32_t rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// canonical Huffman code of `n` symbols with lengths `len` -> look-up table indexed by the next `tb` bits of the stream (LSB first).
// payload(sym) gives bits 8-31 of a symbol's entry.  Returns false for an over-subscribed or (non-trivially) incomplete code.
template <class Payload>
inline bool build_table(const uint8_t* len, int n, int tb, uint32_t* tab, int tab_cap, Payload payload, bool allow_incomplete) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    int left = 1, nsyms = 0;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; nsyms += count[l]; }
    if (nsyms == 0) {                                           // no code at all: every look-up is an error (length 0 marks it)
        for (int i = 0; i < (1 << tb); i++) tab[i] = 0;
        return allow_incomplete;
    }
    int maxlen = 15;
    while (maxlen > 1 && !count[maxlen]) maxlen--;
    if (left > 0 && !(allow_incomplete && maxlen == 1)) return false;   // incomplete: only a single one-bit code (RFC 1951 3.2.7; zlib's rule)
    uint32_t next[16];
    {
        uint32_t code = 0;
        for (int l = 1; l <= 15; l++) { code = (code + (uint32_t) count[l - 1]) << 1; next[l] = code; }
    }
    for (int i = 0; i < (1 << tb); i++) tab[i] = 0;             // length 0 = invalid code
    // second-level tables: for every primary index, the longest code behind it
    uint8_t sub_bits[1 << kLitBits];
    std::memset(sub_bits, 0, (size_t) 1 << tb);
    uint32_t codes[288];
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        if (!l) continue;
        const uint32_t r = rev_bits(next[l]++, l);
        codes[s] = r;
        if (l > tb) { const uint32_t pi = r & ((1u << tb) - 1u); if (l - tb > sub_bits[pi]) sub_bits[pi] = (uint8_t) (l - tb); }
    }
    int used = 1 << tb;
    for (int pi = 0; pi < (1 << tb); pi++)
        if (sub_bits[pi]) {
            const int sz = 1 << sub_bits[pi];
            if (used + sz > tab_cap) return false;
            tab[pi] = F_SUB | ((uint32_t) used << 16) | ((uint32_t) sub_bits[pi] << 8) | (uint32_t) tb;
            for (int k = 0; k < sz; k++) tab[used + k] = 0;
            used += sz;
        }
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        if (!l) continue;
        const uint32_t r = codes[s];
        if (l <= tb) {
            const uint32_t e = payload(s) | (uint32_t) l;
            for (uint32_t k = r; k < (1u << tb); k += 1u << l) tab[k] = e;
        } else {
            const uint32_t pi = r & ((1u << tb) - 1u), pe = tab[pi];
            const int sb = (int) ((pe >> 8) & 0x1fu), base = (int) (pe >> 16);
            const uint32_t e = payload(s) | (uint32_t) (l - tb);
            for (uint32_t k = r >> tb; k < (1u << sb); k += 1u << (l - tb)) tab[base + (int) k] = e;
        }
    }
    return true;
}

inline uint32_t lit_payload(int s) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (s < 256) return F_LIT | ((uint32_t) s << 16);
    if (s == 256) return F_EOB;
    if (s > 285) return 0x1f00u;                                // 286, 287: never valid in a stream (31 extra bits marks it)
    return ((uint32_t) base[s - 257] << 16) | ((uint32_t) extra[s - 257] << 8);
}
inline uint32_t dist_payload(int s) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (s > 29) return 0x1f00u;
    return ((uint32_t) base[s] << 16) | ((uint32_t) extra[s] << 8);
}

struct Inflater {
    const uint8_t* in = nullptr;    // the whole compressed file
    size_t in_len = 0, pos = 0;     // pos: next byte to load into the bit buffer
    uint64_t bits = 0;
    int nbits = 0;
    // state between calls
    int block = 0;                  // 0: expect a block header; 1 stored; 2 Huffman
    bool final_block = false;
    uint32_t stored_left = 0;
    uint64_t total_out = 0;         // bytes of the current member produced so far (history available to matches)
    Tables t;
    uint8_t prev_len[288 + 32] = {0};   // the code lengths t was built from (valid while have_tables)
    bool have_tables = false;

    void reset_member() { bits = 0; nbits = 0; block = 0; final_block = false; stored_left = 0; total_out = 0; }

    // ---- speculative decoding (hf_io.cpp ParallelInflate): several decoders inside ONE stream ----
    // A stream without matches (the reference writes its .cov.gz with Z_HUFFMAN_ONLY: ptBlock.c:2271) has no history to resolve, so a decoder may
    // start at any block boundary.  stop_bit: run() returns AT_BOUNDARY at the first block boundary at or behind that bit of the input;
    // literal_only: a length / distance symbol ends the run with ERR_MATCH (the caller falls back to one decoder).
    size_t stop_bit = (size_t) -1;
    uint64_t stop_out = (uint64_t) -1;   // ... or at the first block boundary once the member has produced that many bytes
    bool literal_only = false;
    bool saw_match = false;          // run() decoded a match since the flag was cleared
    size_t bit_pos() const { return pos * 8 - (size_t) nbits; }
    // continue at bit `b` of the input, at a block header (`produced_so_far`: bytes of the member before it — what matches may reach back into)
    void seek_bit(size_t b, uint64_t produced_so_far) {
        pos = b >> 3; bits = 0; nbits = 0;
        refill();
        drop((int) (b & 7));
        block = 0; final_block = false; stored_left = 0; total_out = produced_so_far;
    }
    // Is there a plausible NON-FINAL DYNAMIC block header at bit b?  Everything read_block_header checks (code-length code and literal / distance
    // codes complete, an end-of-block code, repeats in range); the decoder is left behind that header, ready to decode the block.
    bool try_dynamic_header_at(size_t b) {
        if ((b >> 3) + 16 > in_len) return false;
        uint64_t v;
        std::memcpy(&v, in + (b >> 3), 8);
        v >>= (b & 7);
        if ((v & 7u) != 4u) return false;                       // BFINAL = 0, BTYPE = 2 (LSB first: bits 1-2 = 0b10)
        if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) return false;   // HLIT, HDIST
        seek_bit(b, 0);
        have_tables = false;                                    // (whatever a failed attempt left in the tables is not the last block's code)
        return read_block_header() == OK && block == 2 && !final_block;
    }

    inline void refill() {
        if (pos + 8 <= in_len) {
            uint64_t v;
            std::memcpy(&v, in + pos, 8);
            bits |= v << nbits;
            pos += (size_t) ((63 - nbits) >> 3);
            nbits |= 56;
        } else {
            while (nbits <= 56 && pos < in_len) { bits |= (uint64_t) in[pos++] << nbits; nbits += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t) (bits & ((1ull << n) - 1ull)); }
    inline void drop(int n) { bits >>= n; nbits -= n; }
    // bits that were loaded but do not exist (past the end of the file) show up as nbits going negative
    inline bool overrun() const { return nbits < 0; }

    // gzip member header at the current byte position (the bit buffer must be empty).  ERR_HEADER: not a gzip member.
    int read_gzip_header() {
        size_t p = pos;
        if (p + 10 > in_len) return ERR_TRUNCATED;
        if (in[p] != 0x1f || in[p + 1] != 0x8b || in[p + 2] != 8) return ERR_HEADER;
        const int flg = in[p + 3];
        if (flg & 0xe0) return ERR_HEADER;
        p += 10;
        if (flg & 4) { if (p + 2 > in_len) return ERR_TRUNCATED; const size_t xl = in[p] | ((size_t) in[p + 1] << 8); p += 2 + xl; }
        if (flg & 8) { while (p < in_len && in[p]) p++; p++; }
        if (flg & 16) { while (p < in_len && in[p]) p++; p++; }
        if (flg & 2) p += 2;
        if (p > in_len) return ERR_TRUNCATED;
        pos = p;
        reset_member();
        return OK;
    }
    // After a member's trailer: OK = the header of another member has been read; END_OF_MEMBER = the input ends here, or what follows
    // does not start with the gzip magic (trailing garbage: ignored, as gzread does); ERR_TRUNCATED = the bytes that follow START
    // a gzip member (1f 8b) but the file ends inside its header — a concatenated / bgzip file cut there must not load silently with
    // its last rows missing.
    int next_member() {
        if (pos >= in_len) return END_OF_MEMBER;
        const int rc = read_gzip_header();
        if (rc == OK) return OK;
        if (rc == ERR_TRUNCATED && in_len - pos >= 2 && in[pos] == 0x1f && in[pos + 1] == 0x8b) return ERR_TRUNCATED;
        return END_OF_MEMBER;
    }
    // after END_OF_MEMBER: CRC-32 and ISIZE of the trailer (the decoder has consumed whole bytes only up to the end of the last block)
    int read_gzip_trailer(uint32_t* crc, uint32_t* isize) {
        // give back the whole bytes still in the bit buffer
        const int whole = nbits >> 3;
        pos -= (size_t) whole;
        bits = 0; nbits = 0;
        if (pos + 8 > in_len) return ERR_TRUNCATED;
        *crc = in[pos] | ((uint32_t) in[pos + 1] << 8) | ((uint32_t) in[pos + 2] << 16) | ((uint32_t) in[pos + 3] << 24);
        *isize = in[pos + 4] | ((uint32_t) in[pos + 5] << 8) | ((uint32_t) in[pos + 6] << 16) | ((uint32_t) in[pos + 7] << 24);
        pos += 8;
        return OK;
    }

    int read_block_header() {
        refill();
        if (nbits < 3) return ERR_TRUNCATED;
        final_block = peek(1) != 0; drop(1);
        const int type = (int) peek(2); drop(2);
        if (type == 0) {
            drop(nbits & 7);                                     // to the next byte boundary
            refill();
            if (nbits < 32) return ERR_TRUNCATED;
            const uint32_t len = peek(16); drop(16);
            const uint32_t nlen = peek(16); drop(16);
            if ((len ^ 0xffffu) != nlen) return ERR_DATA;
            pos -= (size_t) (nbits >> 3);                        // the bytes of the block are copied straight from the input
            bits = 0; nbits = 0;
            stored_left = len;
            block = 1;
            return OK;
        }
        if (type == 3) return ERR_DATA;
        uint8_t len[288 + 32];
        int nlit, ndist;
        if (type == 1) {
            for (int i = 0; i < 144; i++) len[i] = 8;
            for (int i = 144; i < 256; i++) len[i] = 9;
            for (int i = 256; i < 280; i++) len[i] = 7;
            for (int i = 280; i < 288; i++) len[i] = 8;
            for (int i = 0; i < 32; i++) len[288 + i] = 5;
            nlit = 288; ndist = 32;
        } else {
            refill();
            if (nbits < 14) return ERR_TRUNCATED;
            nlit = (int) peek(5) + 257; drop(5);
            ndist = (int) peek(5) + 1; drop(5);
            const int ncl = (int) peek(4) + 4; drop(4);
            if (nlit > 286 || ndist > 30) return ERR_DATA;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncl; i++) {
                refill();
                if (nbits < 3) return ERR_TRUNCATED;
                cl[order[i]] = (uint8_t) peek(3); drop(3);
            }
            uint32_t cltab[1 << 7];
            if (!build_table(cl, 19, 7, cltab, 1 << 7, [](int s) { return (uint32_t) s << 16; }, false)) return ERR_DATA;
            int i = 0;
            while (i < nlit + ndist) {
                refill();
                const uint32_t e = cltab[peek(7)];
                const int l = (int) (e & 0xffu);
                if (!l) return ERR_DATA;
                drop(l);
                const int sym = (int) (e >> 16);
                if (sym < 16) { len[i++] = (uint8_t) sym; continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0) return ERR_DATA; val = len[i - 1]; rep = 3 + (int) peek(2); drop(2); }
                else if (sym == 17) { rep = 3 + (int) peek(3); drop(3); }
                else { rep = 11 + (int) peek(7); drop(7); }
                if (overrun()) return ERR_TRUNCATED;
                if (i + rep > nlit + ndist) return ERR_DATA;
                while (rep--) len[i++] = (uint8_t) val;
            }
            if (overrun()) return ERR_TRUNCATED;
            if (len[256] == 0) return ERR_DATA;                  // no end-of-block code
            // the distance lengths follow the literal/length lengths: move them to their own array position
            std::memmove(len + 288, len + nlit, (size_t) ndist);
            for (int k = nlit; k < 288; k++) len[k] = 0;
            for (int k = ndist; k < 32; k++) len[288 + k] = 0;
        }
        // (a Huffman-only stream starts a new block every 16 K literals, and the code of a coverage track — a dozen symbols — is often the
        // same as the block's before: the tables are rebuilt only when a code length changed; building them cost as much as decoding the block)
        if (have_tables && std::memcmp(len, prev_len, sizeof prev_len) == 0) { block = 2; return OK; }
        have_tables = false;
        if (!build_table(len, type == 1 ? 288 : nlit, kLitBits, t.lit, (int) (sizeof t.lit / 4), lit_payload, true)) return ERR_DATA;
        if (!build_table(len + 288, type == 1 ? 32 : ndist, kDistBits, t.dist, (int) (sizeof t.dist / 4), dist_payload, true)) return ERR_DATA;
        build_pairs(t.lit, t.pair);
        std::memcpy(prev_len, len, sizeof prev_len);
        have_tables = true;
        block = 2;
        return OK;
    }

    // Decode into out[0 .. cap): `hist` bytes before `out` are valid history of the same member (min(total_out, 32768) are needed).
    // Returns OK (cap nearly reached: call again with the next buffer), END_OF_MEMBER, or an error; *produced = bytes written.
    // At most cap bytes are written and nothing is read before out - hist.
    int run(uint8_t* out, size_t cap, size_t hist, size_t* produced) {
        uint8_t* o = out;
        uint8_t* const oend = out + cap;
        int rc = OK;
        for (;;) {
            if (block == 0) {
                if (final_block) { rc = END_OF_MEMBER; break; }
                if (bit_pos() >= stop_bit || total_out + (uint64_t) (o - out) >= stop_out) { rc = AT_BOUNDARY; break; }
                rc = read_block_header();
                if (rc != OK) break;
            }
            if (block == 1) {                                    // stored: the bit buffer holds whole bytes only
                size_t n = stored_left;
                if (n > (size_t) (oend - o)) n = (size_t) (oend - o);
                if (n > in_len - pos) { rc = ERR_TRUNCATED; break; }
                std::memcpy(o, in + pos, n);
                o += n; pos += n; stored_left -= (uint32_t) n;
                if (stored_left) break;                          // output full
                block = 0;
                continue;
            }
            // Huffman block: the bit buffer and the positions live in locals (stores through `o` may alias the members)
            bool full = false;
            uint64_t bb = bits;
            int nb = nbits;
            size_t ip = pos;
            const uint8_t* const inp = in;
            const size_t ilen = in_len;
            const uint32_t* const lt = t.lit;
            const uint32_t* const dt = t.dist;
#define HFZ_REFILL() do { \
                if (ip + 8 <= ilen) { uint64_t v_; std::memcpy(&v_, inp + ip, 8); bb |= v_ << nb; ip += (size_t) ((63 - nb) >> 3); nb |= 56; } \
                else { while (nb <= 56 && ip < ilen) { bb |= (uint64_t) inp[ip++] << nb; nb += 8; } } } while (0)
#define HFZ_PEEK(n) ((uint32_t) (bb & ((1ull << (n)) - 1ull)))
#define HFZ_DROP(n) do { bb >>= (n); nb -= (n); } while (0)
            const uint32_t* const pt = t.pair;
            for (;;) {
                if ((size_t) (oend - o) < 258 + 8) { full = true; break; }   // the next symbol might not fit
                HFZ_REFILL();
                {   // pairs of literals: up to five per refill (5 x kPairBits <= the 56 bits a refill guarantees)
                    uint32_t p2 = pt[HFZ_PEEK(kPairBits)];
                    if (p2 & F_PAIR) {
                        int left = 5;
                        do {
                            const uint16_t two = (uint16_t) (p2 >> 8);
                            std::memcpy(o, &two, 2); o += 2;
                            HFZ_DROP((int) (p2 & 0xffu));
                            p2 = pt[HFZ_PEEK(kPairBits)];
                        } while ((p2 & F_PAIR) && --left);
                        if (nb < 0) { rc = ERR_TRUNCATED; break; }
                        continue;                                   // room check and refill again
                    }
                }
                uint32_t e = lt[HFZ_PEEK(kLitBits)];
                if (e & F_SUB) { HFZ_DROP((int) (e & 0xffu)); e = lt[(e >> 16) + HFZ_PEEK((int) ((e >> 8) & 0x1fu))]; }
                int l = (int) (e & 0xffu);
                if (!l) { rc = nb < 0 ? ERR_TRUNCATED : ERR_DATA; break; }
                HFZ_DROP(l);
                if (e & F_LIT) {
                    *o++ = (uint8_t) (e >> 16);
                    // a second and a third literal out of the same refill (>= 56 - 15 bits are left)
                    e = lt[HFZ_PEEK(kLitBits)];
                    if ((e & (F_LIT | 0xffu)) > F_LIT) {        // a literal with a primary-table code
                        HFZ_DROP((int) (e & 0xffu)); *o++ = (uint8_t) (e >> 16);
                        e = lt[HFZ_PEEK(kLitBits)];
                        if ((e & (F_LIT | 0xffu)) > F_LIT) { HFZ_DROP((int) (e & 0xffu)); *o++ = (uint8_t) (e >> 16); }
                    }
                    if (nb < 0) { rc = ERR_TRUNCATED; break; }
                    continue;
                }
                if (e & F_EOB) { if (nb < 0) { rc = ERR_TRUNCATED; break; } block = 0; break; }
                const int xb = (int) ((e >> 8) & 0x1fu);
                if (xb == 0x1f) { rc = ERR_DATA; break; }
                if (literal_only) { rc = ERR_MATCH; break; }
                saw_match = true;
                const unsigned length = (unsigned) (e >> 16) + HFZ_PEEK(xb);
                HFZ_DROP(xb);
                if (nb < 32) HFZ_REFILL();                       // a distance needs at most 15 + 13 bits
                uint32_t d = dt[HFZ_PEEK(kDistBits)];
                if (d & F_SUB) { HFZ_DROP((int) (d & 0xffu)); d = dt[(d >> 16) + HFZ_PEEK((int) ((d >> 8) & 0x1fu))]; }
                l = (int) (d & 0xffu);
                if (!l) { rc = nb < 0 ? ERR_TRUNCATED : ERR_DATA; break; }
                HFZ_DROP(l);
                const int dxb = (int) ((d >> 8) & 0x1fu);
                if (dxb == 0x1f) { rc = ERR_DATA; break; }
                const size_t dist = (size_t) (d >> 16) + HFZ_PEEK(dxb);
                HFZ_DROP(dxb);
                if (nb < 0) { rc = ERR_TRUNCATED; break; }
                const size_t have = (size_t) (o - out) + hist;
                if (dist > have || dist > total_out + (size_t) (o - out)) { rc = ERR_DATA; break; }
                const uint8_t* s = o - dist;
                uint8_t* const stop = o + length;
                if (dist >= 8) {
                    do { std::memcpy(o, s, 8); o += 8; s += 8; } while (o < stop);   // may run up to 7 bytes past: room was checked
                } else {
                    do { *o++ = *s++; } while (o < stop);
                }
                o = stop;
            }
#undef HFZ_REFILL
#undef HFZ_PEEK
#undef HFZ_DROP
            bits = bb; nbits = nb; pos = ip;
            if (rc != OK || full) break;
        }
        *produced = (size_t) (o - out);
        total_out += *produced;
        return rc;
    }
};

}  // namespace hfz
