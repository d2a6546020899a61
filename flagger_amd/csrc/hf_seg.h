// hf_seg.h — HF_ALGO_SCAN: one WORKGROUP (one wavefront) per chunk segment, the whole forward / backward / decode of the
// segment in ONE launch (k_seg_fb<., true>: phases A-D below in one kernel, the segments of a chunk hand their products to
// each other inside the launch) — or in two (k_seg_prod = phase A, then k_seg_fb<., false> = B-D: the fall-back after a hand-off
// timed out, the choice of hf_create for chunks with more segments than the device holds workgroups, HF_SEG_LAUNCHES=2)
// (BASELINE north_star: "one contig-chunk per workgroup ... wavefront prefix-scan for the forward/backward recurrences").
//
// A chunk of T windows (hmm.c:333-545 runs it strictly sequentially) is cut into n = ceil(T / HF_SEG_SPLIT) equal SEGMENTS of
// at most 64*LMAX windows (one-wavefront workgroups: a CU is busy for the sum of its workgroups' steps, and they spread most
// evenly).  Inside a segment lane j owns the L = ceil(n_windows / 64) consecutive windows j*L .. j*L+L-1, in both directions:
//   A  (k_seg_prod) lane product Q_j = A_{jL} ... A_{jL+L-1}, A_t = T_t∘e_t: ONE precomputed 128-byte row per window (below);
//      the product of the whole segment for the chunk's other segments;
//   B  (k_seg_fb) prefix and suffix scans of Q over the 64 lanes (DPP row shifts / broadcasts, no LDS traffic inside a row of
//      16 lanes) and the products of the chunk's other segments (staged in LDS once, then two short chains): every lane gets
//      the normalised forward vector entering its first window and the direction of b at its last one;
//   C  forward REPLAY of the lane's windows (f·A, pre-inner sums, division by the scale, log: hmm.c:366-434): the carried-in
//      vector differs from a sequential run in the last ulp, and so may a term f·(T·e) from the reference's (f·T)·e.  The
//      lane's forward vectors and scales STAY IN REGISTERS (round 3: the loop is unrolled over the step index, <= 8 x 5
//      doubles) — nothing is written in this phase;
//   D  backward replay + posterior argmax (hmm.c:470-529, 671-692) reading f and the scales from those registers; the
//      magnitude of the carried-in b from the invariant sum_s f_t[s]·b_t[s]·scale_t = terminationProb of the scaled
//      forward-backward.  Every step writes one WHOLE pair record and one scale, fire and forget.
//
// Rows: every lane needs ITS OWN 128-byte row per window.  A lane reading its row with eight 16-byte loads touches
// 64 different cache lines per instruction and depends on the 32 KiB L1 keeping each line for the seven loads that follow —
// it does not (measured: ~1 000 cycles per wavefront and row).  Here the 64 rows of a step are fetched COOPERATIVELY with
// LDS-DMA (global_load_lds_dwordx4: instruction q moves rows 8q..8q+7 complete, 8 lanes x 16 bytes each, straight into the
// wavefront's 8 KiB LDS block — no staging registers) and every lane then reads its row with eight conflict-free
// ds_read_b128; the piece rotation that makes the reads conflict-free is applied on the SOURCE side of the DMA.  The row
// offsets a lane needs for the DMA (those of eight OTHER lanes) come from a per-segment LDS table written once, arranged
// [step][lane & 7][lane >> 3]: two ds_read_b128 per step (round 2 moved them with eight dependent ds_bpermute per step).
//
// Output = the PAIR RECORDS the statistics read (hf_rows.h by emission row, hf_chunks.h per chunk): record(t) = { f_{t-1}[4],
// b_t[4] }, 64 bytes, at the POSITION hf_create planned for window t (plan order: the statistics stream the records of a row of
// A; without a plan positions = slots), and the scales in SLOT order: window w of a segment (w = j*L + i) lives in slot
// slot0 + i*64 + j (512 contiguous bytes per store instruction; the host getters apply the same map).  f_{t-1} of a lane's
// first window is the previous lane's last forward vector (one shuffle), of a segment's first window the carried-in vector.
// Labels leave through LDS.
//
// Round 4: the kernel is STRAIGHT-LINE wherever a lane used to be branched around work that is the identity
// for it — scan levels multiply by an identity matrix where there is no neighbour, lanes past their last window multiply by an
// identity row of the table, lanes without a window store to the segment's spare record — and renormalises once per two products
// from an integer maximum; every piece has a -DHF_SEG_... switch below for same-box A/B runs (profiles/tools/build_variants.sh).
#pragma once
#include "hf_scan.h"

#define HF_SEG_LMAX 8                        // windows per lane at most: a chunk longer than 64*HF_SEG_LMAX windows is split
#define HF_SEG_SPLIT (64 * HF_SEG_LMAX)      // windows per segment a chunk is cut by (equal parts of at most this)
#define HF_SEG_PSTAGE 24                     // segment products of a chunk staged in LDS by k_seg_fb (the rest: global loads)
#ifndef HF_SEG_OCC
#define HF_SEG_OCC 3        // wavefronts per SIMD the register allocation of k_seg_fb aims at (168 VGPRs: the lane's 8 x 5 doubles stay in registers)
#endif

// SegDesc: hf_device.h

// one-launch mode (k_seg_fb<., true>): hand-off of a segment's product to the chunk's other segments — write-through stores,
// a drained queue, then the flag (= the launch's epoch); readers poll the flag and read past their caches
// The hand-offs here and in hf_rows.h order "data stores, then flag / ticket" with relaxed atomics + s_waitcnt vmcnt(0): on GFX9 /
// CDNA vmcnt counts stores as well as loads, so the wait drains them; gfx10+ tracks stores in vscnt and the same code would publish
// a flag before its data.  This library is written for gfx950 only — refuse any other device target at compile time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "hf_seg.h / hf_rows.h: the in-launch hand-offs rely on GFX9 (CDNA) vmcnt semantics; build with --offload-arch=gfx950"
#endif
#define HF_SEG_SPIN_MAX (1 << 16)            // polls of one flag before the wait is given up (HF_FLAG_SYNC: the host falls back to two launches)
__device__ __forceinline__ void seg_xcu_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double seg_xcu_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Round 6 (VERDICT r05 #2a): a segment product travels as 8 x 16-byte `sc0 sc1` accesses instead of 16 + 16 eight-byte system-scope atomics
// (MI355X_MICROARCH.md, Workgroup dispatch: a dwordx2 sc1 store costs 2.7x a dwordx4's time per byte, 8-byte sc1 loads run at 0.54-0.70x the
// 16-byte rate).  Same scope bits as the atomics' lowering, same protocol: write-through data, drained queue, then the flag; the reader polls
// the flag, then loads past its caches.  Inline assembly: hipcc has no builtin for a 16-byte store with both scope bits (a volatile access
// gets them, with an s_waitcnt vmcnt(0) behind EVERY access).  -DHF_SEG_WIDE=0: the 8-byte atomics of rounds 3-5 (same-box A/B).
#ifndef HF_SEG_WIDE
#define HF_SEG_WIDE 1
#endif
#ifndef HF_SEG_STAGGER
#define HF_SEG_STAGGER 0           // cycles between the starts of the grid's parts (0: all workgroups start together)
#endif
#ifndef HF_SEG_STAGGER_PARTS
#define HF_SEG_STAGGER_PARTS 2
#endif
typedef unsigned hf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void seg_xcu_store_m4(double* p, const double m[16]) {
#if HF_SEG_WIDE
#pragma unroll
    for (int k = 0; k < 8; k++) {
        hf_u32x4 v;
        v.x = (unsigned) __double2loint(m[2 * k]); v.y = (unsigned) __double2hiint(m[2 * k]);
        v.z = (unsigned) __double2loint(m[2 * k + 1]); v.w = (unsigned) __double2hiint(m[2 * k + 1]);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p + 2 * k), "v"(v) : "memory");
    }
#else
#pragma unroll
    for (int k = 0; k < 16; k++) seg_xcu_store(p + k, m[k]);
#endif
}
// (the loads and their wait in ONE asm statement: the compiler does not count an asm load on its vmcnt scoreboard, so the registers are only
// valid behind the statement's own s_waitcnt; nothing else of the wavefront is in flight at the two places this is called from)
__device__ __forceinline__ void seg_xcu_load_m4(const double* p, double m[16]) {
#if HF_SEG_WIDE
    hf_u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %8, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off offset:48 sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:64 sc0 sc1\n\t"
                 "global_load_dwordx4 %5, %8, off offset:80 sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:96 sc0 sc1\n\t"
                 "global_load_dwordx4 %7, %8, off offset:112 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(p) : "memory");
    const hf_u32x4 v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
    for (int k = 0; k < 8; k++) { m[2 * k] = __hiloint2double((int) v[k].y, (int) v[k].x); m[2 * k + 1] = __hiloint2double((int) v[k].w, (int) v[k].z); }
#else
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = seg_xcu_load(p + k);
#endif
}

// -DHF_SEG_TRACE: s_memtime stamps of k_seg_fb<true>'s phases, one row of HF_SEG_TRACE_N words per workgroup, dumped by
// hf_destroy to $HF_SEG_TRACE_FILE (profiles/tools/seg_trace.sh / seg_trace.py).  Not in a normal build.
#define HF_SEG_TRACE_N 25
#ifdef HF_SEG_TRACE
__device__ unsigned long long* g_seg_trace = nullptr;
#define TR_DECL unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tr_t = clock64(); (void) tr_acc; (void) tr_t
#define TR_STAMP(k) do { if (BWD && g_seg_trace && lane == 0) g_seg_trace[(size_t) g * HF_SEG_TRACE_N + (k)] = clock64(); } while (0)
#define TR_LAP(k) do { const unsigned long long n_ = clock64(); tr_acc[k] += n_ - tr_t; tr_t = n_; } while (0)
#define TR_RESET() do { tr_t = clock64(); } while (0)
#define TR_FLUSH(k0, n) do { if (BWD && g_seg_trace && lane == 0) for (int q_ = 0; q_ < (n); q_++) g_seg_trace[(size_t) g * HF_SEG_TRACE_N + (k0) + q_] = tr_acc[q_]; } while (0)
#else
#define TR_DECL
#define TR_STAMP(k)
#define TR_LAP(k)
#define TR_RESET()
#define TR_FLUSH(k0, n)
#endif

// 1 / sc of the replays.  The scale of a window is a normal number (> 1e-50: hmm.c:412-415 raises an error below; at a chunk's first window
// >= 0.25e-40 by the emission floor), so the special cases that the IEEE division's expansion carries (v_div_scale / v_div_fmas / v_div_fixup:
// twelve instructions) cannot occur.  -DHF_SEG_FASTRCP=1: v_rcp_f64 and two Newton steps (five instructions, within 1 ulp of the quotient: the
// product nf * (1 / sc) already differs from nf / sc in the last bit); measured in round 6: profiles/r06_ab_micro.txt.
#ifndef HF_SEG_FASTRCP
#define HF_SEG_FASTRCP 0
#endif
__device__ __forceinline__ double seg_recip(double sc) {
#if HF_SEG_FASTRCP
    double r = __builtin_amdgcn_rcp(sc);
    r = fma(fma(-sc, r, 1.0), r, r);
    r = fma(fma(-sc, r, 1.0), r, r);
    return r;
#else
    return 1.0 / sc;
#endif
}
__device__ __forceinline__ void v4_renorm(double v[4]) {
    int e;
    (void) frexp(fmax(fmax(v[0], v[1]), fmax(v[2], v[3])), &e);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = ldexp(v[k], -e);
}
// v <- v·M (row vector), M row-major [pre*4 + s]
__device__ __forceinline__ void v4_mul_right(double v[4], const double* __restrict__ M) {
    double u[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double s = v[0] * M[j];
        s = fma(v[1], M[4 + j], s); s = fma(v[2], M[8 + j], s); s = fma(v[3], M[12 + j], s);
        u[j] = s;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = u[j];
}
// v <- M·v (column vector)
__device__ __forceinline__ void v4_mul_left(double v[4], const double* __restrict__ M) {
    double u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double s = M[i * 4] * v[0];
        s = fma(M[i * 4 + 1], v[1], s); s = fma(M[i * 4 + 2], v[2], s); s = fma(M[i * 4 + 3], v[3], s);
        u[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = u[i];
}

// power-of-two renormalisation: the largest entry into [0.5, 1).  Entries are probabilities (>= 0, finite, or NaN), so the order of
// the doubles is the order of their high words: an integer maximum (v_max3_u32) gives the exponent without 15 v_max_f64 and a frexp.
// A largest entry that is zero or denormal leaves the matrix as it is; a NaN entry stays a NaN (and reaches the scale of its window).
__device__ __forceinline__ void m4_renorm_tree(M4& a) {
    unsigned h = (unsigned) __double2hiint(a.m[0]);
#pragma unroll
    for (int i = 1; i < 16; i++) { const unsigned g = (unsigned) __double2hiint(a.m[i]); h = g > h ? g : h; }
    const int be = (int) ((h >> 20) & 0x7ffu);
    int ne = be ? 1022 - be : 0;                 // minus frexp's exponent of a normal number
    asm volatile("" : "+v"(ne));                 // (ONE select, on the exponent: hipcc otherwise selects between x and ldexp(x) sixteen times)
#pragma unroll
    for (int i = 0; i < 16; i++) a.m[i] = ldexp(a.m[i], ne);
}

// ---- DPP moves of a 4x4 matrix (gfx9 row_shr / row_shl / row_bcast): lanes without a source keep their own value ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the same move with 0.0 for lanes without a source (row shifts: bound_ctrl; broadcasts: rows outside the mask)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp0_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
// the same move with `oldv` for the lanes the move does not write
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_old_f64(double oldv, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(oldv), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(oldv), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ void m4_dpp(M4& dst, const M4& src) {
#pragma unroll
    for (int i = 0; i < 16; i++) dst.m[i] = dpp_f64<CTRL, ROW_MASK>(src.m[i]);
}
#define HF_DPP_ROW_SHR(n) (0x110 + (n))
#define HF_DPP_ROW_SHL(n) (0x100 + (n))
#define HF_DPP_ROW_BCAST15 0x142
#define HF_DPP_ROW_BCAST31 0x143

#define HF_DPP_WAVE_SHR1 0x138
#define HF_DPP_WAVE_SHL1 0x130
// a level inside a row of 16 lanes.  HF_SEG_IDLANE: the neighbour's matrix arrives with zeros where there is no neighbour (bound_ctrl),
// the diagonal is set to one there, and EVERY lane multiplies — by the identity where the level does not apply: exact, no branch, no
// copy of the product back into place.
__device__ __forceinline__ void m4_dpp_or_identity_fix(M4& X, bool has_source) {
#pragma unroll
    for (int d = 0; d < 16; d += 5) X.m[d] = __hiloint2double(has_source ? __double2hiint(X.m[d]) : 0x3ff00000, __double2loint(X.m[d]));   // (the low word is 0 already)
}
template <int CTRL>
__device__ __forceinline__ void m4_dpp0(M4& dst, const M4& src) {
#pragma unroll
    for (int i = 0; i < 16; i++) dst.m[i] = dpp0_f64<CTRL, 0xf>(src.m[i]);
}
// inclusive prefix product over the 64 lanes: lane l ends with Q_0 ... Q_l (power-of-two renormalised: exact)
__device__ __forceinline__ void m4_scan_prefix(M4& Pq, int lane) {
    M4 Lft, R;
#define HF_STEP_SHR(n, RN)                                                                              \
    m4_dpp0<HF_DPP_ROW_SHR(n)>(Lft, Pq);                                                                 \
    m4_dpp_or_identity_fix(Lft, (lane & 15) >= (n));                                                    \
    m4_mul(R, Lft, Pq); Pq = R;                                                                         \
    if (RN) m4_renorm_tree(Pq);
    HF_STEP_SHR(1, false) HF_STEP_SHR(2, 1) HF_STEP_SHR(4, false) HF_STEP_SHR(8, 1)
#undef HF_STEP_SHR
    // the rows the broadcast does not write keep the `old` operand: the identity — every lane multiplies, no branch, no copy
#pragma unroll
    for (int i = 0; i < 16; i++) Lft.m[i] = dpp_old_f64<HF_DPP_ROW_BCAST15, 0xa>((i % 5 == 0) ? 1.0 : 0.0, Pq.m[i]);   // rows 1, 3 <- lane 15 of rows 0, 2
    m4_mul(R, Lft, Pq); Pq = R;
    if (false) m4_renorm_tree(Pq);
#pragma unroll
    for (int i = 0; i < 16; i++) Lft.m[i] = dpp_old_f64<HF_DPP_ROW_BCAST31, 0xc>((i % 5 == 0) ? 1.0 : 0.0, Pq.m[i]);   // rows 2, 3 <- lane 31
    m4_mul(R, Lft, Pq); Pq = R;
    m4_renorm_tree(Pq);
}
// the four levels of the suffix scan inside a row of 16 lanes: lane l ends with Q_l ... Q_(last lane of its row)
__device__ __forceinline__ void m4_scan_suffix_rows(M4& Sq, int lane) {
    M4 Rgt, R;
#define HF_STEP_SHL(n, RN)                                                                              \
    m4_dpp0<HF_DPP_ROW_SHL(n)>(Rgt, Sq);                                                                 \
    m4_dpp_or_identity_fix(Rgt, (lane & 15) + (n) < 16);                                                \
    m4_mul(R, Sq, Rgt); Sq = R;                                                                         \
    if (RN) m4_renorm_tree(Sq);
    HF_STEP_SHL(1, false) HF_STEP_SHL(2, 1) HF_STEP_SHL(4, false) HF_STEP_SHL(8, 1)
#undef HF_STEP_SHL
}
// inclusive suffix product over the 64 lanes: lane l ends with Q_l ... Q_63 (without HF_SEG_VECSUF)
__device__ __forceinline__ void m4_scan_suffix(M4& Sq, int lane) {
    M4 Rgt, R;
    m4_scan_suffix_rows(Sq, lane);
#pragma unroll
    for (int i = 0; i < 16; i++) Rgt.m[i] = __shfl(Sq.m[i], (lane | 15) + 1);   // first lane of the next row
    if (!(lane & 16)) { m4_mul(R, Sq, Rgt); Sq = R; if (false) m4_renorm_tree(Sq); }       // rows 0, 2
#pragma unroll
    for (int i = 0; i < 16; i++) Rgt.m[i] = __shfl(Sq.m[i], 32);
    if (lane < 32) { m4_mul(R, Sq, Rgt); Sq = R; if (false) m4_renorm_tree(Sq); }
    if (true) m4_renorm_tree(Sq);
}

// ---- LDS of a segment workgroup: nc CACHED row blocks | the streaming row block | the offset table [LMAX][8][8] u32 | the labels [64*LMAX] ----
// nc = 0 (a device full of segments: twelve workgroups per CU = 3 per SIMD, what k_seg_fb's registers allow): 10.5 KiB — 13 KiB was measured
// to admit only eleven.  Round 5: a context with FEWER segments than the device holds at that rate gives every workgroup the LDS that
// would otherwise lie idle (hf_create: the largest nc at which all segments are still resident together): the rows of the lane's first nc
// steps stay in LDS blocks of their own from the first walk (the lane products) on, and the two replays read them there instead of
// fetching them again — 24 dependent fetch -> compute steps per workgroup become 8 + 2 (8 - nc), and the first walk's fetches of the cached
// steps are all in flight together.  That is the regime of the reference's own default (16 kb windows: ~380 k windows for a human diploid
// assembly) and of every rank's share at 8 GPUs.  The streaming block doubles as the workgroup's scratch (parked matrices, the chunk's
// segment products, the record transposition): cached blocks are never written after the first walk.
__host__ __device__ constexpr size_t seg_lds_bytes(int nc = 0) { return (size_t) (nc + 1) * 8192 + (size_t) 64 * HF_SEG_LMAX * 5; }
static_assert((size_t) HF_SEG_PSTAGE * 128 <= 8192, "the staged segment products fit the row block");
static_assert(64 * HF_SEG_LMAX >= 3 * 128, "the label area holds three parked matrices (seg_suffix_side)");

// the segment's row indices, read coalesced (lane l takes windows l, 64 + l, ...)
// (windows past the segment's end get `ident`, the table's identity row: hf_create writes it once behind the rows of A)
__device__ __forceinline__ void seg_load_arows(const int32_t* __restrict__ arow_seg, int n, int L, int lane, int32_t ident, int32_t rr[HF_SEG_LMAX]) {
#pragma unroll
    for (int c = 0; c < HF_SEG_LMAX; c++) { const int w = c * 64 + lane; rr[c] = (c < L && w < n) ? arow_seg[w] : ident; }
}
// ... and filed as BYTE OFFSETS of the rows (index << 7; hf_create keeps n_arows < 2^25) where the cooperative fetch reads
// them: window w = j*L + i (lane j's i-th) at [i][j & 7][j >> 3]; windows past the segment's end point at the identity row
__device__ __forceinline__ void seg_offsets_store(const int32_t rr[HF_SEG_LMAX], int L, int lane, uint32_t* __restrict__ s_off) {
    const uint32_t inv = (65536u + (uint32_t) L - 1u) / (uint32_t) L;   // w / L for w < 512, L <= 8: (w * inv) >> 16, exact
#pragma unroll
    for (int c = 0; c < HF_SEG_LMAX; c++)
        if (c < L) {
            const uint32_t w = (uint32_t) (c * 64 + lane), j = (w * inv) >> 16, i = w - j * (uint32_t) L;
            s_off[(i * 8 + (j & 7u)) * 8 + (j >> 3)] = ((uint32_t) rr[c] & 0x7fffffffu) << 7;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- cooperative row fetch through the wavefront's 8 KiB LDS blocks (see the header) ----
// Row r of the step (the row lane r needs) occupies bytes r*128 .. r*128+127 of the step's block; piece p of the row sits in slot
// (p + (r >> 1)) & 7: the eight ds_read_b128 of a lane group then cover all 64 banks exactly once.
// Step i lives in block i when i < nc (cached: fetched once per pass), in block nc (the streaming block) otherwise.
struct RowFetch {
    const char* __restrict__ base;          // the table of rows of A (wave-uniform)
    const uint32_t* __restrict__ my_off;    // this lane's eight offsets of step 0: s_off + (lane >> 3) * 8; step i: + i * 64
    uint32_t lds0;                          // LDS byte address of block 0 (wave-uniform)
    int nc;                                 // cached steps
    uint32_t off_even, off_odd;             // the 16-byte piece this lane moves, for even / odd instructions
};
__device__ __forceinline__ int rows_block(const RowFetch& F, int step) { return step < F.nc ? step : F.nc; }
__device__ __forceinline__ RowFetch rowfetch_init(const double* rows, const uint32_t* __restrict__ s_off, double* s_rows, int nc, int lane) {
    // lane (sub, part) of instruction q moves 16 bytes of row r = 8q + sub: piece (part - (r >> 1)) & 7 = (c0 - 4q) & 7, i.e.
    // one of two values; 32-bit byte offsets from the (wave-uniform) table base
    const int part = lane & 7, sub = lane >> 3;
    const uint32_t c0 = (uint32_t) (part - (sub >> 1));
    RowFetch F;
    F.base = reinterpret_cast<const char*>(rows); F.my_off = s_off + sub * 8;
    F.lds0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (uintptr_t) (__attribute__((address_space(3))) char*) s_rows);
    F.nc = nc;
    F.off_even = (c0 & 7u) << 4; F.off_odd = ((c0 + 4u) & 7u) << 4;
    return F;
}
__device__ __forceinline__ void rows_issue(const RowFetch& F, int step) {
    const uint4* __restrict__ t = reinterpret_cast<const uint4*>(F.my_off + step * 64);
    const uint4 o0 = t[0], o1 = t[1];
    const uint32_t o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
    const uint32_t lb = F.lds0 + (uint32_t) rows_block(F, step) * 8192u;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t off = o[q] | ((q & 1) ? F.off_odd : F.off_even);
        // The DMA is inline assembly on purpose: with the builtin, hipcc (ROCm 7.2) keeps a pending LDS write on its vmcnt
        // scoreboard and turns every wait before a later ds_read into vmcnt(0) — which would also wait for the backward
        // replay's record stores.  M0 = LDS address of the instruction's 1 KiB; nothing else in these kernels uses M0.
        // Unknown to the scoreboard, the DMA can only make the compiler's own counted waits longer.
        const uint32_t la = lb + (uint32_t) q * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(off), "s"(F.base) : "memory");
    }
}
// wait until at most `newer_steps` row fetches (8 DMA instructions each) issued AFTER the one that is needed are still in flight
// (vmcnt counts vector-memory instructions in issue order; the immediate is a constant: a wave-uniform switch)
__device__ __forceinline__ void rows_wait(int newer_steps) {
#define HF_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (((n) >> 4) << 14)) | (7 << 4) | (15 << 8))
    switch (newer_steps) {
        case 0: HF_VMCNT(0); break;  case 1: HF_VMCNT(8); break;  case 2: HF_VMCNT(16); break; case 3: HF_VMCNT(24); break;
        case 4: HF_VMCNT(32); break; case 5: HF_VMCNT(40); break; case 6: HF_VMCNT(48); break; default: HF_VMCNT(56); break;
    }
#undef HF_VMCNT
}
// the lane's row of `step` out of its block (the caller has waited for the fetch: rows_wait; a cached step needs no wait after the first walk)
// (`s_rows` must NOT be declared __restrict__: the rows arrive through the LDS-DMA of rows_issue — inline assembly)
__device__ __forceinline__ void rows_read(const RowFetch& F, const double* s_rows, int step, int lane, double E[16]) {
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double2* row = reinterpret_cast<const double2*>(s_rows + rows_block(F, step) * 1024) + lane * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 d = row[(k + (lane >> 1)) & 7]; E[2 * k] = d.x; E[2 * k + 1] = d.y; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0) only: the reads have returned before the block is refilled
    __builtin_amdgcn_wave_barrier();
}
// park / fetch a lane's matrix in the same block, same rotation (the block is idle during the scans)
__device__ __forceinline__ void m4_park(const M4& Q, int lane, double* __restrict__ blk) {
    double2* __restrict__ row = reinterpret_cast<double2*>(blk) + lane * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) row[(k + (lane >> 1)) & 7] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
}
__device__ __forceinline__ void m4_unpark(M4& Q, int lane, const double* __restrict__ blk) {
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const double2* __restrict__ row = reinterpret_cast<const double2*>(blk) + lane * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 d = row[(k + (lane >> 1)) & 7]; Q.m[2 * k] = d.x; Q.m[2 * k + 1] = d.y; }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}

// posterior argmax (hmm.c:671-692, common.c:292-304: strict >, first maximum wins) without the four divisions in the
// common case: q[s] = fl(p[s] / total) is monotone in p[s], and when the largest p exceeds every other one by more than
// 2^-49 relative, its quotient (>= 1/4 of a normal number) stays strictly the largest after rounding — the reference's
// answer.  Near-ties, an all-zero or a NaN posterior take the reference's own arithmetic (posterior_label).
__device__ __forceinline__ int posterior_label_fast(const double f[4], const double b[4], double sc) {
    double p[4];
#pragma unroll
    for (int s = 0; s < 4; s++) p[s] = f[s] * b[s] * sc;
    const double total = ((p[0] + p[1]) + p[2]) + p[3];
    double mx = p[0]; int idx = 0;
#pragma unroll
    for (int s = 1; s < 4; s++) if (mx < p[s]) { mx = p[s]; idx = s; }
    const double thr = mx * (1.0 - 0x1p-49);
    bool close = !(total == total) || !(mx > 0.0);
#pragma unroll
    for (int s = 0; s < 4; s++) close |= (s != idx) && (p[s] >= thr);
    if (close) return posterior_label(f, b, sc);
    return idx;
}

// ------------------------------------------------------------------------------------------
// Rows of A_t = T_t∘e_t.  The transition factor of a window depends on its region, its validity mask (3 bits), a region
// change and chunk-first-ness only (hmm_utils.c:2278-2292, hmm.c:398-400, 333-364) — iteration-invariant CLASSES; together
// with the emission key (region, x, x_prev) of hf_scan.h that makes a few thousand distinct 4x4 matrices per pass.
// They are multiplied out once per pass (row = class table ∘ emission row of this iteration's tables; one row per
// (key, class) that occurs at an interior window, one per slow window: by k_tables itself, hf_scan.h, or — after
// k_tables_nb — by k_arows), and the segment kernels fetch ONE 128-byte row per window and step: no transition-table
// lookup, no second factor, no region tables in LDS, and the row of a window is a precomputed index (hf_create: d_arow,
// bit 31 = chunk-first) instead of a function of two records and a slow-list rank.
// The product f·(T·e) differs from the reference's (f·T)·e in the last bit of a term; the segment kernels never were
// bit-identical to a sequential run (the carried-in vectors differ in the last bit already).
// A NaN in a row (hmm_utils.c:783-786) reaches the scale of the window that uses it: k_seg_fb raises HF_FLAG_NAN there.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_arows(int n_rows, const int32_t* __restrict__ src, const int32_t* __restrict__ cls,
                                               const double* __restrict__ lutE, const DevParams* __restrict__ P,
                                               double* __restrict__ lutA) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int id = (int) (i >> 4), o = (int) (i & 15);          // o = s*4 + pre (state-major, HF_PS)
    if (id >= n_rows) return;
    const int c = cls[id], k = c & 0xff, pre = o & 3, st = o >> 2;
    const DevRegion* __restrict__ R = &P->reg[c >> 8];
    const double t = k == 9 ? R->trans[4][st] : (k == 8 ? 1.0 / (HF_NSTATES + 1) : R->tcond[k][pre * 4 + st]);
    lutA[(int64_t) id * 16 + o] = t * lutE[(int64_t) src[id] * 16 + o];
}

// Suffix side of phase B.  Without HF_SEG_VECSUF: the inclusive suffix scan over the 64 lanes, xs = the product of the lanes after
// this one.  With it: only the four levels inside the rows of 16 lanes; xs = the product of the lanes after this one IN ITS ROW (the
// identity for a row's last lane), and the products of rows 1..3 (the inclusive suffix of their first lanes) parked in LDS (s_T, 3 x 16
// doubles in the label area, idle until phase D): the end vector is later carried across the rows as a VECTOR (seg_suffix_apply),
// 3 x 16 multiply-adds instead of two levels of 64 (+ 64 ds_bpermute + renormalisations) — what is consumed is suffix·u, never the suffix.
__device__ __forceinline__ void seg_suffix_side(M4& Q, int lane, double xs[16], double* s_T) {
    m4_scan_suffix_rows(Q, lane);
    if ((lane & 15) == 0 && lane > 0) {
        double2* dst = reinterpret_cast<double2*>(s_T) + ((lane >> 4) - 1) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    M4 X;
    m4_dpp0<HF_DPP_ROW_SHL(1)>(X, Q);
    m4_dpp_or_identity_fix(X, (lane & 15) != 15);
#pragma unroll
    for (int k = 0; k < 16; k++) xs[k] = X.m[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// direction of b at the lane's last window: everything after it (in the segment: xs and the later rows; then u, the end vector carried
// in through the chunk's later segments) applied to u
__device__ __forceinline__ void seg_suffix_apply(const double u[4], const double xs[16], int lane, const double* s_T, double bdir[4]) {
#pragma unroll
    for (int s = 0; s < 4; s++) bdir[s] = u[s];
    const int row = lane >> 4;
#pragma unroll
    for (int k = 3; k >= 1; k--) {                               // z_(k-1) = T_k · z_k for the lanes of rows < k
        double M[16], z[4] = {bdir[0], bdir[1], bdir[2], bdir[3]};
        const double2* src = reinterpret_cast<const double2*>(s_T) + (k - 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; q++) { const double2 t2 = src[q]; M[2 * q] = t2.x; M[2 * q + 1] = t2.y; }
        v4_mul_left(z, M);
#pragma unroll
        for (int s = 0; s < 4; s++) bdir[s] = row < k ? z[s] : bdir[s];
    }
    v4_renorm(bdir);
    v4_mul_left(bdir, xs);                                       // (a row's last lane: the identity)
}

// ------------------------------------------------------------------------------------------
// The lane product Q_j = A_{jL} ... A_{jL+L-1} (phase A; k_seg_prod and the one-launch k_seg_fb run the same code, so their results are
// the same bits).  A chunk-first window is left out of its lane's product: it belongs to the start vector.  All 64 lanes run all L
// steps (the row fetch is cooperative).  The steps past a lane's last window fetch the table's IDENTITY row and are multiplied like any
// other (exact), the first factor is taken as it is, and the loop runs two products per trip with the roles of Q and R swapped — no
// branch, no copy of a product back into place, one power-of-two renormalisation per trip.
// Fetch schedule: the steps that have a block of their own (the cached ones, and the first step of the streaming block) are issued
// together, at most seven steps = 56 DMA instructions in flight (vmcnt counts 63); a further step is issued as soon as its block is free.
// On entry nothing has been issued; on return nothing is in flight and blocks 0 .. nc-1 hold the rows of steps 0 .. nc-1.
// ------------------------------------------------------------------------------------------
template <bool CACHED>
__device__ __forceinline__ void seg_lane_product(const RowFetch& F, const double* s_rows, int lane, int L, bool chunk_first, M4& Q) {
    double E[16];
    M4 A, R;
#define HF_ROW_TO_M4(dst) _Pragma("unroll") for (int k_ = 0; k_ < 16; k_++) (dst).m[k_] = E[HF_PS(k_ >> 2, k_ & 3)]
    int issued = 1;
    if constexpr (CACHED) {
        issued = F.nc + 1 < L ? F.nc + 1 : L;
        if (issued > 7) issued = 7;
        for (int st = 0; st < issued; st++) rows_issue(F, st);
    } else rows_issue(F, 0);
    // step i: wait for its rows, read them, issue the next step whose block is free (a block of its own, or the streaming block once
    // the step before it has been read out).  Without cached blocks: one block, the next step's fetch issued behind every read — no counting
    auto step_rows = [&](int i) {
        if constexpr (CACHED) {
            rows_wait(issued - i - 1);
            rows_read(F, s_rows, i, lane, E);
            if (issued < L && (issued <= F.nc || issued - 1 <= i)) { rows_issue(F, issued); issued++; }
        } else {
            rows_wait(0);
            rows_read(F, s_rows, i, lane, E);
            if (i + 1 < L) rows_issue(F, i + 1);
        }
    };
    step_rows(0);
    HF_ROW_TO_M4(Q);                                             // the first factor: no product with the identity
    if (chunk_first) m4_identity(Q);                             // (lane 0 of a chunk's first segment)
    int i = 1;
#pragma unroll 1
    for (; i + 1 < L; i += 2) {
        step_rows(i);
        HF_ROW_TO_M4(A);
        m4_mul(R, Q, A);
        step_rows(i + 1);
        HF_ROW_TO_M4(A);
        m4_mul(Q, R, A);
        m4_renorm_tree(Q);
    }
    if (i < L) {                                                 // L even: one more factor
        step_rows(i);
        HF_ROW_TO_M4(A);
        m4_mul(R, Q, A);
        Q = R;
        m4_renorm_tree(Q);
    }                                                            // (L == 1: the row as it is; the scans renormalise)
#undef HF_ROW_TO_M4
}

// ------------------------------------------------------------------------------------------
// k_seg_prod: phase A for every segment: the lane products (lane-minor: 1 KiB per store instruction; k_seg_fb's scans start
// from them) and the product of the whole segment (used by the chunk's OTHER segments only).  A chunk-first window is left
// out of its lane's product: it belongs to the start vector.  All 64 lanes run all L steps (the row fetch is cooperative);
// lanes past their last window fetch row 0 and skip the arithmetic.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64, 4) k_seg_prod(const SegDesc* __restrict__ sd, const int32_t* __restrict__ arow,
                                                    const double* __restrict__ lutA, double* __restrict__ Qs, double* __restrict__ Pseg) {
    extern __shared__ __attribute__((aligned(16))) double s_W[];
    double* __restrict__ blk = s_W;
    uint32_t* __restrict__ s_off = reinterpret_cast<uint32_t*>(s_W + 1024);
    const int g = blockIdx.x, lane = threadIdx.x;
    const SegDesc d = sd[g];
    const int L = d.L, a = lane * L;
    {
        int32_t rr[HF_SEG_LMAX];
        seg_load_arows(arow + d.t0, d.n, L, lane, d.ident_row, rr);
        seg_offsets_store(rr, L, lane, s_off);
    }
    const RowFetch F = rowfetch_init(lutA, s_off, blk, 0, lane);
    M4 Q;
    seg_lane_product<false>(F, blk, lane, L, a == 0 && d.k == 0, Q);   // the chunk's first window starts the chain (hmm.c:333-364)
    {
        double2* __restrict__ dst = reinterpret_cast<double2*>(Qs) + (int64_t) g * 8 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k * 64] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    if (d.nseg == 1) return;                                  // nobody reads the product of a one-segment chunk
    m4_scan_prefix(Q, lane);                                  // lane 63: the product of the segment
    if (lane == 63) {
        double2* dst = reinterpret_cast<double2*>(Pseg + (int64_t) g * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
}

// ------------------------------------------------------------------------------------------
// k_seg_fb: one workgroup per segment: phases B-D of the header.  BWD = false: forward only (EM_runForwardForList,
// hmm.c:790-816): log-likelihood and error flags, nothing else is written.
// ------------------------------------------------------------------------------------------
// CACHED = false: nc is the constant 0 (a device full of segments: BASELINE configs[2]) — with nc a run-time value the same kernel was 4-5 us
// slower at that size (scalar branches and block arithmetic in every step of the three walks; profiles/r05_ab_segfb_regression.txt)
template <bool BWD, bool FUSED, bool CACHED = false>
__global__ void __launch_bounds__(64, HF_SEG_OCC) k_seg_fb(const SegDesc* __restrict__ sd, const int32_t* __restrict__ arow,
                                                           const double* __restrict__ lutA, const DevParams* __restrict__ P,
                                                           const double* __restrict__ Qs, double* Pseg, unsigned* ready, unsigned epoch, unsigned wait_epoch,
                                                           const int32_t* __restrict__ pos, double* __restrict__ recs, double* __restrict__ scale_s,
                                                           int8_t* __restrict__ label, double* __restrict__ seg_ll,
                                                           unsigned* __restrict__ flags, int32_t g0, int nc_arg,
                                                           const int32_t* __restrict__ seg_of_block) {
    static_assert(FUSED || !CACHED, "cached row blocks: one-launch mode only");
    const int nc = CACHED ? nc_arg : 0;
    constexpr int LM = HF_SEG_LMAX;
    extern __shared__ __attribute__((aligned(16))) double s_W[];
    // nc (wave-uniform, hf_create): the rows of the lane's steps 0 .. nc-1 stay in LDS blocks 0 .. nc-1 after the first walk (one-launch mode
    // only: the host passes 0 otherwise); block nc streams the other steps and is the workgroup's scratch
    double* s_rows = s_W;                                                     // block 0
    double* blk = s_W + nc * 1024;                                            // the streaming / scratch block
    uint32_t* __restrict__ s_off = reinterpret_cast<uint32_t*>(s_W + (nc + 1) * 1024);   // the row offsets of the replay
    int8_t* __restrict__ s_lab = reinterpret_cast<int8_t*>(s_W + (nc + 1) * 1024) + 64 * LM * 4;   // [64 * LM] labels of the segment
    double* s_P = blk;                                                        // prologue (two launches): the chunk's segment products, in the (still idle) row block
    double* __restrict__ s_T = reinterpret_cast<double*>(s_lab);             // phase B: the suffix products of rows 1..3 (384 of the label area's 512 bytes, idle until phase D)
    // g0: the launch's first segment (round 5: a context whose pair records exceed the Infinity Cache runs the pass in SUB-PASSES of whole
    // chunks — k_seg_fb, then k_pair_sums, per sub-pass — through one record buffer that holds a sub-pass at a time; `recs` then points
    // p0 records before the buffer, so that the plan's global positions land inside it: hf_estep.hip enqueue_pass)
    // seg_of_block (round 6, hf_create's XCD plan): the launch's block b runs segment seg_of_block[g0 + b] — all segments of a chunk on block
    // indices congruent mod 8, i.e. (observed, for speed only) on one XCD; < 0: a padding block of the plan.  Null: block b runs segment g0 + b.
    int g = (int) blockIdx.x + g0;
    const int lane = threadIdx.x;
    if (FUSED && BWD) KSTAMP(1);
    if (seg_of_block) { g = seg_of_block[g]; if (g < 0) return; }
    const SegDesc d = sd[g];
#if HF_SEG_STAGGER > 0
    // EXPERIMENT (round 6): the workgroups of a launch all start together and walk the same phases in step — the row fetches of twelve wavefronts
    // meet in the CU's one vector-memory path (64 B/clk: a forward step of all twelve takes 96 KB / 64 = 1 536 cycles, what the trace measures),
    // the scans meet in the VALU.  Later parts of the grid start HF_SEG_STAGGER cycles later each, so that one part's fetches meet another's scans.
    if (FUSED && BWD) {
        const unsigned part = (unsigned) (((unsigned long long) blockIdx.x * HF_SEG_STAGGER_PARTS) / gridDim.x);
        if (part) {
            const unsigned long long t_end = clock64() + (unsigned long long) part * HF_SEG_STAGGER;
            while (clock64() < t_end) __builtin_amdgcn_s_sleep(16);
        }
    }
#endif
    const int L = d.L, n = d.n;
    const int a = lane * L;
    const int m = n - a < L ? (n - a > 0 ? n - a : 0) : L;
    const bool chunk_first = a == 0 && d.k == 0;                           // this lane's first window starts the chunk
    unsigned bad = 0;
    double fin[4], bdir[4];
    TR_DECL;
    TR_STAMP(0);
#ifdef HF_SEG_TRACE
    if (BWD && g_seg_trace && lane == 0) g_seg_trace[(size_t) g * HF_SEG_TRACE_N + 24] = wall_clock64();
#endif
    const RowFetch F = rowfetch_init(lutA, s_off, s_rows, nc, lane);
    {
        int32_t rr[LM];
        seg_load_arows(arow + d.t0, n, L, lane, d.ident_row, rr);
        // forward vector entering the chunk: start∘e of its first window = row (0, s) of that window's row of A
        double v[4], u[4];
        {
            const double* __restrict__ A0 = lutA + (int64_t) d.chunk_slow0 * 16;
#pragma unroll
            for (int s = 0; s < 4; s++) v[s] = A0[HF_PS(0, s)];
        }
        const DevRegion* __restrict__ Rl = &P->reg[d.reg_last];
#pragma unroll
        for (int s = 0; s < 4; s++) u[s] = Rl->trans[s][4];     // the end vector (hmm.c:452-467)
        M4 Q;
        double xv[16], xs[16];     // the products of the lanes before / after this one (exclusive prefix / suffix)
        if constexpr (FUSED) {
            // ---- A (one launch): the lane product is computed here (k_seg_prod's loop); the segment's product is what the prefix
            // scan leaves in lane 63: it is PUBLISHED for the chunk's other segments, theirs are awaited (seg_gather) ----
            seg_offsets_store(rr, L, lane, s_off);
            seg_lane_product<CACHED>(F, s_rows, lane, L, chunk_first, Q);
            TR_STAMP(1);
            if (BWD) m4_park(Q, lane, blk);
            m4_scan_prefix(Q, lane);
            if (d.nseg > 1) {
                if (lane == 63) seg_xcu_store_m4(Pseg + (int64_t) g * 16, Q.m);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the product has left this CU before the flag does
                if (lane == 63) __hip_atomic_store(ready + g, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            TR_STAMP(2);
            { M4 X_; m4_dpp0<HF_DPP_WAVE_SHR1>(X_, Q); m4_dpp_or_identity_fix(X_, lane > 0);   // one DPP move per word instead of a ds_bpermute; lane 0: the identity
#pragma unroll
              for (int k = 0; k < 16; k++) xv[k] = X_.m[k]; }
            {
                const double sv = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
                for (int s = 0; s < 4; s++) v[s] /= sv;
            }
            if (BWD) v4_renorm(u);
            if (BWD) {
                // the second scan needs nothing from the other segments: it runs BEFORE their products are awaited, so that the
                // skew between the segments of a chunk is spent here instead of in the polls below (k_seg_fb 60.5 -> 59 us)
                m4_unpark(Q, lane, blk);
                seg_suffix_side(Q, lane, xs, s_T);
            }
            TR_STAMP(3);
            if (d.nseg > 1) {
                // lane l takes the product of segment base + l of the chunk (waits for its flag: bounded)
                double Tm[16];
#pragma unroll
                for (int k = 0; k < 16; k++) Tm[k] = 0.0;
                int have_base = -1;
                // the gathered products go through the (idle) row block: lane l parks segment base + l's matrix, a chain step reads
                // matrix q with one broadcast read per 16 bytes — LDS instructions instead of 32 v_readlane per step (-0.5 us)
                auto stage = [&]() {
                    M4 T;
#pragma unroll
                    for (int k = 0; k < 16; k++) T.m[k] = Tm[k];
                    __builtin_amdgcn_s_waitcnt(0xC07F);          // earlier reads of the block have returned
                    __builtin_amdgcn_wave_barrier();
                    m4_park(T, lane, blk);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                auto from_lane = [&](int src, double M[16]) {
                    const double2* __restrict__ row = reinterpret_cast<const double2*>(blk) + src * 8;
#pragma unroll
                    for (int k = 0; k < 8; k++) { const double2 t2 = row[(k + (src >> 1)) & 7]; M[2 * k] = t2.x; M[2 * k + 1] = t2.y; }
                };

                auto gather = [&](int base) {
                    if (base == have_base) return;
                    have_base = base;
                    const int q = base + lane;
                    if (q < d.nseg && q != d.k) {
                        const unsigned* r = ready + d.seg0 + q;
                        int spins = 0;
                        while (__hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != wait_epoch) {   // wait_epoch == epoch (tests: a value nobody publishes)
                            if (++spins > HF_SEG_SPIN_MAX) { bad |= HF_FLAG_SYNC; break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        seg_xcu_load_m4(Pseg + (int64_t) (d.seg0 + q) * 16, Tm);
                    }
                    stage();
                };

                for (int base = 0; base < d.k; base += 64) {                  // through the chunk's earlier segments
                    gather(base);
                    const int hi = d.k < base + 64 ? d.k : base + 64;
                    for (int q = base; q < hi; q++) { double M[16]; from_lane(q - base, M); v4_mul_right(v, M); if (false || ((q - base) & 1) || q + 1 == hi) v4_renorm(v); }
                }
                if (BWD)                                                      // ... and back through the later ones
                    for (int base = ((d.nseg - 1) >> 6) << 6; base >= 0 && base + 64 > d.k + 1; base -= 64) {
                        gather(base);
                        const int lo = d.k + 1 > base ? d.k + 1 : base;
                        for (int q = (d.nseg - 1 < base + 63 ? d.nseg - 1 : base + 63); q >= lo; q--) { double M[16]; from_lane(q - base, M); v4_mul_left(u, M); if (false || ((q - base) & 1) || q == lo) v4_renorm(u); }
                    }
            }
            TR_STAMP(4);
        } else {
        // ---- A (two launches): loads, in the order they are consumed (vmcnt counts in order): row indices, the products of the
        // chunk's other segments, the start row; LAST the lane product of k_seg_prod — the offset table and the two chains over the
        // other segments run while it is still in flight ----
        const int nst = d.nseg < HF_SEG_PSTAGE ? d.nseg : HF_SEG_PSTAGE;
        double2 pst[(HF_SEG_PSTAGE * 8 + 63) / 64];
        if (d.nseg > 1) {
            const double2* __restrict__ src = reinterpret_cast<const double2*>(Pseg + (int64_t) d.seg0 * 16);
#pragma unroll
            for (int c = 0; c < (HF_SEG_PSTAGE * 8 + 63) / 64; c++) if (c * 64 + lane < nst * 8) pst[c] = src[c * 64 + lane];
        }
        {
            const double2* __restrict__ src = reinterpret_cast<const double2*>(Qs) + (int64_t) g * 8 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 8; k++) { const double2 q2 = src[k * 64]; Q.m[2 * k] = q2.x; Q.m[2 * k + 1] = q2.y; }
        }
        seg_offsets_store(rr, L, lane, s_off);
        TR_STAMP(1);
        if (d.nseg > 1) {
#pragma unroll
            for (int c = 0; c < (HF_SEG_PSTAGE * 8 + 63) / 64; c++) if (c * 64 + lane < nst * 8) reinterpret_cast<double2*>(s_P)[c * 64 + lane] = pst[c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        {
            const double sv = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
            for (int s = 0; s < 4; s++) v[s] /= sv;
        }
        for (int q = 0; q < d.k; q++) {                          // through the chunk's earlier segments
            if (q < HF_SEG_PSTAGE) v4_mul_right(v, s_P + q * 16); else v4_mul_right(v, Pseg + (int64_t) (d.seg0 + q) * 16);
            v4_renorm(v);
        }
        TR_STAMP(2);
        if (BWD) {                                               // ... and back through the later ones
            v4_renorm(u);
            for (int q = d.nseg - 1; q > d.k; q--) {
                if (q < HF_SEG_PSTAGE) v4_mul_left(u, s_P + q * 16); else v4_mul_left(u, Pseg + (int64_t) (d.seg0 + q) * 16);
                v4_renorm(u);
            }
        }
        TR_STAMP(3);
        // ---- B: scans over the lanes of the wavefront; Q waits for the second scan in the (idle) row block ----
        if (BWD) m4_park(Q, lane, blk);
        TR_STAMP(4);
        m4_scan_prefix(Q, lane);
        { M4 X_; m4_dpp0<HF_DPP_WAVE_SHR1>(X_, Q); m4_dpp_or_identity_fix(X_, lane > 0);
#pragma unroll
          for (int k = 0; k < 16; k++) xv[k] = X_.m[k]; }
        }
        if (true) v4_mul_right(v, xv);
        {
            const double su = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
            for (int s = 0; s < 4; s++) fin[s] = v[s] / su;
        }
        if (chunk_first) { fin[0] = 1.0; fin[1] = 0.0; fin[2] = 0.0; fin[3] = 0.0; }   // (1,0,0,0)·A_first = start∘e
        TR_STAMP(5);
        if (BWD) {
            if constexpr (!FUSED) {
                m4_unpark(Q, lane, blk);
                seg_suffix_side(Q, lane, xs, s_T);
            }
            seg_suffix_apply(u, xs, lane, s_T, bdir);
            // (keep the four values HERE: left to itself the compiler sinks this arithmetic into the branch of phase D that uses it and
            // carries xs — 32 registers — across the replays, spilling the forward state instead: 195 spilled registers against 48)
#pragma unroll
            for (int s = 0; s < 4; s++) asm volatile("" : "+v"(bdir[s]));
        }
    }
    TR_STAMP(6);
    // ---- C: forward replay (hmm.c:333-434).  fs[i], ss[i]: forward vector and scale of the lane's i-th window (registers) ----
    double f[4] = {fin[0], fin[1], fin[2], fin[3]};
    double fs[LM][4], ss[LM];
    double A[16];
    // log-likelihood of the lane's windows: sum of log(scale) (hmm.c:428) as log(product of the mantissas) + (sum of the
    // exponents)·ln 2 — one log per lane instead of one (~95 instructions) per window; <= HF_SEG_LMAX mantissas in [0.5, 1)
    double lm = 1.0, scl = 1.0;
    int le = 0;
    if (nc < L) rows_issue(F, nc);                              // the first step that is not cached (nc == 0: every step is fetched again, as before round 5)
    TR_RESET();
#pragma unroll
    for (int i = 0; i < LM; i++) {
        if (i < L) {                                            // wave-uniform
            if (i >= nc) rows_wait(0);
            rows_read(F, s_rows, i, lane, A);
            TR_LAP(0);
            if (i >= nc && i + 1 < L) rows_issue(F, i + 1);     // in flight during this step
            TR_LAP(1);
            if (i < m) {
                double nf[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    double acc = f[0] * A[HF_PS(0, s)];
#pragma unroll
                    for (int p = 1; p < 4; p++) acc = fma(f[p], A[HF_PS(p, s)], acc);
                    nf[s] = acc;
                }
                const double sc = ((nf[0] + nf[1]) + nf[2]) + nf[3];
                if (!(chunk_first && i == 0) && sc < 1e-50) bad |= HF_FLAG_SCALE;   // hmm.c:412-415 (not at the chunk's first window)
                if (!(sc == sc)) bad |= HF_FLAG_NAN;                      // a NaN emission value (hmm_utils.c:783-786)
                // one division per window: f = nf * (1 / sc) differs from nf / sc in the last bit at most.  (ADVICE r04 asked what a SUBNORMAL scale
                // does to the reciprocal: it cannot occur.  Only a chunk's first window is exempt from the 1e-50 test, and there the scale is a
                // sum of start x emission values of which the Gaussian states' carry the reference's 1e-40 floor per component
                // (hmm_utils.c:787-790): >= 0.25e-40.  The negative-binomial tables have no floor, but a zero row gives 0 / 0 in the reference too.
                // A guard here — even in the unrolled first step only — cost the kernel 1-2 us: profiles/r05_ab_segfb_regression.txt.)
                const double rsc = seg_recip(sc);
#pragma unroll
                for (int s = 0; s < 4; s++) { f[s] = nf[s] * rsc; fs[i][s] = f[s]; }
                { int e2; lm *= frexp(sc, &e2); le += e2; }                // hmm.c:428, see above
                scl = sc; ss[i] = sc;
            }
            TR_LAP(2);
        }
    }
    TR_STAMP(7);
    {
        double ll = log(lm) + (double) le * 0.693147180559945309417232121458;
        // sum over the 64 lanes without LDS traffic: inclusive row scan (lanes without a source add 0), then the row totals
        ll += dpp0_f64<HF_DPP_ROW_SHR(1), 0xf>(ll); ll += dpp0_f64<HF_DPP_ROW_SHR(2), 0xf>(ll);
        ll += dpp0_f64<HF_DPP_ROW_SHR(4), 0xf>(ll); ll += dpp0_f64<HF_DPP_ROW_SHR(8), 0xf>(ll);
        ll += dpp0_f64<HF_DPP_ROW_BCAST15, 0xa>(ll); ll += dpp0_f64<HF_DPP_ROW_BCAST31, 0xc>(ll);
        if (lane == 63) seg_ll[g] = ll;
    }
    TR_STAMP(8);
    // ---- D: backward replay + labels (hmm.c:452-545, 671-692) ----
    if (BWD) {
        const int jl = m - 1;                                   // the lane's last window (< 0: none)
        const DevRegion* __restrict__ Rl = &P->reg[d.reg_last];
        const int64_t slot_ij = (int64_t) d.slot0 + lane;       // + i*64
        // f of the window before the lane's first one: the previous lane's last forward vector, the carried-in vector for lane 0
        double fp[4];
#pragma unroll
        for (int s = 0; s < 4; s++) { fp[s] = __shfl_up(f[s], 1); if (lane == 0) fp[s] = fin[s]; }
        double b[4] = {0.0, 0.0, 0.0, 0.0};
        if (jl >= 0) {
            if (d.k == d.nseg - 1 && a + jl == n - 1) {         // the chunk's last window, hmm.c:452-467
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = Rl->trans[s][4] / scl;
                double2* __restrict__ dst = reinterpret_cast<double2*>(recs) + (int64_t) d.spare_pos * 4;   // its f: the chunk's spare record
                dst[0] = make_double2(f[0], f[1]); dst[1] = make_double2(f[2], f[3]);
            } else {                                            // direction from the scans, magnitude from the invariant at this window
                const double term = Rl->trans[0][4];
                double dot = 0.0;
#pragma unroll
                for (int s = 0; s < 4; s++) dot += f[s] * bdir[s];
                const double kk = term / (scl * dot);
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = bdir[s] * kk;
            }
            s_lab[a + jl] = (int8_t) posterior_label_fast(f, b, scl);
        }
        // record k = { f_{k-1}, b_k } and scale k leave at the top of the step that consumes b_k; window k's row turns b_k into
        // b_{k-1}.  All lanes run k = L-1 .. 1 (cooperative fetch), a lane joins at its last window.  The first step still
        // holds the row of window L-1 from the forward replay (a lane that joins there has m = L).
        // Record k = { f_{k-1}, b_k } goes to the POSITION hf_create planned for the window (the statistics then stream the
        // records of a row of A; positions are scattered).  Through the (momentarily idle) row block, so that four adjacent lanes
        // write one WHOLE 64-byte record: written lane by lane as 16-byte pieces, the same bytes cost 13 us more per launch
        // (measured: partial-line writes, profiles/r03c_ablation.txt).  Lanes without a window k write nothing.
        const int32_t* __restrict__ pos_seg = pos + d.t0;
        auto store_rec = [&](int k, bool act, int32_t pk, const double* __restrict__ fk, double sck) {
            // Round 6 (HF_SEG_RECPAD, default on): the staged records lie 80 bytes apart instead of 64 — at a 64-byte stride the sixteen lanes of a
            // quarter-wavefront hit four groups of four banks (a 16-way conflict on a quarter of the LDS: 1 024 conflict cycles per wavefront and
            // launch, 32 % of the LDS cycles: profiles/r05h_pmc_sq_b.json); 20 dwords apart they cover all 64 banks exactly once, on the write and
            // on the read side.  No extra instruction: only the address constants change (round 4 had ROTATED the pieces instead — selects per
            // register — and lost 0.3 us).  5 120 of the row block's 8 192 bytes.  HF_SEG_RECPAD=2: 64 bytes apart, piece j of record r in slot
            // j ^ ((r >> 2) & 3) — conflict-free on both sides, and 0.8 us slower than the padding (profiles/r06_ab_swizzle.txt): not the default.
#ifndef HF_SEG_RECPAD
#define HF_SEG_RECPAD 1
#endif
            constexpr int RS = HF_SEG_RECPAD == 1 ? 5 : 4;      // double2 per staged record
            constexpr bool SWZ = HF_SEG_RECPAD == 2;            // (2: 64 bytes apart, piece j of record r at slot j ^ ((r >> 2) & 3))
            const int wsz = SWZ ? ((lane >> 2) & 3) : 0;
            double2* __restrict__ mine = reinterpret_cast<double2*>(blk) + lane * RS;
            // (rotating the pieces so that these writes are free of bank conflicts — lane by lane at a 64-byte stride they hit four banks of
            // 64 — was measured 0.3 us SLOWER, profiles/r04e_ab_variants.txt: the conflicts are not on the kernel's critical path)
            // (a lane without a window k passes forward values it never computed — they go to the segment's spare record, which nobody reads.
            // They pass through an empty asm where they are used: to the compiler a value the asm defines — whatever the register holds,
            // but nothing indeterminate to exploit — at no instruction; zero-filling the forty registers ahead of the replays cost
            // the kernel ~1 us, profiles/r05_ab_segfb_regression.txt)
            double f0 = fk[0], f1 = fk[1], f2 = fk[2], f3 = fk[3], sc_ = sck;
            asm("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(sc_));
            mine[0 ^ wsz] = make_double2(f0, f1); mine[1 ^ wsz] = make_double2(f2, f3);
            mine[2 ^ wsz] = make_double2(b[0], b[1]); mine[3 ^ wsz] = make_double2(b[2], b[3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // store instruction q writes records 16 q .. 16 q + 15: lane l holds piece l & 3 of record 16 q + (l >> 2)
            const double2* __restrict__ img = reinterpret_cast<const double2*>(blk) + (lane >> 2) * RS + ((lane & 3) ^ (SWZ ? ((lane >> 4) & 3) : 0));
            const double2 v0 = img[0], v1 = img[16 * RS], v2 = img[32 * RS], v3 = img[48 * RS];
            double2* __restrict__ R2 = reinterpret_cast<double2*>(recs) + (lane & 3);
            // lanes without a window k write THE SEGMENT'S SPARE RECORD (SegDesc.trash_pos, behind the sub-pass's positions) and a padding slot
            // of the scales (a segment owns 64 L slots): straight-line stores instead of five regions of masked execution per step
            const int32_t pact = act ? pk : d.trash_pos;    // the record of lane r goes to position pact(r): lanes 4r .. 4r+3 of instruction r >> 4
            const int32_t p0 = __shfl(pact, lane >> 2), p1 = __shfl(pact, 16 + (lane >> 2)), p2 = __shfl(pact, 32 + (lane >> 2)), p3 = __shfl(pact, 48 + (lane >> 2));
            R2[(int64_t) p0 * 4] = v0;
            R2[(int64_t) p1 * 4] = v1;
            R2[(int64_t) p2 * 4] = v2;
            R2[(int64_t) p3 * 4] = v3;
            if (scale_s) scale_s[slot_ij + (int64_t) k * 64] = sc_;   // (wave-uniform: an EM pass passes no scale array — only the host getters read it, and they run the kernel again with one; non-temporal stores: k_pair_sums +3.5 us, it reads these records out of the cache; profiles/r04i_ab_variants.txt)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                    // the block has been read out: the next row fetch may land
        };
        int32_t pk_next = jl >= 0 ? pos_seg[a + jl] : 0;        // position of the record of the lane's last window (the first step's, when that is window L-1)
        TR_STAMP(9);
        TR_RESET();
#pragma unroll
        for (int k = LM - 1; k >= 1; k--) {
            if (k < L) {                                        // wave-uniform
                if (k < L - 1) {
                    if (k >= nc) rows_wait(0);                  // (vmcnt(0): the record stores too, as before; a cached step waits for nothing)
                    rows_read(F, s_rows, k, lane, A);
                }
                TR_LAP(3);
                const bool act = k <= jl;                       // this lane has a window k
                const int32_t pk = pk_next;
                pk_next = (k - 1 <= jl && k - 1 >= 0) ? pos_seg[a + k - 1] : 0;   // in flight during this step
                store_rec(k, act, pk, fs[k - 1], ss[k]);
                TR_LAP(4);
                if (k >= 2 && k - 1 >= nc) rows_issue(F, k - 1);
                TR_LAP(5);
                if (act) {
                    double nb[4];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        double acc = A[HF_PS(p, 0)] * b[0];
#pragma unroll
                        for (int s = 1; s < 4; s++) acc = fma(A[HF_PS(p, s)], b[s], acc);
                        nb[p] = acc;
                    }
                    const double sc = ss[k - 1];
                    if (sc < 1e-50) bad |= HF_FLAG_SCALE;                     // hmm.c:521-524
                    const double rsc = seg_recip(sc);
#pragma unroll
                    for (int s = 0; s < 4; s++) b[s] = nb[s] * rsc;
                    s_lab[a + k - 1] = (int8_t) posterior_label_fast(fs[k - 1], b, sc);
                }
                TR_LAP(6);
            }
        }
        TR_STAMP(10);
        store_rec(0, jl >= 0, pk_next, fp, ss[0]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int8_t* __restrict__ dst = label + d.t0;
        for (int w = lane; w < n; w += 64) dst[w] = s_lab[w];
    }
    TR_STAMP(11);
    TR_FLUSH(12, 7);
#ifdef HF_SEG_TRACE
    if (BWD && g_seg_trace && lane == 0) {
        g_seg_trace[(size_t) g * HF_SEG_TRACE_N + 19] = (unsigned long long) L;
        g_seg_trace[(size_t) g * HF_SEG_TRACE_N + 20] = (unsigned long long) n;
        g_seg_trace[(size_t) g * HF_SEG_TRACE_N + 21] = wall_clock64();
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_seg_trace[(size_t) g * HF_SEG_TRACE_N + 22] = hw; g_seg_trace[(size_t) g * HF_SEG_TRACE_N + 23] = xcc;
    }
#endif
    if (bad) atomicOr(flags, bad);
}
