// hf_seg.h — HF_ALGO_SCAN: one WORKGROUP per chunk segment, the whole forward / backward / decode of the segment in two
// launches (BASELINE north_star: "one contig-chunk per workgroup ... wavefront prefix-scan for the forward/backward
// recurrences").
//
// A chunk of T windows (hmm.c:333-545 runs it strictly sequentially) is cut into n = ceil(T / HF_SEG_SPLIT) equal SEGMENTS of
// at most NL*LMAX windows (NL = 64*NW lanes per workgroup; NW = 1 by default: a CU is busy for the sum of its workgroups'
// steps, and one-wavefront workgroups spread most evenly).  Inside a segment lane j owns the L = ceil(n_windows / NL)
// consecutive windows j*L .. j*L+L-1, in both directions:
//   A  (k_seg_prod) lane product Q_j = A_{jL} ... A_{jL+L-1}, A_t = T_t∘e_t: ONE precomputed 128-byte row per window (below);
//      the product of the whole segment for the chunk's other segments;
//   B  (k_seg_fb) prefix and suffix scans of Q over the 64 lanes of a wavefront (DPP row shifts / broadcasts, no LDS traffic
//      inside a row of 16 lanes), the wave totals through LDS when NW > 1, and the products of the chunk's other segments:
//      every lane gets the normalised forward vector entering its first window and the direction of b at its last one;
//   C  forward REPLAY of the lane's windows (f·A, pre-inner sums, division by the scale, log: hmm.c:366-434): the carried-in
//      vector differs from a sequential run in the last ulp, and so may a term f·(T·e) from the reference's (f·T)·e;
//   D  backward replay + posterior argmax (hmm.c:470-529, 671-692); the magnitude of the carried-in b from the invariant
//      sum_s f_t[s]·b_t[s]·scale_t = terminationProb of the scaled forward-backward.
//
// Rows: every lane needs ITS OWN 128-byte row per window.  A lane reading its row with eight 16-byte loads touches
// 64 different cache lines per instruction and depends on the 32 KiB L1 keeping each line for the seven loads that follow —
// it does not (measured: ~1 000 cycles per wavefront and row).  Here the 64 rows of a step are fetched COOPERATIVELY with
// LDS-DMA (global_load_lds_dwordx4: instruction q moves rows 8q..8q+7 complete, 8 lanes x 16 bytes each, straight into the
// wavefront's 8 KiB LDS block — no staging registers) and every lane then reads its row with eight conflict-free
// ds_read_b128; the piece rotation that makes the reads conflict-free is applied on the SOURCE side of the DMA.
//
// Output = the PAIR RECORDS the statistics read (hf_rows.h by emission row, hf_chunks.h per chunk): record(t) = { f_{t-1}[4],
// b_t[4] }, 64 bytes, and the scales — both in SLOT order: window w of a segment (w = j*L + i) lives in slot
// slot0 + i*NL + j, so that at every step the lanes of a wavefront write 64 consecutive records (the statistics address
// records by slot; the host getters apply the same map).  Labels leave through LDS, coalesced.
#pragma once
#ifndef HF_SEG_R2_GUARD
#define HF_SEG_R2_GUARD
#endif
#include "hf_scan.h"

#ifndef HF_SEG_WAVES
#define HF_SEG_WAVES 1      // wavefronts per workgroup (1: the finest load balance over the CUs; measured best or equal from 0.2 M to 6 M windows)
#endif
#ifndef HF_SEG_LMAX
#define HF_SEG_LMAX 8       // windows per lane at most: a chunk longer than 64*HF_SEG_WAVES*HF_SEG_LMAX windows is split
#endif
#ifndef HF_SEG_SPLIT
#define HF_SEG_SPLIT (64 * HF_SEG_WAVES * HF_SEG_LMAX)   // windows per segment a chunk is cut by (equal parts of at most this)
#endif
#ifndef HF_SEG_OCC
#define HF_SEG_OCC 4        // wavefronts per SIMD the register allocation aims at
#endif

// SegDesc: hf_device.h

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

// power-of-two renormalisation with a tree maximum (the chain of m4_renorm is 15 dependent operations)
__device__ __forceinline__ void m4_renorm_tree(M4& a) {
    double t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = fmax(a.m[i], a.m[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = fmax(t[i], t[i + 4]);
    const double mx = fmax(fmax(t[0], t[1]), fmax(t[2], t[3]));
    int e;
    (void) frexp(mx, &e);
    if (mx > 0.0) {
#pragma unroll
        for (int i = 0; i < 16; i++) a.m[i] = ldexp(a.m[i], -e);
    }
}

// ---- DPP moves of a 4x4 matrix (gfx9 row_shr / row_shl / row_bcast): lanes without a source keep their own value ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
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

// inclusive prefix product over the 64 lanes: lane l ends with Q_0 ... Q_l (power-of-two renormalised: exact)
__device__ __forceinline__ void m4_scan_prefix(M4& Pq, int lane) {
    M4 Lft, R;
#define HF_STEP_SHR(n)                                                                                  \
    m4_dpp<HF_DPP_ROW_SHR(n)>(Lft, Pq);                                                                  \
    if ((lane & 15) >= (n)) { m4_mul(R, Lft, Pq); Pq = R; m4_renorm_tree(Pq); }
    HF_STEP_SHR(1) HF_STEP_SHR(2) HF_STEP_SHR(4) HF_STEP_SHR(8)
#undef HF_STEP_SHR
    m4_dpp<HF_DPP_ROW_BCAST15, 0xa>(Lft, Pq);                 // rows 1, 3 <- lane 15 of rows 0, 2
    if (lane & 16) { m4_mul(R, Lft, Pq); Pq = R; m4_renorm_tree(Pq); }
    m4_dpp<HF_DPP_ROW_BCAST31, 0xc>(Lft, Pq);                 // rows 2, 3 <- lane 31
    if (lane >= 32) { m4_mul(R, Lft, Pq); Pq = R; m4_renorm_tree(Pq); }
}
// inclusive suffix product: lane l ends with Q_l ... Q_63
__device__ __forceinline__ void m4_scan_suffix(M4& Sq, int lane) {
    M4 Rgt, R;
#define HF_STEP_SHL(n)                                                                                  \
    m4_dpp<HF_DPP_ROW_SHL(n)>(Rgt, Sq);                                                                  \
    if ((lane & 15) + (n) < 16) { m4_mul(R, Sq, Rgt); Sq = R; m4_renorm_tree(Sq); }
    HF_STEP_SHL(1) HF_STEP_SHL(2) HF_STEP_SHL(4) HF_STEP_SHL(8)
#undef HF_STEP_SHL
#pragma unroll
    for (int i = 0; i < 16; i++) Rgt.m[i] = __shfl(Sq.m[i], (lane | 15) + 1);   // first lane of the next row
    if (!(lane & 16)) { m4_mul(R, Sq, Rgt); Sq = R; m4_renorm_tree(Sq); }       // rows 0, 2
#pragma unroll
    for (int i = 0; i < 16; i++) Rgt.m[i] = __shfl(Sq.m[i], 32);
    if (lane < 32) { m4_mul(R, Sq, Rgt); Sq = R; m4_renorm_tree(Sq); }
}

// ---- cooperative row fetch through the wavefront's 8 KiB LDS block (see the header) ----
// Row r of the step (the row lane r needs) occupies bytes r*128 .. r*128+127 of the block; piece p of the row sits in slot
// (p + (r >> 1)) & 7: the eight ds_read_b128 of a lane group then cover all 64 banks exactly once.
__device__ __forceinline__ void rows_issue(const double* __restrict__ rows, int32_t ridx, int lane, double* __restrict__ blk) {
    // lane (sub, part) of instruction q moves 16 bytes of row r = 8q + sub: piece (part - (r >> 1)) & 7 = (c0 - 4q) & 7, i.e.
    // one of two values; 32-bit byte offsets from the (wave-uniform) table base
    const int part = lane & 7, sub = lane >> 3;
    const uint32_t c0 = (uint32_t) (part - (sub >> 1));
    const uint32_t off_even = (c0 & 7u) << 4, off_odd = ((c0 + 4u) & 7u) << 4;
    const char* __restrict__ base = reinterpret_cast<const char*>(rows);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t idx = (uint32_t) __shfl(ridx, q * 8 + sub);
        const uint32_t off = (idx << 7) + ((q & 1) ? off_odd : off_even);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (base + off),
                                         (__attribute__((address_space(3))) void*) (blk + q * 128), 16, 0, 0);
    }
}
__device__ __forceinline__ void rows_read(const double* __restrict__ blk, int lane, double E[16]) {
    __builtin_amdgcn_s_waitcnt(0);          // the wavefront's own LDS-DMA has landed (vmcnt) — nothing else orders it
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double2* __restrict__ row = reinterpret_cast<const double2*>(blk) + lane * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 d = row[(k + (lane >> 1)) & 7]; E[2 * k] = d.x; E[2 * k + 1] = d.y; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);          // the reads have returned before the block is refilled
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
// k_tables_nb — by k_arows), and the segment kernels fetch ONE 128-byte row per window and step: no transition-table lookup, no second factor, no region tables in LDS, and the row of a window is a
// precomputed index (hf_create: d_arow, bit 31 = chunk-first) instead of a function of two records and a slow-list rank.
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

#define HF_AROW_ID(r) ((r) & 0x7fffffff)

// the lane's product of A_t over its m windows (a chunk-first window is left out: it belongs to the start vector).  All 64
// lanes run all L steps (the row fetch is cooperative); lanes past their last window fetch row 0 and skip the arithmetic.
__device__ __forceinline__ void seg_lane_product(const int32_t* __restrict__ arow_seg, int a, int m, int L,
                                                 const double* __restrict__ lutA, double* __restrict__ blk, int lane, M4& Q) {
    m4_identity(Q);
    int32_t r = m > 0 ? arow_seg[a] : 0;
    int32_t r1 = m > 1 ? arow_seg[a + 1] : 0;              // row indices are fetched two steps ahead
    rows_issue(lutA, HF_AROW_ID(r), lane, blk);
#pragma unroll 1
    for (int i = 0; i < L; i++) {
        double E[16];
        rows_read(blk, lane, E);
        const int32_t rn = r1;
        if (i + 1 < L) rows_issue(lutA, HF_AROW_ID(rn), lane, blk);   // in flight during this step
        r1 = i + 2 < m ? arow_seg[a + i + 2] : 0;
        if (i < m && r >= 0) {
            M4 A, R;
#pragma unroll
            for (int k = 0; k < 16; k++) A.m[k] = E[HF_PS(k >> 2, k & 3)];
            m4_mul(R, Q, A);
            Q = R;
            m4_renorm_tree(Q);
        }
        r = rn;
    }
}

// bytes of dynamic LDS of the segment kernels: wave totals | ll partials (+ padding) | labels | row blocks
template <int NW>
__host__ __device__ constexpr size_t seg_lds_bytes() {
    return (NW * 16 + 2 * NW) * 8 + (size_t) 64 * NW * HF_SEG_LMAX + (size_t) NW * 8192;
}

// ------------------------------------------------------------------------------------------
// k_seg_prod: phase A for every segment: the lane products (lane-minor: 1 KiB per store instruction; k_seg_fb's scans start
// from them) and the product of the whole segment (used by the chunk's OTHER segments only).
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NW * 64, HF_SEG_OCC) k_seg_prod(const SegDesc* __restrict__ sd, const int32_t* __restrict__ arow,
                                                                   const double* __restrict__ lutA,
                                                                   double* __restrict__ Qs, double* __restrict__ Pseg) {
    constexpr int NL = NW * 64;
    extern __shared__ __attribute__((aligned(16))) double s_W[];
    const int g = blockIdx.x;
    const SegDesc d = sd[g];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = wave * 64 + lane;
    double* __restrict__ blk = s_W + NW * 16 + 2 * NW + (64 * NW * HF_SEG_LMAX) / 8 + wave * 1024;
    const int a = j * d.L;
    const int m = d.n - a < d.L ? (d.n - a > 0 ? d.n - a : 0) : d.L;
    M4 Q;
    seg_lane_product(arow + d.t0, a, m, d.L, lutA, blk, lane, Q);
    {
        double2* __restrict__ dst = reinterpret_cast<double2*>(Qs) + (int64_t) g * 8 * NL + j;
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k * NL] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    if (d.nseg == 1) return;                                  // nobody reads the product of a one-segment chunk
    m4_scan_prefix(Q, lane);                                  // lane 63: the product of the wavefront
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 16; k++) s_W[wave * 16 + k] = Q.m[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) Q.m[k] = s_W[k];
        for (int w = 1; w < NW; w++) {
            M4 B, R;
#pragma unroll
            for (int k = 0; k < 16; k++) B.m[k] = s_W[w * 16 + k];
            m4_mul(R, Q, B);
            Q = R;
            m4_renorm_tree(Q);
        }
        double2* dst = reinterpret_cast<double2*>(Pseg + (int64_t) g * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
}

// ------------------------------------------------------------------------------------------
// k_seg_fb: one workgroup per segment: phases B-D of the header.  BWD = false: forward only (EM_runForwardForList,
// hmm.c:790-816): log-likelihood and error flags, nothing else is written.
// ------------------------------------------------------------------------------------------
template <int NW, bool BWD>
__global__ void __launch_bounds__(NW * 64, HF_SEG_OCC) k_seg_fb(const SegDesc* __restrict__ sd, const int32_t* __restrict__ arow,
                                                                 const double* __restrict__ lutA, const DevParams* __restrict__ P,
                                                                 const double* __restrict__ Qs, const double* __restrict__ Pseg,
                                                                 double* __restrict__ recs,
                                                                 double* __restrict__ scale_s, int8_t* __restrict__ label,
                                                                 double* __restrict__ seg_ll, unsigned* __restrict__ flags) {
    constexpr int NL = NW * 64;
    extern __shared__ __attribute__((aligned(16))) double s_W[];           // [NW][16] wave totals
    double* __restrict__ s_red = s_W + NW * 16;                           // [NW] log-likelihood partials (+ NW of padding)
    int8_t* __restrict__ s_lab = reinterpret_cast<int8_t*>(s_red + 2 * NW);   // [NL * LMAX] labels of the segment
    const int g = blockIdx.x;
    const SegDesc d = sd[g];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = wave * 64 + lane;
    double* __restrict__ blk = s_red + 2 * NW + (64 * NW * HF_SEG_LMAX) / 8 + wave * 1024;   // this wavefront's 8 KiB row block
    const int L = d.L, n = d.n;
    const int a = j * L;
    const int m = n - a < L ? (n - a > 0 ? n - a : 0) : L;
    const int32_t* __restrict__ arow_seg = arow + d.t0;
    const bool chunk_first = a == 0 && d.k == 0;                           // this lane's first window starts the chunk
    unsigned bad = 0;
    double fin[4], bdir[4];
    {
        // ---- A: the lane product, from k_seg_prod ----
        M4 Q;
        {
            const double2* __restrict__ src = reinterpret_cast<const double2*>(Qs) + (int64_t) g * 8 * NL + j;
#pragma unroll
            for (int k = 0; k < 8; k++) { const double2 v = src[k * NL]; Q.m[2 * k] = v.x; Q.m[2 * k + 1] = v.y; }
        }
        // ---- B: scans over the lanes of the wavefront; Q waits for the second scan in the (idle) row block ----
        if (BWD) m4_park(Q, lane, blk);
        m4_scan_prefix(Q, lane);
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 16; k++) s_W[wave * 16 + k] = Q.m[k];
        }
        double xv[16];                                          // exclusive prefix: the product of lanes 0..lane-1
#pragma unroll
        for (int k = 0; k < 16; k++) xv[k] = __shfl_up(Q.m[k], 1);
        __syncthreads();
        // forward vector entering the segment: start∘e of the chunk's first window (row (0, s) of its row of A), through the
        // products of the chunk's earlier segments and of the earlier wavefronts
        double v[4];
        {
            const double* __restrict__ A0 = lutA + (int64_t) d.chunk_slow0 * 16;
            double sv = 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) { v[s] = A0[HF_PS(0, s)]; sv += v[s]; }
#pragma unroll
            for (int s = 0; s < 4; s++) v[s] /= sv;
        }
        for (int q = 0; q < d.k; q++) { v4_mul_right(v, Pseg + (int64_t) (d.seg0 + q) * 16); v4_renorm(v); }
        for (int w = 0; w < wave; w++) { v4_mul_right(v, s_W + w * 16); v4_renorm(v); }
        if (lane > 0) v4_mul_right(v, xv);
        {
            const double su = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
            for (int s = 0; s < 4; s++) fin[s] = v[s] / su;
        }
        if (chunk_first) { fin[0] = 1.0; fin[1] = 0.0; fin[2] = 0.0; fin[3] = 0.0; }   // (1,0,0,0)·A_first = start∘e
        if (BWD) {
            m4_unpark(Q, lane, blk);
            m4_scan_suffix(Q, lane);
#pragma unroll
            for (int k = 0; k < 16; k++) xv[k] = __shfl_down(Q.m[k], 1);   // exclusive suffix: lanes lane+1..63
            // direction of b at the lane's last window: everything after it applied to the end vector (hmm.c:452-467)
            double u[4];
            const DevRegion* __restrict__ Rl = &P->reg[d.reg_last];
#pragma unroll
            for (int s = 0; s < 4; s++) u[s] = Rl->trans[s][4];
            v4_renorm(u);
            for (int q = d.nseg - 1; q > d.k; q--) { v4_mul_left(u, Pseg + (int64_t) (d.seg0 + q) * 16); v4_renorm(u); }
            for (int w = NW - 1; w > wave; w--) { v4_mul_left(u, s_W + w * 16); v4_renorm(u); }
#pragma unroll
            for (int s = 0; s < 4; s++) bdir[s] = u[s];
            if (lane < 63) v4_mul_left(bdir, xv);
        }
    }
    // ---- C: forward replay (hmm.c:333-434) ----
    double f[4] = {fin[0], fin[1], fin[2], fin[3]};
    // log-likelihood of the lane's windows: sum of log(scale) (hmm.c:428) as log(product of the mantissas) + (sum of the
    // exponents)·ln 2 — one log per lane instead of one (~95 instructions) per window; <= HF_SEG_LMAX mantissas in [0.5, 1)
    double lm = 1.0, scl = 1.0;
    int le = 0;
    const int64_t slot_ij = (int64_t) d.slot0 + j;                         // + i*NL
    // The outputs of step i (scale, f) are STORED at the top of step i+1, after the wait for that step's rows: a store issued
    // right before the wait would make every step pay the full store latency (vmcnt counts loads and stores alike).
    auto store_fwd = [&](int i) {
        scale_s[slot_ij + (int64_t) i * NL] = scl;
        // f_t is the first half of record t+1: the lane's next slot, the next lane's first slot, or the next segment's
        int64_t sf = i + 1 < L ? slot_ij + (int64_t) (i + 1) * NL : slot_ij + 1;
        if (a + i + 1 == n) sf = d.next_slot;
        double2* __restrict__ dst = reinterpret_cast<double2*>(recs) + sf * 4;
        dst[0] = make_double2(f[0], f[1]); dst[1] = make_double2(f[2], f[3]);
    };
    {
        int32_t r = m > 0 ? arow_seg[a] : 0;
        int32_t r1 = m > 1 ? arow_seg[a + 1] : 0;                          // row indices are fetched two steps ahead
        rows_issue(lutA, HF_AROW_ID(r), lane, blk);
#pragma unroll 1
        for (int i = 0; i < L; i++) {
            double A[16];
            rows_read(blk, lane, A);
            if (BWD && i >= 1 && i - 1 < m) store_fwd(i - 1);
            const int32_t rn = r1;
            if (i + 1 < L) rows_issue(lutA, HF_AROW_ID(rn), lane, blk);
            r1 = i + 2 < m ? arow_seg[a + i + 2] : 0;
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
                if (r >= 0 && sc < 1e-50) bad |= HF_FLAG_SCALE;           // hmm.c:412-415 (not at the chunk's first window)
                if (!(sc == sc)) bad |= HF_FLAG_NAN;                      // a NaN emission value (hmm_utils.c:783-786)
#pragma unroll
                for (int s = 0; s < 4; s++) f[s] = nf[s] / sc;
                { int e2; lm *= frexp(sc, &e2); le += e2; }                // hmm.c:428, see above
                scl = sc;
            }
            r = rn;
        }
        if (BWD && L - 1 < m) store_fwd(L - 1);
    }
    double ll = log(lm) + (double) le * 0.693147180559945309417232121458;
    for (int o = 32; o > 0; o >>= 1) ll += __shfl_down(ll, o);
    if (lane == 0) s_red[wave] = ll;
    // ---- D: backward replay + labels (hmm.c:452-545, 671-692) ----
    if (BWD) {
        const int jl = m - 1;                                   // the lane's last window (< 0: none)
        const DevRegion* __restrict__ Rl = &P->reg[d.reg_last];
        double b[4] = {0.0, 0.0, 0.0, 0.0};
        if (jl >= 0) {
            if (d.k == d.nseg - 1 && a + jl == n - 1) {         // the chunk's last window, hmm.c:452-467
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = Rl->trans[s][4] / scl;
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
        // b_k is stored at the top of the step that consumes it (see store_fwd); scale and f of a step are loaded one step ahead
        auto store_bwd = [&](int k) {
            double2* __restrict__ dst = reinterpret_cast<double2*>(recs) + (slot_ij + (int64_t) k * NL) * 4 + 2;
            dst[0] = make_double2(b[0], b[1]); dst[1] = make_double2(b[2], b[3]);
        };
        // window k's row turns b_k into b_{k-1}; all lanes run k = L-1 .. 1 (cooperative fetch), a lane joins at its last window
        auto row_of = [&](int k) { return (k >= 1 && k <= jl) ? HF_AROW_ID(arow_seg[a + k]) : 0; };
        int32_t rkm1 = row_of(L - 2);                           // the row of the NEXT step, fetched one step ahead
        rows_issue(lutA, row_of(L - 1), lane, blk);
        double nsc = 1.0;
        double2 nf01 = make_double2(0.0, 0.0), nf23 = nf01;
        if (jl >= 1 && jl == L - 1) {
            nsc = scale_s[slot_ij + (int64_t) (jl - 1) * NL];
            const double2* __restrict__ fsrc = reinterpret_cast<const double2*>(recs) + (slot_ij + (int64_t) jl * NL) * 4;
            nf01 = fsrc[0]; nf23 = fsrc[1];
        }
#pragma unroll 1
        for (int k = L - 1; k >= 1; k--) {
            double A[16];
            rows_read(blk, lane, A);
            const bool act = k <= jl;                           // this lane has a window k
            if (act) store_bwd(k);
            const double sc = nsc;
            const double2 f01 = nf01, f23 = nf23;
            if (k >= 2) {
                rows_issue(lutA, rkm1, lane, blk);
                rkm1 = row_of(k - 2);
                if (k - 1 <= jl) {                              // scale and f of step k-1, in flight during this step
                    nsc = scale_s[slot_ij + (int64_t) (k - 2) * NL];
                    const double2* __restrict__ fsrc = reinterpret_cast<const double2*>(recs) + (slot_ij + (int64_t) (k - 1) * NL) * 4;
                    nf01 = fsrc[0]; nf23 = fsrc[1];
                }
            }
            if (act) {
                double nb[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    double acc = A[HF_PS(p, 0)] * b[0];
#pragma unroll
                    for (int s = 1; s < 4; s++) acc = fma(A[HF_PS(p, s)], b[s], acc);
                    nb[p] = acc;
                }
                if (sc < 1e-50) bad |= HF_FLAG_SCALE;                     // hmm.c:521-524
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = nb[s] / sc;
                const double fi[4] = {f01.x, f01.y, f23.x, f23.y};
                s_lab[a + k - 1] = (int8_t) posterior_label_fast(fi, b, sc);
            }
        }
        if (jl >= 0) store_bwd(0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; w++) s += s_red[w];
        seg_ll[g] = s;
    }
    if (BWD) {
        int8_t* __restrict__ dst = label + d.t0;
        for (int w = threadIdx.x; w < n; w += NL) dst[w] = s_lab[w];
    }
    if (bad) atomicOr(flags, bad);
}
