// hf_seg.h — HF_ALGO_SCAN, second generation: one WORKGROUP per chunk segment, the whole forward / backward / decode of
// the segment in one kernel (BASELINE north_star: "one contig-chunk per workgroup ... wavefront prefix-scan for the
// forward/backward recurrences").
//
// A chunk of T windows (hmm.c:333-545 runs it strictly sequentially) is cut into n = ceil(T / (NL*LMAX)) equal SEGMENTS
// (NL = 64*NW lanes per workgroup); chunks of up to NL*LMAX windows are ONE segment.  Inside a segment lane j owns the
// L = ceil(n_windows / NL) consecutive windows j*L .. j*L+L-1, in both directions:
//   A  lane product Q_j = A_{jL} ... A_{jL+L-1}, A_t = T_t∘E_t (rows gathered from this iteration's tables, k_tables);
//   B  Kogge-Stone prefix and suffix scans of Q over the 64 lanes of a wavefront, the NW wave totals through LDS, and —
//      chunks of several segments only — the products of the chunk's other segments (k_seg_prod, a separate, cheap launch):
//      every lane gets the normalised forward vector entering its first window and the direction of b at its last one;
//   C  forward REPLAY of the lane's windows in the reference's exact operation order ((f·T)·e, pre-inner sums, division by
//      the scale, log): only the carried-in vector differs from a sequential run, in the last ulp;
//   D  backward replay + posterior argmax (hmm.c:470-529, 671-692); the magnitude of the carried-in b from the invariant
//      sum_s f_t[s]·b_t[s]·scale_t = terminationProb of the scaled forward-backward.
// No lane product, tile product or carry vector ever goes through HBM, and nothing is computed twice for one-segment chunks.
//
// Output = the PAIR RECORDS the statistics by emission row read (hf_rows.h): record(t) = { f_{t-1}[4], b_t[4] }, 64 bytes,
// and the scales — both in SLOT order: window w of a segment (w = j*L + i) lives in slot slot0 + i*NL + j, so that at every
// step the lanes of a wavefront write 64 consecutive records (the statistics plan addresses records by slot; the host
// getters apply the same map).  Labels leave through LDS, coalesced.
#pragma once
#include "hf_scan.h"

#ifndef HF_SEG_WAVES
#define HF_SEG_WAVES 8      // wavefronts per workgroup
#endif
#ifndef HF_SEG_LMAX
#define HF_SEG_LMAX 8       // windows per lane at most: a chunk longer than 64*HF_SEG_WAVES*HF_SEG_LMAX windows is split
#endif

struct SegDesc {
    long long t0;            // global index of the segment's first window
    int n, L;                // windows of the segment, windows per lane
    int slot0, next_slot;    // first record slot; the slot whose f half takes f of the segment's LAST window
    int slow0;               // slow-list position of the first slow window at or after t0 (hf_scan.h)
    int chunk_slow0;         // slow-list position of the chunk's first window (its private row: start∘e)
    int seg0, k, nseg;       // first segment of the chunk, this segment's position in it, segments of the chunk
    int reg_first, reg_last; // region of the chunk's first / last window
    int chunk;               // chunk index
    int pad0, pad1;
};
static_assert(sizeof(SegDesc) == 64, "SegDesc is one 64-byte load");

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

// slow-list position of the lane's first slow window: seg.slow0 + the slow windows of the segment before window a.
// Wave scan of the per-lane counts, wave totals through LDS (one block barrier).
template <int NW>
__device__ __forceinline__ int seg_slow_base(const uint32_t* __restrict__ rec_seg, int a, int m, int slow0, int wave, int lane,
                                             int* __restrict__ s_cnt) {
    int cnt = 0;
    for (int i = 0; i < m; i++) cnt += (int) REC_SLOW(rec_seg[a + i]);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
    if (lane == 63) s_cnt[wave] = inc;
    __syncthreads();
    int before = slow0;
    for (int w = 0; w < wave; w++) before += s_cnt[w];
    return before + inc - cnt;
}

// the lane's product of A_t over its m windows (chunk-first windows excluded, as in k_carry's start vector); every row the
// pass uses goes through here: a NaN row raises HF_FLAG_NAN (hmm_utils.c:783-786)
__device__ __forceinline__ void seg_lane_product(const uint32_t* __restrict__ rec_seg, uint32_t rp, int a, int m, int sidx,
                                                 const RowSrc& S, const double* __restrict__ s_tab, M4& Q, unsigned& nan) {
    m4_identity(Q);
    if (m <= 0) return;
    uint32_t r = rec_seg[a];
    double E[16];
    load_row(row_ptr(S, r, rp, sidx), E);
#pragma unroll 1
    for (int i = 0; i < m; i++) {
        const int sn = sidx + (int) REC_SLOW(r);
        uint32_t rn = 0;
        double En[16];
        if (i + 1 < m) { rn = rec_seg[a + i + 1]; load_row(row_ptr(S, rn, r, sn), En); }   // in flight during this window
        if (row_has_nan(E)) nan |= HF_FLAG_NAN;
        if (!REC_FIRST(r)) {
            double Tm[16];
            lds_Tm(s_tab, r, Tm);
            M4 A, R;
#pragma unroll
            for (int k = 0; k < 16; k++) A.m[k] = Tm[HF_PS(k >> 2, k & 3)] * E[HF_PS(k >> 2, k & 3)];
            m4_mul(R, Q, A);
            Q = R;
            m4_renorm(Q);
        }
        if (i + 1 < m) {
#pragma unroll
            for (int k = 0; k < 16; k++) E[k] = En[k];
        }
        r = rn; sidx = sn;
    }
}

// LDS of the segment kernels after the transition tables: wave totals, partial sums, counts, label bytes
template <int NW>
__host__ __device__ constexpr size_t seg_lds_doubles() { return (size_t) NW * 16 + NW + NW; }

// ------------------------------------------------------------------------------------------
// k_seg_prod: product of one segment (chunks of several segments only) -> Pseg[segment][16]
// ------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NW * 64) k_seg_prod(int n_list, const int32_t* __restrict__ seg_list, const SegDesc* __restrict__ sd,
                                                       const uint32_t* __restrict__ rec, const RowSrc S, const DevParams* __restrict__ P,
                                                       double* __restrict__ Pseg) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    double* __restrict__ s_W = s_tab + P->n_regions * HF_TAB_STRIDE;
    int* __restrict__ s_cnt = reinterpret_cast<int*>(s_W + NW * 16 + NW);
    const int g = seg_list[blockIdx.x];
    const SegDesc d = sd[g];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = wave * 64 + lane;
    const int a = j * d.L;
    const int m = d.n - a < d.L ? (d.n - a > 0 ? d.n - a : 0) : d.L;
    const uint32_t* __restrict__ rec_seg = rec + d.t0;
    const int sidx = seg_slow_base<NW>(rec_seg, a, m, d.slow0, wave, lane, s_cnt);
    const uint32_t rp = (m > 0 && !(a == 0 && d.k == 0)) ? rec_seg[a - 1] : 0u;
    M4 Q;
    unsigned nan = 0;
    seg_lane_product(rec_seg, rp, a, m, sidx, S, s_tab, Q, nan);
    // ordered tree product over lanes: after step dd, lane l (l % 2dd == 0) holds the product of lanes l..l+2dd-1
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        M4 Rgt, R;
        m4_shfl_down(Rgt, Q, dd);
        if ((lane & (2 * dd - 1)) == 0) { m4_mul(R, Q, Rgt); Q = R; m4_renorm(Q); }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) s_W[wave * 16 + k] = Q.m[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NW; w++) {
            M4 B, R;
#pragma unroll
            for (int k = 0; k < 16; k++) B.m[k] = s_W[w * 16 + k];
            m4_mul(R, Q, B);
            Q = R;
            m4_renorm(Q);
        }
        double2* dst = reinterpret_cast<double2*>(Pseg + (int64_t) g * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
}

// ------------------------------------------------------------------------------------------
// k_seg_fb: one workgroup per segment: phases A-D of the header.  BWD = false: forward only (EM_runForwardForList,
// hmm.c:790-816): log-likelihood and error flags, nothing else is written.
// ------------------------------------------------------------------------------------------
template <int NW, bool BWD>
__global__ void __launch_bounds__(NW * 64) k_seg_fb(const SegDesc* __restrict__ sd, const uint32_t* __restrict__ rec, const RowSrc S,
                                                     const DevParams* __restrict__ P, const double* __restrict__ Pseg,
                                                     double* __restrict__ recs, double* __restrict__ scale_s,
                                                     int8_t* __restrict__ label, double* __restrict__ seg_ll,
                                                     unsigned* __restrict__ flags) {
    constexpr int NL = NW * 64;
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    double* __restrict__ s_W = s_tab + P->n_regions * HF_TAB_STRIDE;      // [NW][16] wave totals
    double* __restrict__ s_red = s_W + NW * 16;                           // [NW] log-likelihood partials
    int* __restrict__ s_cnt = reinterpret_cast<int*>(s_red + NW);         // [NW] (+ padding to NW doubles)
    int8_t* __restrict__ s_lab = reinterpret_cast<int8_t*>(s_red + 2 * NW);   // [n] labels of the segment
    const int g = blockIdx.x;
    const SegDesc d = sd[g];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = wave * 64 + lane;
    const int L = d.L, n = d.n;
    const int a = j * L;
    const int m = n - a < L ? (n - a > 0 ? n - a : 0) : L;
    const uint32_t* __restrict__ rec_seg = rec + d.t0;
    const int sidx0 = seg_slow_base<NW>(rec_seg, a, m, d.slow0, wave, lane, s_cnt);
    const bool chunk_first = a == 0 && d.k == 0;                           // this lane's first window starts the chunk
    const uint32_t rp0 = (m > 0 && !chunk_first) ? rec_seg[a - 1] : 0u;
    unsigned bad = 0;
    // ---- A: lane product ----
    M4 Q;
    seg_lane_product(rec_seg, rp0, a, m, sidx0, S, s_tab, Q, bad);
    // ---- B: scans over the lanes of the wavefront ----
    double fin[4], bdir[4];
    {
        M4 X;      // exclusive prefix of this lane (product of lanes 0..lane-1 of the wavefront)
        {
            M4 Pq = Q;
#pragma unroll
            for (int d2 = 1; d2 < 64; d2 <<= 1) {
                M4 Lft, R;
                m4_shfl_up(Lft, Pq, d2);
                if (lane >= d2) { m4_mul(R, Lft, Pq); Pq = R; m4_renorm(Pq); }
            }
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 16; k++) s_W[wave * 16 + k] = Pq.m[k];
            }
            m4_shfl_up(X, Pq, 1);
        }
        M4 Y;      // exclusive suffix (product of lanes lane+1..63)
        if (BWD) {
            M4 Sq = Q;
#pragma unroll
            for (int d2 = 1; d2 < 64; d2 <<= 1) {
                M4 Rgt, R;
                m4_shfl_down(Rgt, Sq, d2);
                if (lane + d2 < 64) { m4_mul(R, Sq, Rgt); Sq = R; m4_renorm(Sq); }
            }
            m4_shfl_down(Y, Sq, 1);
        }
        __syncthreads();
        // forward vector entering the segment: start∘e of the chunk's first window (its row is the chunk's first entry of
        // the slow list), through the products of the chunk's earlier segments
        double v[4];
        {
            const DevRegion* __restrict__ R = &P->reg[d.reg_first];
            const double* __restrict__ E0 = S.Es + (int64_t) d.chunk_slow0 * 16;
            double sv = 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) { v[s] = E0[HF_PS(0, s)] * R->trans[4][s]; sv += v[s]; }
#pragma unroll
            for (int s = 0; s < 4; s++) v[s] /= sv;
        }
        for (int q = 0; q < d.k; q++) { v4_mul_right(v, Pseg + (int64_t) (d.seg0 + q) * 16); v4_renorm(v); }
        for (int w = 0; w < wave; w++) { v4_mul_right(v, s_W + w * 16); v4_renorm(v); }
        if (lane > 0) v4_mul_right(v, X.m);
        {
            const double su = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
            for (int s = 0; s < 4; s++) fin[s] = v[s] / su;
        }
        if (chunk_first) { fin[0] = 1.0; fin[1] = 0.0; fin[2] = 0.0; fin[3] = 0.0; }   // (1,0,0,0)·A_first = start∘e
        if (BWD) {
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
            if (lane < 63) v4_mul_left(bdir, Y.m);
        }
    }
    // ---- C: forward replay (hmm.c:333-434) ----
    double f[4] = {fin[0], fin[1], fin[2], fin[3]};
    double ll = 0.0, scl = 1.0;
    uint32_t r_last = 0, r_before_last = rp0;
    int sidx_last = sidx0;
    if (m > 0) {
        uint32_t rp = rp0, r = rec_seg[a];
        int sidx = sidx0;
        double E[16];
        load_row(row_ptr(S, r, rp, sidx), E);
        const int64_t slot_ij = (int64_t) d.slot0 + j;                     // + i*NL
#pragma unroll 1
        for (int i = 0; i < m; i++) {
            const int sn = sidx + (int) REC_SLOW(r);
            uint32_t rn = 0;
            double En[16];
            if (i + 1 < m) { rn = rec_seg[a + i + 1]; load_row(row_ptr(S, rn, r, sn), En); }
            double Tm[16];
            lds_Tm(s_tab, r, Tm);
            double nf[4], sc = 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                double acc = 0.0;
#pragma unroll
                for (int p = 0; p < 4; p++) acc += (f[p] * Tm[HF_PS(p, s)] * E[HF_PS(p, s)]);
                nf[s] = acc;
                sc += acc;
            }
            if (!REC_FIRST(r) && sc < 1e-50) bad |= HF_FLAG_SCALE;        // hmm.c:412-415
#pragma unroll
            for (int s = 0; s < 4; s++) f[s] = nf[s] / sc;
            ll += log(sc);                                                 // hmm.c:428
            scl = sc;
            if (BWD) {
                scale_s[slot_ij + (int64_t) i * NL] = sc;
                // f_t is the first half of record t+1: the lane's next slot, the next lane's first slot, or the next segment's
                int64_t sf = i + 1 < L ? slot_ij + (int64_t) (i + 1) * NL : slot_ij + 1;
                if (a + i + 1 == n) sf = d.next_slot;
                double2* __restrict__ dst = reinterpret_cast<double2*>(recs) + sf * 4;
                dst[0] = make_double2(f[0], f[1]); dst[1] = make_double2(f[2], f[3]);
            }
            if (i + 1 < m) {
#pragma unroll
                for (int k = 0; k < 16; k++) E[k] = En[k];
                r_before_last = r; r = rn; sidx = sn;
            }
        }
        r_last = r; sidx_last = sidx;
    }
    for (int o = 32; o > 0; o >>= 1) ll += __shfl_down(ll, o);
    if (lane == 0) s_red[wave] = ll;
    // ---- D: backward replay + labels (hmm.c:452-545, 671-692) ----
    if (BWD && m > 0) {
        const int jl = m - 1;
        const DevRegion* __restrict__ Rl = &P->reg[d.reg_last];
        double b[4];
        if (d.k == d.nseg - 1 && a + jl == n - 1) {     // the chunk's last window, hmm.c:452-467
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = Rl->trans[s][4] / scl;
        } else {                                        // direction from the scans, magnitude from the invariant at this window
            const double term = Rl->trans[0][4];
            double dot = 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) dot += f[s] * bdir[s];
            const double kk = term / (scl * dot);
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = bdir[s] * kk;
        }
        s_lab[a + jl] = (int8_t) posterior_label(f, b, scl);
        const int64_t slot_ij = (int64_t) d.slot0 + j;
        {
            double2* __restrict__ dst = reinterpret_cast<double2*>(recs) + (slot_ij + (int64_t) jl * NL) * 4 + 2;
            dst[0] = make_double2(b[0], b[1]); dst[1] = make_double2(b[2], b[3]);
        }
        // window k's row and transition table turn b_k into b_{k-1}
        uint32_t rk = r_last, rkm1 = r_before_last;
        int sk = sidx_last;
        double E[16];
        if (jl >= 1) load_row(row_ptr(S, rk, rkm1, sk), E);
#pragma unroll 1
        for (int k = jl; k >= 1; k--) {
            // prefetch: row of window k-1 (needs the record before it)
            const int skm1 = sk - (int) REC_SLOW(rkm1);
            uint32_t rkm2 = 0;
            double En[16];
            if (k >= 2) { rkm2 = (a + k - 2 >= 0 && !(a + k - 2 == -1)) ? rec_seg[a + k - 2] : 0u; load_row(row_ptr(S, rkm1, rkm2, skm1), En); }
            const int64_t slot_prev = slot_ij + (int64_t) (k - 1) * NL;
            const double sc = scale_s[slot_prev];
            const double2* __restrict__ fsrc = reinterpret_cast<const double2*>(recs) + (slot_ij + (int64_t) k * NL) * 4;
            const double2 f01 = fsrc[0], f23 = fsrc[1];
            double Tm[16];
            lds_Tm(s_tab, rk, Tm);
            double nb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int p = 0; p < 4; p++) nb[p] += Tm[HF_PS(p, s)] * E[HF_PS(p, s)] * b[s];
            if (sc < 1e-50) bad |= HF_FLAG_SCALE;                         // hmm.c:521-524
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = nb[s] / sc;
            const double fi[4] = {f01.x, f01.y, f23.x, f23.y};
            s_lab[a + k - 1] = (int8_t) posterior_label(fi, b, sc);
            double2* __restrict__ dst = reinterpret_cast<double2*>(recs) + slot_prev * 4 + 2;
            dst[0] = make_double2(b[0], b[1]); dst[1] = make_double2(b[2], b[3]);
            if (k >= 2) {
#pragma unroll
                for (int q = 0; q < 16; q++) E[q] = En[q];
            }
            rk = rkm1; rkm1 = rkm2; sk = skm1;
        }
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
