// hf_scan.h — HF_ALGO_SCAN: tile-parallel forward/backward for 4-state chains on gfx950.
//
// The scaled forward recurrence f_t ∝ f_{t-1}·A_t (A_t[pre][s] = T_t(pre,s)·e_t(pre,s), hmm.c:366-420)
// is a product of 4x4 non-negative matrices, hence associative.  A chunk is cut into tiles of 64·L
// windows, one wavefront per tile, every lane owning L consecutive windows:
//   k_tables     this iteration's emission rows: one per (region, x, x_prev) that occurs at an interior window, and
//                one per contig-end / chunk-first window (generic evaluation with the window's own beta)
//   k_prod_tile  lane products and the product of every tile (all tiles of all chunks at once)
//   k_carry      per chunk: sweep over its tile products -> carried-in forward vector and carried-in backward
//                direction of every tile
//   k_fb_tile    per tile: Kogge-Stone scan of the 64 lane products (power-of-two renormalisation after every
//                product: exact, nothing underflows) gives each lane the product of everything before it; the lane
//                then REPLAYS its L windows with the reference's exact operation order from the carried-in,
//                normalised vector (hmm.c:333-420).  The same wavefront then runs the mirror image with suffix
//                products (hmm.c:470-529) + posterior argmax (hmm.c:671-692) with f and scale still in registers; the
//                absolute magnitude of a carried-in b vector is recovered from the invariant
//                sum_s f_t[s]·b_t[s]·scale_t = terminationProb of the scaled forward-backward.
// No emission value is ever written to HBM per window: a row is 128 B gathered from the tables (L2-resident) where
// it is used.
#pragma once
#include "hf_device.h"

struct M4 { double m[16]; };   // row-major: m[pre*4 + s]

// Emission rows and the LDS transition tables are stored STATE-major: entry (pre, s) at s*4 + pre, so that the four
// values of one state column are contiguous (the statistics kernel works one column at a time).
#define HF_PS(p, s) ((s) * 4 + (p))

// Conditional-transition and start tables of every region, staged once per block in LDS (136 doubles per region:
// tcond[8][16], start[4], pad): the per-window lookup no longer waits on global memory behind the window record.
#define HF_TAB_STRIDE 136
__device__ __forceinline__ void fill_tab(const DevParams* __restrict__ P, double* __restrict__ s_tab) {
    const int n = P->n_regions * HF_TAB_STRIDE;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / HF_TAB_STRIDE, k = i % HF_TAB_STRIDE;
        // k < 128: mask k>>4, entry (pre, s) of the state-major block at (k & 15) = s*4 + pre
        s_tab[i] = k < 128 ? P->reg[r].tcond[k >> 4][(k & 3) * 4 + ((k >> 2) & 3)] : (k < 132 ? P->reg[r].trans[4][k - 128] : 0.0);
    }
    __syncthreads();
}

__device__ __forceinline__ void lds_Tm(const double* __restrict__ s_tab, uint32_t r, double Tm[16]) {
    if (REC_FIRST(r)) {                       // chunk-first window: the start row for every pre
        const double* __restrict__ s = s_tab + REC_REGION(r) * HF_TAB_STRIDE + 128;
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = s[k >> 2];
    } else if (REC_REGCHG(r)) {               // region change => 1/(S+1), hmm.c:398-400
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = 1.0 / (HF_NSTATES + 1);
    } else {
        const double* __restrict__ s = s_tab + REC_REGION(r) * HF_TAB_STRIDE + REC_VMASK(r) * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = s[k];
    }
}

__device__ __forceinline__ void m4_identity(M4& a) {
#pragma unroll
    for (int i = 0; i < 16; i++) a.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

// c = a·b (fused multiply-adds: scan products only steer carried-in directions)
__device__ __forceinline__ void m4_mul(M4& c, const M4& a, const M4& b) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double s = a.m[i * 4] * b.m[j];
            s = fma(a.m[i * 4 + 1], b.m[4 + j], s);
            s = fma(a.m[i * 4 + 2], b.m[8 + j], s);
            s = fma(a.m[i * 4 + 3], b.m[12 + j], s);
            c.m[i * 4 + j] = s;
        }
}

// scale by a power of two so that the largest entry is in [0.5, 1): exact
__device__ __forceinline__ void m4_renorm(M4& a) {
    double mx = a.m[0];
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmax(mx, a.m[i]);
    int e;
    (void) frexp(mx, &e);
    if (mx > 0.0) {
#pragma unroll
        for (int i = 0; i < 16; i++) a.m[i] = ldexp(a.m[i], -e);
    }
}

__device__ __forceinline__ void m4_shfl_up(M4& dst, const M4& src, int d) {
#pragma unroll
    for (int i = 0; i < 16; i++) dst.m[i] = __shfl_up(src.m[i], d);
}
__device__ __forceinline__ void m4_shfl_down(M4& dst, const M4& src, int d) {
#pragma unroll
    for (int i = 0; i < 16; i++) dst.m[i] = __shfl_down(src.m[i], d);
}

// ------------------------------------------------------------------------------------------
// Emission rows.  Interior windows (beta == beta_star, not chunk-first) read the row of their key
// (region, x, x_prev) from lutE / lutC; the others ("slow": contig ends, chunk-first; REC_SLOW) have a private row
// in Es / Cs, indexed by their position in the context's slow list (ascending window order).  A tile knows the
// list position of its first slow window (TileDesc.slow0); the rank inside the tile comes from ballots.
// ------------------------------------------------------------------------------------------
struct RowSrc {
    const double* lutE;   // [R*M*M][16]
    const double* lutC;   // [R*M*M][K][4]: component probability of the collapsed state, [component][previous state]
    const double* Es;     // [n_slow][16]
    const double* Cs;     // [n_slow][K][4]
    int M, K;
    int n_lut;            // R*M*M: Es / Cs are the rows n_lut.. of the same buffers as lutE / lutC
};

__device__ __forceinline__ int64_t row_key(const RowSrc& S, uint32_t r, uint32_t rprev) {
    return ((int64_t) REC_REGION(r) * S.M + REC_X(r)) * S.M + REC_X(rprev);
}

// rr[j] = record of the lane's j-th window (0 when outside the chunk); sidx[j] = slow-list position if it is slow
template <int L>
__device__ __forceinline__ void slow_index(const uint32_t rr[L], int lane, int slow0, int sidx[L]) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    int below = 0;
#pragma unroll
    for (int j = 0; j < L; j++) below += __popcll(__ballot(REC_SLOW(rr[j]) != 0) & lt);
    int mine = 0;
#pragma unroll
    for (int j = 0; j < L; j++) { sidx[j] = slow0 + below + mine; mine += (int) REC_SLOW(rr[j]); }
}

// wave-uniform: does the tile hold any slow window?  fills sidx (zeros when it does not)
template <int L>
__device__ __forceinline__ void tile_slow_index(const uint32_t rr[L], int lane, int slow0, int sidx[L]) {
    bool slow = false;
#pragma unroll
    for (int i = 0; i < L; i++) { slow |= REC_SLOW(rr[i]) != 0; sidx[i] = 0; }
    if (__any(slow)) slow_index<L>(rr, lane, slow0, sidx);
}

__device__ __forceinline__ const double2* row_ptr(const RowSrc& S, uint32_t r, uint32_t rprev, int sidx) {
    return reinterpret_cast<const double2*>(REC_SLOW(r) ? S.Es + (int64_t) sidx * 16 : S.lutE + row_key(S, r, rprev) * 16);
}

__device__ __forceinline__ const double2* crow_ptr(const RowSrc& S, uint32_t r, uint32_t rprev, int sidx) {
    return reinterpret_cast<const double2*>(REC_SLOW(r) ? S.Cs + ((int64_t) sidx * 4) * S.K
                                                         : S.lutC + (row_key(S, r, rprev) * 4) * S.K);
}

__device__ __forceinline__ int32_t row_index(const RowSrc& S, uint32_t r, uint32_t rprev, int sidx) {
    return REC_SLOW(r) ? (int32_t) (S.n_lut + sidx) : (int32_t) row_key(S, r, rprev);
}

// Cooperative row fetch: every lane needs the 128-byte row `ridx` of `rows`.  A lane reading its own row touches 64
// different cache lines per load instruction; here instruction q fetches rows q*8..q*8+7 with 8 lanes x 16 B each
// (coalesced), and the 16 values of a lane's own row come back through a swizzled, conflict-free LDS transposition.
// ALL 64 lanes must call both halves (lanes without a window pass any valid row, e.g. 0).
//   coop_rows_issue: the global loads (can be issued a whole loop iteration early)
//   coop_rows_finish: LDS round trip through the wave-private buffer xp (64*8 double2 = 8 KiB)
__device__ __forceinline__ void coop_rows_issue(const double* __restrict__ rows, int32_t ridx, int lane, double2 v[8]) {
    const int part = lane & 7, sub = lane >> 3;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int32_t idx = __shfl(ridx, q * 8 + sub);
        v[q] = reinterpret_cast<const double2*>(rows)[(int64_t) idx * 8 + part];
    }
}
__device__ __forceinline__ void coop_rows_finish(const double2 v[8], int lane, double2* __restrict__ xp, double Ev[16]) {
    const int part = lane & 7, sub = lane >> 3;
#pragma unroll
    for (int q = 0; q < 8; q++) { const int src = q * 8 + sub; xp[src * 8 + (part ^ (src & 7))] = v[q]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 d = xp[lane * 8 + (k ^ (lane & 7))]; Ev[2 * k] = d.x; Ev[2 * k + 1] = d.y; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void load_row(const double2* __restrict__ src, double Ev[16]) {
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = src[k]; Ev[2 * k] = v.x; Ev[2 * k + 1] = v.y; }
}

__device__ __forceinline__ bool row_has_nan(const double Ev[16]) {
    bool isnan = false;
#pragma unroll
    for (int k = 0; k < 16; k++) isnan |= Ev[k] != Ev[k];
    return isnan;
}

// the lane's window records plus the record of the window before its first one (x_prev of window 0 of the lane)
template <int L>
__device__ __forceinline__ uint32_t load_recs(const uint32_t* __restrict__ rec, int64_t t0, int64_t T, int64_t a, int lane,
                                              uint32_t rr[L]) {
#pragma unroll
    for (int i = 0; i < L; i++) rr[i] = (a + i < T) ? rec[t0 + a + i] : 0u;
    uint32_t rp = __shfl_up(rr[L - 1], 1);
    if (lane == 0) rp = a > 0 ? rec[t0 + a - 1] : 0u;
    return rp;
}

// ------------------------------------------------------------------------------------------
// k_tables: this iteration's emission rows.
//   job < n_keys            key = (region*M + x)*M + x_prev of an interior window (beta == beta_star):
//                           lutE[key][16] = E[pre][s], lutC[key][K][4] = component probabilities of the collapsed
//                           state, [component][previous state] — the per-iteration constants apply
//   job - n_keys < n_slow   the slow window slow_w[k]: the same two rows with the window's own beta (or the
//                           chunk-first row, hmm.c:338-352), in Es[k] / Cs[k]
// A row costs one ITEM per (state, distinct alpha of its column, mixture component) — one exp each; HiFi alpha,
// K = 6: 1 + 3 + 4 + 3*6 = 26 (DevParams.item_*).  A block takes HF_TABLE_JOBS_PER_BLOCK rows: its 256 threads sweep
// the rows x items grid flat (every lane busy whatever K and the alpha pattern are: with ~1 window per key — coverage spread
// over the whole 0..250 range — this kernel evaluates as many rows as there are windows), park the values in LDS, and then
// sweep the rows x outputs grid: E[pre][s] (the collapsed state: components summed in index order, hmm_utils.c:753-758)
// and the component table.  The same device functions as a direct per-window evaluation, so the values are identical.
// NaNs are stored, not reported: only a window that actually uses the row raises HF_E_NAN.
// ------------------------------------------------------------------------------------------
// rows per block: few when there are few rows (more blocks than CUs: the kernel is then latency-bound), many otherwise
#define HF_TABLE_JOBS_SMALL 8
#define HF_TABLE_JOBS_LARGE 32
// flags: 1 star (table key), 2 first, 4 active; bits 8..: transition class of the job's row of A (hf_seg.h); row: index of the row in lutE / lutC units
struct TableJob { double x, px, bt; int64_t row; int32_t r; int32_t flags; };
template <int HF_TABLE_JOBS_PER_BLOCK>
__global__ void __launch_bounds__(256) k_tables(int n_keys, const int32_t* __restrict__ keys, int n_slow,
                                                const int64_t* __restrict__ slow_w, const uint32_t* __restrict__ rec,
                                                const double* __restrict__ beta, int M, int K,
                                                const DevParams* __restrict__ P, double* __restrict__ lutE,
                                                double* __restrict__ lutC, double* __restrict__ Es,
                                                double* __restrict__ Cs, unsigned* __restrict__ flags,
                                                const int32_t* __restrict__ cls, double* __restrict__ lutA) {
    // cls / lutA (statistics by emission row, hf_seg.h): job j also writes row j of A = (transition table of class cls[j]) ∘ E;
    // `keys` is then the list of the (key, class) pairs that occur — a key with two classes is evaluated twice (same values)
    __shared__ TableJob s_job[HF_TABLE_JOBS_PER_BLOCK];
    __shared__ double s_val[HF_TABLE_JOBS_PER_BLOCK][HF_TABLE_MAX_ITEMS];
    __shared__ int s_base[16];
    __shared__ unsigned char s_item[3][HF_TABLE_MAX_ITEMS];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *flags = 0u;   // first kernel of every pass
    const int ncol = P->ncomp[3], n_items = P->n_items;
    const bool te = hf_err_is_truncexp(P);
    const int job0 = blockIdx.x * HF_TABLE_JOBS_PER_BLOCK;
    const int n_lut_rows = (int) ((Es - lutE) / 16);   // Es / Cs are the rows n_lut.. of the same buffers
    if (tid < HF_TABLE_JOBS_PER_BLOCK) {
        const int job = job0 + tid;
        TableJob J;
        J.x = 0.0; J.px = 0.0; J.bt = P->beta_star; J.row = 0; J.r = 0; J.flags = 0;
        if (job < n_keys) {
            const int64_t key = keys[job];
            const int64_t MM = (int64_t) M * M;
            const int64_t idx = key % MM;
            J.r = (int) (key / MM); J.x = (double) (idx / M); J.px = (double) (idx % M); J.row = key; J.flags = 1 | 4;
            if (cls) J.flags |= (cls[job] & 0xff) << 8;
        } else if (job - n_keys < n_slow) {
            const int k = job - n_keys;
            const int64_t t = slow_w[k];
            const uint32_t rw = rec[t];
            const bool first = REC_FIRST(rw) != 0;
            J.r = (int) REC_REGION(rw); J.x = (double) REC_X(rw); J.px = first ? 0.0 : (double) REC_X(rec[t - 1]);
            J.bt = beta[t]; J.row = (int64_t) n_lut_rows + k; J.flags = (first ? 2 : 0) | 4;
            if (cls) J.flags |= (cls[job] & 0xff) << 8;
        }
        s_job[tid] = J;
    }
    if (tid < 16) s_base[tid] = P->item_base[tid];
    if (tid < n_items) { s_item[0][tid] = P->item_s[tid]; s_item[1][tid] = P->item_u[tid]; s_item[2][tid] = P->item_c[tid]; }
    __syncthreads();
    // ---- the items ----
    unsigned nan = 0;
    for (int w = tid; w < HF_TABLE_JOBS_PER_BLOCK * n_items; w += 256) {
        const int jl = w / n_items, it = w - jl * n_items;
        const TableJob J = s_job[jl];
        if (!(J.flags & 4)) continue;
        const int s = s_item[0][it], u = s_item[1][it], c = s_item[2][it];
        const bool star = (J.flags & 1) != 0, first = (J.flags & 2) != 0;
        const DevRegion* __restrict__ R = &P->reg[J.r];
        double v;
        if (s == 0 && te) v = star ? hf_trunc_exp_star(R, J.x) : hf_trunc_exp(R->lambda, R->trunc_point, J.x, J.bt);
        else {
            const double alpha = first ? 0.0 : P->ualpha[s][u];
            v = star ? hf_gauss_comp_star(R->m1[s][u][c], R->gvar[s][c], R->gnorm[s][c], J.x, J.px, alpha, J.bt, &nan)
                     : hf_gauss_comp(R->mean[s][c], R->var[s][c], R->weight[s][c], J.x, J.px, alpha, J.bt, &nan);
        }
        s_val[jl][it] = v;
    }
    __syncthreads();
    // ---- the outputs: 16 values of E and 4*ncol of the component table per row ----
    const int n_out = 16 + 4 * ncol;
    for (int w = tid; w < HF_TABLE_JOBS_PER_BLOCK * n_out; w += 256) {
        const int jl = w / n_out, o = w - jl * n_out;
        const TableJob J = s_job[jl];
        if (!(J.flags & 4)) continue;
        const bool first = (J.flags & 2) != 0;
        if (o < 16) {
            const int pre = o >> 2, s = o & 3;
            const int u = (first || (s == 0 && te)) ? 0 : P->umap[pre * 4 + s];
            const int b0 = s_base[s * 4 + u];
            double e;
            if (s < 3) e = s_val[jl][b0];
            else {
                e = 0.0;
                for (int c = 0; c < ncol; c++) e += s_val[jl][b0 + c];
            }
            if (first && pre != 0) e = 0.0;
            lutE[J.row * 16 + HF_PS(pre, s)] = e;
            if (lutA) {
                const int k = J.flags >> 8;
                const DevRegion* __restrict__ R = &P->reg[J.r];
                const double t = k == 9 ? R->trans[4][s] : (k == 8 ? 1.0 / (HF_NSTATES + 1) : R->tcond[k][pre * 4 + s]);
                lutA[(int64_t) (job0 + jl) * 16 + HF_PS(pre, s)] = t * e;
            }
        } else {   // component probabilities per PREVIOUS STATE (the value of its alpha): no select in the consumer
            const int q = o - 16, c = q >> 2, pre = q & 3;
            const int u = first ? 0 : P->umap[pre * 4 + 3];
            lutC[(J.row * 4) * K + c * 4 + pre] = first ? 0.0 : s_val[jl][s_base[3 * 4 + u] + c];
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_prod_tile: one wavefront per tile.  Each lane multiplies the matrices A = T∘E of its L windows into its lane
// product Q_l (written to Qs: the scans of k_fb_tile start from it), and an ordered shuffle tree over the lanes
// gives the tile product Pt.  Chunk-first windows are excluded from Q_l and Pt.  Every row a pass uses goes
// through here once: this is where a NaN row raises HF_E_NAN (hmm_utils.c:783-786).
// ------------------------------------------------------------------------------------------
#ifndef HF_PROD_BLOCKS
#define HF_PROD_BLOCKS 3
#endif
template <int L>
__global__ void __launch_bounds__(256, HF_PROD_BLOCKS) k_prod_tile(int ntiles, const TileDesc* __restrict__ td,
                                                   const uint32_t* __restrict__ rec, const DevParams* __restrict__ P,
                                                   const RowSrc S, double* __restrict__ Qs, double* __restrict__ Pt,
                                                   unsigned* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const TileDesc d = td[tile];
    const int64_t t0 = d.t0, T = d.T, base = d.base;
    const int64_t a = base + (int64_t) lane * L;
    uint32_t rr[L];
    const uint32_t rp = load_recs<L>(rec, t0, T, a, lane, rr);
    int sidx[L];
    tile_slow_index<L>(rr, lane, d.slow0, sidx);
    // rows through the cooperative fetch: window i+1's loads are in flight while window i is multiplied in
    double2* __restrict__ xp = reinterpret_cast<double2*>(s_tab + P->n_regions * HF_TAB_STRIDE) + (threadIdx.x >> 6) * 512;
    int32_t ridx[L];
#pragma unroll
    for (int i = 0; i < L; i++) ridx[i] = (a + i < T) ? row_index(S, rr[i], i == 0 ? rp : rr[i - 1], sidx[i]) : 0;
    double2 vq[8];
    coop_rows_issue(S.lutE, ridx[0], lane, vq);
    unsigned nan = 0;
    M4 Q;
    m4_identity(Q);
#pragma unroll
    for (int i = 0; i < L; i++) {
        double Ecur[16];
        coop_rows_finish(vq, lane, xp, Ecur);
        if (i + 1 < L) coop_rows_issue(S.lutE, ridx[i + 1], lane, vq);
        if (a + i < T) {
            if (row_has_nan(Ecur)) nan |= HF_FLAG_NAN;
            if (!REC_FIRST(rr[i])) {
                double Tm[16];
                lds_Tm(s_tab, rr[i], Tm);
                M4 A, R;
#pragma unroll
                for (int k = 0; k < 16; k++) A.m[k] = Tm[HF_PS(k >> 2, k & 3)] * Ecur[HF_PS(k >> 2, k & 3)];
                m4_mul(R, Q, A);
                Q = R;
                m4_renorm(Q);
            }
        }
    }
    {   // lane-minor: the 64 lanes of a wavefront write / read 1 KiB contiguous per instruction
        double2* dst = reinterpret_cast<double2*>(Qs) + (int64_t) tile * 8 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k * 64] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    // ordered tree product over lanes: after step d, lane l (l % 2d == 0) holds the product of lanes l..l+2d-1
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        M4 Rgt, R;
        m4_shfl_down(Rgt, Q, dd);
        if ((lane & (2 * dd - 1)) == 0) { m4_mul(R, Q, Rgt); Q = R; m4_renorm(Q); }
    }
    if (lane == 0) {
        double2* dst = reinterpret_cast<double2*>(Pt + (int64_t) tile * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    if (nan) atomicOr(flags, nan);
}

__device__ __forceinline__ void load_lane_product(M4& Q, const double* __restrict__ Qs, int tile, int lane) {
    const double2* __restrict__ src = reinterpret_cast<const double2*>(Qs) + (int64_t) tile * 8 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = src[k * 64]; Q.m[2 * k] = v.x; Q.m[2 * k + 1] = v.y; }
}

// ------------------------------------------------------------------------------------------
// k_carry: per chunk, sweep the tile products.  cf[tile] = normalised forward vector entering the tile
// (tile 0 of a chunk: unused, the tile kernel starts from window 0); cb[tile] = direction of b at the LAST window
// of the tile (= product of all later tiles applied to the end vector).
// Wave 0 sweeps forward, wave 1 backward; each stages the tile products through its half of LDS in batches of
// HF_CARRY_BATCH tiles (coalesced loads by all 64 lanes), then lane 0 runs the dependent chain out of LDS.
// ------------------------------------------------------------------------------------------
#define HF_CARRY_BATCH 128
__global__ void __launch_bounds__(128) k_carry(const CarryDesc* __restrict__ cdesc, const double* __restrict__ Es,
                                               const DevParams* __restrict__ P, const double* __restrict__ Pt,
                                               double* __restrict__ cf, double* __restrict__ cb) {
    __shared__ __attribute__((aligned(16))) double s_pt[2][HF_CARRY_BATCH * 16];
    __shared__ __attribute__((aligned(16))) double s_out[2][HF_CARRY_BATCH * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const CarryDesc D = cdesc[blockIdx.x];
    const int k0 = D.k0, nt = D.nt;
    if (nt <= 0) return;
    double* __restrict__ sp = s_pt[wave];
    double* __restrict__ so = s_out[wave];
    double v[4];
    if (wave == 0) {   // start∘e of the chunk's first window (its row is the chunk's first entry of the slow list)
        const DevRegion* __restrict__ R = &P->reg[D.reg_first];
        const double* __restrict__ E0 = Es + (int64_t) D.slow0 * 16;
        double sv = 0.0;
#pragma unroll
        for (int s = 0; s < 4; s++) { v[s] = E0[HF_PS(0, s)] * R->trans[4][s]; sv += v[s]; }
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] /= sv;
    } else {
        const DevRegion* __restrict__ R = &P->reg[D.reg_last];
        double sw = 0.0;
#pragma unroll
        for (int s = 0; s < 4; s++) { v[s] = R->trans[s][4]; sw += v[s]; }
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] /= sw;
    }
    const int nb = (nt + HF_CARRY_BATCH - 1) / HF_CARRY_BATCH;
    for (int bi = 0; bi < nb; bi++) {
        // forward takes the batches in increasing order, backward in decreasing order
        const int b0 = wave == 0 ? bi * HF_CARRY_BATCH : (nb - 1 - bi) * HF_CARRY_BATCH;
        const int n = nt - b0 < HF_CARRY_BATCH ? nt - b0 : HF_CARRY_BATCH;
        {
            const double2* __restrict__ src = reinterpret_cast<const double2*>(Pt + (int64_t) (k0 + b0) * 16);
            double2* dst = reinterpret_cast<double2*>(sp);
            for (int i = lane; i < n * 8; i += 64) dst[i] = src[i];
        }
        __builtin_amdgcn_s_waitcnt(0);   // the wave's own LDS stores have landed (one wave per half: no block barrier)
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // the chain carries the vector renormalised by a power of two (exact, and nothing else on the dependent
            // path: one lane of one wavefront per SIMD pays the full latency of every instruction)
            if (wave == 0) {
                for (int k = 0; k < n; k++) {
                    so[k * 4 + 0] = v[0]; so[k * 4 + 1] = v[1]; so[k * 4 + 2] = v[2]; so[k * 4 + 3] = v[3];
                    const double* __restrict__ M = sp + k * 16;
                    double u[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        double s = v[0] * M[j];
                        s = fma(v[1], M[4 + j], s); s = fma(v[2], M[8 + j], s); s = fma(v[3], M[12 + j], s);
                        u[j] = s;
                    }
                    int e;
                    (void) frexp(fmax(fmax(u[0], u[1]), fmax(u[2], u[3])), &e);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = ldexp(u[j], -e);
                }
            } else {
                for (int k = n - 1; k >= 0; k--) {        // only the direction of b is used
                    so[k * 4 + 0] = v[0]; so[k * 4 + 1] = v[1]; so[k * 4 + 2] = v[2]; so[k * 4 + 3] = v[3];
                    const double* __restrict__ M = sp + k * 16;
                    double u[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        double s = M[i * 4] * v[0];
                        s = fma(M[i * 4 + 1], v[1], s); s = fma(M[i * 4 + 2], v[2], s); s = fma(M[i * 4 + 3], v[3], s);
                        u[i] = s;
                    }
                    int e;
                    (void) frexp(fmax(fmax(u[0], u[1]), fmax(u[2], u[3])), &e);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = ldexp(u[i], -e);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (wave == 0) {
            // what a tile starts from is the carried vector divided by its sum (the reference's f sums to 1): one tile
            // per lane, off the chain
            double2* __restrict__ dst = reinterpret_cast<double2*>(cf + (int64_t) (k0 + b0) * 4);
            const double2* __restrict__ so2 = reinterpret_cast<const double2*>(so);
            for (int k = lane; k < n; k += 64) {
                const double2 a = so2[k * 2], b = so2[k * 2 + 1];
                const double sv = a.x + a.y + b.x + b.y;
                dst[k * 2] = make_double2(a.x / sv, a.y / sv); dst[k * 2 + 1] = make_double2(b.x / sv, b.y / sv);
            }
        } else {
            double* __restrict__ dst = cb + (int64_t) (k0 + b0) * 4;
            for (int i = lane; i < n * 4; i += 64) dst[i] = so[i];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// k_fb_tile: one wavefront per tile, forward then (BWD) backward + posterior labels.
// Lane l owns windows a = base + l*L .. a+L-1 in both directions.  Backward: b_i = A_{i+1}·b_{i+1} / scale_i
// (hmm.c:470-529); the vector a lane starts from is b at its own LAST window: direction = (product of the lane
// products after it)·cb[tile], magnitude from sum_s f·b·scale = terminationProb with the lane's own f and scale;
// the chunk's last window is b_{T-1}[s] = M[s][End] / scale_{T-1} (hmm.c:452-467).
// ------------------------------------------------------------------------------------------
// f and scale of a lane's windows wait for the backward half in LDS (not in 40 VGPRs): 165 VGPRs, 3 blocks per CU
// (measured 77 -> 68 us against 190 VGPRs / 2 blocks; at 4 blocks the kernel spills)
#ifndef HF_FB_BLOCKS
#define HF_FB_BLOCKS 3
#endif
// RECS (with BWD): instead of the lane-minor arrays F, B the pass writes one 64-byte PAIR RECORD per window into F —
// record t = { f_{t-1}[4], b_t[4] } (window-major, N+1 records) — what the statistics by emission row read (hf_rows.h);
// the halves are written out of LDS by neighbouring lanes (16 cache lines per store instruction instead of 64).
#define HF_FW_STRIDE 65   // doubles between the rows of the wave-private f / scale block: conflict-free both ways
template <int L, bool BWD, bool RECS = false>
__global__ void __launch_bounds__(256, HF_FB_BLOCKS) k_fb_tile(int ntiles, const TileDesc* __restrict__ td,
                                                 const uint32_t* __restrict__ rec, const RowSrc S,
                                                 const double* __restrict__ Qs, const DevParams* __restrict__ P,
                                                 const double* __restrict__ cf, const double* __restrict__ cb,
                                                 double* __restrict__ F, double* __restrict__ scale,
                                                 double* __restrict__ B, int8_t* __restrict__ label,
                                                 double* __restrict__ tile_ll, unsigned* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const TileDesc d = td[tile];
    const int64_t t0 = d.t0, T = d.T, base = d.base;
    const int64_t a = base + (int64_t) lane * L;
    uint32_t rr[L];
    const uint32_t rp = load_recs<L>(rec, t0, T, a, lane, rr);
    int sidx[L];
    tile_slow_index<L>(rr, lane, d.slow0, sidx);
    const double2* rowp[L];
#pragma unroll
    for (int i = 0; i < L; i++) rowp[i] = row_ptr(S, rr[i], i == 0 ? rp : rr[i - 1], sidx[i]);
    double Ecur[16];
    if (a < T) load_row(rowp[0], Ecur);           // in flight during the scan
    double carry[4];
    if (base == 0) { carry[0] = 1.0; carry[1] = 0.0; carry[2] = 0.0; carry[3] = 0.0; }  // (1,0,0,0)·A_first = start∘e
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) carry[j] = cf[(int64_t) tile * 4 + j];
    }
    unsigned bad = 0;
    // ---- forward: exclusive prefix product over lanes ----
    double f[4];
    {
        M4 Q;
        load_lane_product(Q, Qs, tile, lane);
        if (base == 0 && lane == 0) {   // the chunk's first window is not in Q_l: prepend A_first = start∘e (row 0 only)
            double Tm[16];
            lds_Tm(s_tab, rr[0], Tm);
            M4 A, R;
#pragma unroll
            for (int k = 0; k < 16; k++) A.m[k] = Tm[HF_PS(k >> 2, k & 3)] * Ecur[HF_PS(k >> 2, k & 3)];
            m4_mul(R, A, Q);
            Q = R;
            m4_renorm(Q);
        }
#pragma unroll
        for (int d2 = 1; d2 < 64; d2 <<= 1) {
            M4 Lft, R;
            m4_shfl_up(Lft, Q, d2);
            if (lane >= d2) { m4_mul(R, Lft, Q); Q = R; m4_renorm(Q); }
        }
        M4 X;
        m4_shfl_up(X, Q, 1);
        double u[4], su = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double s = carry[0] * X.m[j];
            s = fma(carry[1], X.m[4 + j], s); s = fma(carry[2], X.m[8 + j], s); s = fma(carry[3], X.m[12 + j], s);
            u[j] = s; su += s;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = (lane == 0) ? carry[j] : u[j] / su;
    }
    // replay this lane's windows in the reference's operation order (hmm.c:333-420); f and scale stay in registers
    // f and scale of the lane's windows wait for the backward half in wave-private LDS, lane-minor (conflict-free)
    double* __restrict__ s_fw = s_tab + P->n_regions * HF_TAB_STRIDE + (threadIdx.x >> 6) * (L * 5 * HF_FW_STRIDE) + lane;
#define FW(i, s) s_fw[((i) * 5 + (s)) * HF_FW_STRIDE]
#define SCW(i) s_fw[((i) * 5 + 4) * HF_FW_STRIDE]
    double ll = 0.0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        double Enext[16];
        if (i + 1 < L && a + i + 1 < T) load_row(rowp[i + 1], Enext);   // next window's row is in flight during this one
        if (a + i < T) {
            const int64_t t = t0 + a + i;
            const uint32_t r = rr[i];
            double Tm[16];
            lds_Tm(s_tab, r, Tm);
            double nf[4], sc = 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                double acc = 0.0;
#pragma unroll
                for (int p = 0; p < 4; p++) acc += (f[p] * Tm[HF_PS(p, s)] * Ecur[HF_PS(p, s)]);
                nf[s] = acc;
                sc += acc;
            }
            if (!REC_FIRST(r) && sc < 1e-50) bad |= HF_FLAG_SCALE;   // hmm.c:412-415
#pragma unroll
            for (int s = 0; s < 4; s++) f[s] = nf[s] / sc;
            ll += log(sc);                                            // hmm.c:428
            if (!RECS) {
                reinterpret_cast<double2*>(F)[fb_slot<L>(tile, lane, i, 0)] = make_double2(f[0], f[1]);
                reinterpret_cast<double2*>(F)[fb_slot<L>(tile, lane, i, 1)] = make_double2(f[2], f[3]);
            }
            scale[t] = sc;
#pragma unroll
            for (int s = 0; s < 4; s++) FW(i, s) = f[s];
            SCW(i) = sc;
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) FW(i, s) = 0.0;
            SCW(i) = 1.0;
        }
        if (i + 1 < L) {
#pragma unroll
            for (int k = 0; k < 16; k++) Ecur[k] = Enext[k];
        }
    }
    for (int o = 32; o > 0; o >>= 1) ll += __shfl_down(ll, o);
    if (lane == 0) tile_ll[tile] = ll;
    // pair records: window k of the tile is written by lanes 2k', 2k'+1 (16 bytes each) out of the wave's LDS block
    // (wave-uniform record base in scalar registers, one 32-bit lane offset: no 64-bit address arithmetic per store)
#define HF_COOP_HALF(DST_OFF, REC_SHIFT)                                                                               \
    {                                                                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                          \
        __builtin_amdgcn_wave_barrier();                                                                                \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                          \
        const double* __restrict__ s_w = s_fw - lane;                                                                   \
        const int64_t rec0 = t0 + base;                                                                                 \
        const int64_t rec0u = ((int64_t) __builtin_amdgcn_readfirstlane((int) (rec0 >> 32)) << 32) |                    \
                              (uint32_t) __builtin_amdgcn_readfirstlane((int) rec0);                                    \
        double2* __restrict__ Pw = reinterpret_cast<double2*>(F) + rec0u * 4;                                           \
        const int nvalid = __builtin_amdgcn_readfirstlane((int) (T - base < 64 * L ? T - base : 64 * L));               \
        const int sh = lane & 1, kl = lane >> 1;                                                                        \
        _Pragma("unroll") for (int q = 0; q < (64 * L) / 32; q++) {                                                     \
            const int k = 32 * q + kl, owner = k / L, i = k % L;                                                        \
            if (k < nvalid) {                                                                                           \
                const double2 v = make_double2(s_w[(i * 5 + 2 * sh) * HF_FW_STRIDE + owner],                            \
                                               s_w[(i * 5 + 2 * sh + 1) * HF_FW_STRIDE + owner]);                       \
                Pw[(k + (REC_SHIFT)) * 4 + (DST_OFF) + sh] = v;                                                         \
            }                                                                                                           \
        }                                                                                                               \
    }
    if (RECS) HF_COOP_HALF(0, 1)      // f_t goes into record t+1
    if (BWD) {
        // ---- backward: exclusive SUFFIX product over lanes ----
        const int64_t Tm1 = T - 1;
        int jl = (int) (T - a < L ? T - a : L) - 1;          // the lane's last window inside the chunk (< 0: none)
        double b[4];
        double Er[16];
        if (jl == L - 1 && L >= 2) load_row(rowp[L - 1], Er);   // row of the first replayed step, in flight during the scan
        {
            M4 Q;
            load_lane_product(Q, Qs, tile, lane);
#pragma unroll
            for (int d2 = 1; d2 < 64; d2 <<= 1) {
                M4 Rgt, R;
                m4_shfl_down(Rgt, Q, d2);
                if (lane + d2 < 64) { m4_mul(R, Q, Rgt); Q = R; m4_renorm(Q); }
            }
            M4 X;
            m4_shfl_down(X, Q, 1);
            double cv[4];
#pragma unroll
            for (int s = 0; s < 4; s++) cv[s] = cb[(int64_t) tile * 4 + s];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double s = X.m[i * 4] * cv[0];
                s = fma(X.m[i * 4 + 1], cv[1], s); s = fma(X.m[i * 4 + 2], cv[2], s); s = fma(X.m[i * 4 + 3], cv[3], s);
                b[i] = lane == 63 ? cv[i] : s;
            }
        }
        if (jl >= 0) {
            const uint32_t rlast = rec[t0 + T - 1];
            const DevRegion* __restrict__ Rl = &P->reg[REC_REGION(rlast)];
            double fl[4] = {0.0, 0.0, 0.0, 0.0}, scl = 1.0;      // f / scale of the lane's last window
#pragma unroll
            for (int i = 0; i < L; i++)
                if (i == jl) {
#pragma unroll
                    for (int s = 0; s < 4; s++) fl[s] = FW(i, s);
                    scl = SCW(i);
                }
            if (a + jl == Tm1) {   // hmm.c:452-467
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = Rl->trans[s][4] / scl;
            } else {               // jl == L-1: direction from the scan, magnitude from the invariant at this window
                const double term = Rl->trans[0][4];
                double dot = 0.0;
#pragma unroll
                for (int s = 0; s < 4; s++) dot += fl[s] * b[s];
                const double k = term / (scl * dot);
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] *= k;
            }
            {
                const int64_t t = t0 + a + jl;
                {   // jl is wave-divergent only in a chunk's last tile
                    if (!RECS) {
                        const int64_t s0 = fb_slot<L>(tile, lane, jl, 0);
                        reinterpret_cast<double2*>(B)[s0] = make_double2(b[0], b[1]);
                        reinterpret_cast<double2*>(B)[s0 + 64] = make_double2(b[2], b[3]);
                    }
                }
                label[t] = (int8_t) posterior_label(fl, b, scl);
                if (RECS) {   // f of this window is not needed any more: its LDS slots take b for the record write
#pragma unroll
                    for (int i = 0; i < L; i++)
                        if (i == jl) {
#pragma unroll
                            for (int s = 0; s < 4; s++) FW(i, s) = b[s];
                        }
                }
            }
        }
        // replay the other windows (decreasing) in the reference's operation order: window i uses the row of i+1
#pragma unroll
        for (int i = L - 2; i >= 0; i--) {
            double Enext[16];
            if (i >= 1 && i <= jl) load_row(rowp[i], Enext);          // row of window i, used by window i-1
            if (i < jl) {
                const int64_t t = t0 + a + i;
                double Tm[16];
                lds_Tm(s_tab, rr[i + 1], Tm);
                double nb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int p = 0; p < 4; p++) nb[p] += Tm[HF_PS(p, s)] * Er[HF_PS(p, s)] * b[s];
                const double sc = SCW(i);
                if (sc < 1e-50) bad |= HF_FLAG_SCALE;                 // hmm.c:521-524
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = nb[s] / sc;
                if (!RECS) {
                    reinterpret_cast<double2*>(B)[fb_slot<L>(tile, lane, i, 0)] = make_double2(b[0], b[1]);
                    reinterpret_cast<double2*>(B)[fb_slot<L>(tile, lane, i, 1)] = make_double2(b[2], b[3]);
                }
                {
                    const double fi[4] = {FW(i, 0), FW(i, 1), FW(i, 2), FW(i, 3)};
                    label[t] = (int8_t) posterior_label(fi, b, sc);
                }
                if (RECS) {
#pragma unroll
                    for (int s = 0; s < 4; s++) FW(i, s) = b[s];
                }
            }
            if (i >= 1) {
#pragma unroll
                for (int k = 0; k < 16; k++) Er[k] = Enext[k];
            }
        }
    }
    if (BWD && RECS) HF_COOP_HALF(2, 0)   // b_t goes into record t
#undef HF_COOP_HALF
#undef FW
#undef SCW
    if (bad) atomicOr(flags, bad);
}
