// hf_scan.h — what the kernels of HF_ALGO_SCAN share: the 4x4 matrix helpers, the LDS transition tables, the emission-row
// tables and k_tables.
//
// The scaled forward recurrence f_t ∝ f_{t-1}·A_t (A_t[pre][s] = T_t(pre,s)·e_t(pre,s), hmm.c:366-420) is a product of 4x4
// non-negative matrices, hence associative: the segment kernels (hf_seg.h) cut a chunk into segments and lanes, scan the lane
// products and replay every lane's windows.  (Round 1's tile kernels k_prod_tile / k_carry / k_fb_tile lived here; since
// round 2 every HF_ALGO_SCAN pass runs the segment kernels and the per-chunk statistics read their pair records.)
//   k_tables     this iteration's emission rows: one per (region, x, x_prev) that occurs at an interior window, and
//                one per contig-end / chunk-first window (generic evaluation with the window's own beta); and the rows of
//                A = T∘e of hf_seg.h
// No emission value is ever written to HBM per window: a row is 128 B gathered from the tables (L2-resident) where
// it is used.
#pragma once
#include <type_traits>
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
#define HF_TABLE_JOBS_SMALL 8    // (measured again in round 3: 4 / 16 / 32 rows per block are within 1 us at 9 k and 25 k rows; 32 at 260 k rows: 8 / 16 / 64 are 20-45 % slower)
#define HF_TABLE_JOBS_LARGE 32
// TableJob: hf_device.h
struct NoKParams { int32_t n_ranges; };
// TabWork (what the table work reads and writes): hf_device.h
// the item list of a row as the evaluation reads it: LDS copies (k_tables) or the parameter block's own arrays
struct TabItems { const int32_t* base; const uint8_t *s, *u, *c; };

__device__ __forceinline__ void tab_store2(double* p, double a, double b) { *reinterpret_cast<double2*>(p) = make_double2(a, b); }   // two adjacent doubles, one 16-byte store
template <int NT>
__device__ __forceinline__ void tab_sync() {
    if constexpr (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __syncthreads();
}

// the job list of a context, once (hf_create): job < n_keys is the (emission key [, transition class]) pair keys[job] / cls[job] of an interior
// window, job - n_keys < n_slow the slow window slow_w[k] with its own beta (round 5: a pass used to rebuild this from keys / records /
// betas every time — three dependent global loads at the head of a latency-bound kernel)
__global__ void __launch_bounds__(256) k_build_jobs(int n_keys, const int32_t* __restrict__ keys, const int32_t* __restrict__ cls, int n_slow,
                                                    const int64_t* __restrict__ slow_w, const uint32_t* __restrict__ rec, const double* __restrict__ beta,
                                                    int M, int64_t n_lut_rows, double beta_star, TableJob* __restrict__ jobs) {
    const int job = blockIdx.x * 256 + threadIdx.x;
    if (job >= n_keys + n_slow) return;
    TableJob J;
    J.x = 0.0; J.px = 0.0; J.bt = beta_star; J.row = 0; J.r = 0; J.flags = 0;
    if (job < n_keys) {
        const int64_t key = keys[job];
        const int64_t MM = (int64_t) M * M;
        const int64_t idx = key % MM;
        J.r = (int) (key / MM); J.x = (double) (idx / M); J.px = (double) (idx % M); J.row = key; J.flags = 1 | 4;
    } else {
        const int k = job - n_keys;
        const int64_t t = slow_w[k];
        const uint32_t rw = rec[t];
        const bool first = REC_FIRST(rw) != 0;
        J.r = (int) REC_REGION(rw); J.x = (double) REC_X(rw); J.px = first ? 0.0 : (double) REC_X(rec[t - 1]);
        J.bt = beta[t]; J.row = n_lut_rows + k; J.flags = (first ? 2 : 0) | 4;
    }
    if (cls) J.flags |= (cls[job] & 0xff) << 8;
    jobs[job] = J;
}

// the job of thread tid (< JOBS) of the unit whose first job is job0 (one 48-byte load; idle past the list's end)
template <int JOBS>
__device__ __forceinline__ TableJob tables_load_job(const TabWork& W, int job0, int tid) {
    TableJob J;
    J.x = 0.0; J.px = 0.0; J.bt = 0.0; J.row = 0; J.r = 0; J.flags = 0;
    if (tid < JOBS && job0 + tid < W.n_jobs) J = W.jobs[job0 + tid];
    return J;
}

// one unit of JOBS rows by NT threads: the rows x items grid flat (one exp each) into s_val, then the rows x outputs grid — E[pre][s],
// the row of A = (transition table of the job's class)∘E and the component table, two adjacent values (previous states 2h, 2h + 1 of
// one state / component: the tables are state-major) per thread and store.  (Round 5 also ran this function inside k_seg_fb — the tables
// by the first workgroups of the segment kernel's own launch, one launch less per pass: 1.5 .. 10 us SLOWER per pass at every input size,
// the in-launch wait for the tables is a chain of ~8 dependent global round trips where a kernel boundary costs ~2 us;
// profiles/r05_ab_tab_*.txt, profiles/r05_tab_fused_experiment.patch.)
template <int JOBS, int NT>
__device__ __forceinline__ void tables_eval(const TabWork& W, const DevParams* __restrict__ P, const TabItems It, int job0, int tid, TableJob J,
                                            TableJob* __restrict__ s_job, double (*__restrict__ s_val)[HF_TABLE_MAX_ITEMS]) {
    const int ncol = P->ncomp[3], n_items = P->n_items;
    const bool te = hf_err_is_truncexp(P);
    if (tid < JOBS) s_job[tid] = J;
    tab_sync<NT>();
    // ---- the items ----
    unsigned nan = 0;
    for (int w = tid; w < JOBS * n_items; w += NT) {
        const int jl = w / n_items, it = w - jl * n_items;
        const TableJob Jw = s_job[jl];
        if (!(Jw.flags & 4)) continue;
        const int s = It.s[it], u = It.u[it], c = It.c[it];
        const bool star = (Jw.flags & 1) != 0, first = (Jw.flags & 2) != 0;
        const DevRegion* __restrict__ R = &P->reg[Jw.r];
        double v;
        if (s == 0 && te) v = star ? hf_trunc_exp_star(R, Jw.x) : hf_trunc_exp(R->lambda, R->trunc_point, Jw.x, Jw.bt);
        else {
            const double alpha = first ? 0.0 : P->ualpha[s][u];
            v = star ? hf_gauss_comp_star(R->m1[s][u][c], R->gvar[s][c], R->gnorm[s][c], Jw.x, Jw.px, alpha, Jw.bt, &nan)
                     : hf_gauss_comp(R->mean[s][c], R->var[s][c], R->weight[s][c], Jw.x, Jw.px, alpha, Jw.bt, &nan);
        }
        s_val[jl][it] = v;
    }
    tab_sync<NT>();
    // ---- the outputs: 8 pairs of E (and of the row of A) and 2*ncol pairs of the component table per row ----
    const int n_out = 8 + 2 * ncol;
    for (int w = tid; w < JOBS * n_out; w += NT) {
        const int jl = w / n_out, o = w - jl * n_out;
        const TableJob Jw = s_job[jl];
        if (!(Jw.flags & 4)) continue;
        const bool first = (Jw.flags & 2) != 0;
        if (o < 8) {
            const int s = o >> 1, p0 = (o & 1) * 2;            // entries (p0, s), (p0 + 1, s): positions HF_PS(p0, s), + 1
            double e[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int pre = p0 + h;
                const int u = (first || (s == 0 && te)) ? 0 : P->umap[pre * 4 + s];
                const int b0 = It.base[s * 4 + u];
                double v;
                if (s < 3) v = s_val[jl][b0];
                else {
                    v = 0.0;
                    for (int c = 0; c < ncol; c++) v += s_val[jl][b0 + c];
                }
                if (first && pre != 0) v = 0.0;
                e[h] = v;
            }
            tab_store2(W.lutE + Jw.row * 16 + HF_PS(p0, s), e[0], e[1]);
            if (W.lutA) {
                const int k = Jw.flags >> 8;
                const DevRegion* __restrict__ R = &P->reg[Jw.r];
                double t[2];
#pragma unroll
                for (int h = 0; h < 2; h++)
                    t[h] = k == 9 ? R->trans[4][s] : (k == 8 ? 1.0 / (HF_NSTATES + 1) : R->tcond[k][(p0 + h) * 4 + s]);
                tab_store2(W.lutA + (int64_t) (job0 + jl) * 16 + HF_PS(p0, s), t[0] * e[0], t[1] * e[1]);
            }
        } else {   // component probabilities per PREVIOUS STATE (the value of its alpha): no select in the consumer
            const int q = o - 8, c = q >> 1, p0 = (q & 1) * 2;
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int u = first ? 0 : P->umap[(p0 + h) * 4 + 3];
                v[h] = first ? 0.0 : s_val[jl][It.base[3 * 4 + u] + c];
            }
            tab_store2(W.lutC + (Jw.row * 4) * W.K + c * 4 + p0, v[0], v[1]);
        }
    }
}

template <int HF_TABLE_JOBS_PER_BLOCK, bool KARG>
__global__ void __launch_bounds__(256) k_tables(const TabWork W, const DevParams* __restrict__ Pg, unsigned* __restrict__ flags,
                                                const std::conditional_t<KARG, KParams, NoKParams> kp, DevParams* __restrict__ P_out) {
    // W.cls / W.lutA (statistics by emission row, hf_seg.h): job j also writes row j of A = (transition table of class cls[j]) ∘ E;
    // `keys` is then the list of the (key, class) pairs that occur — a key with two classes is evaluated twice (same values)
    __shared__ TableJob s_job[HF_TABLE_JOBS_PER_BLOCK];
    __shared__ double s_val[HF_TABLE_JOBS_PER_BLOCK][HF_TABLE_MAX_ITEMS];
    __shared__ int32_t s_base[16];
    __shared__ uint8_t s_item[3][HF_TABLE_MAX_ITEMS];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *flags = 0u;   // first kernel of every pass
    KSTAMP(0);
    // the jobs of the block first: their loads are in flight while the parameter block is rebuilt below — two dependent rounds of
    // global latency were the larger part of this launch-bound kernel's critical path
    const int job0 = blockIdx.x * HF_TABLE_JOBS_PER_BLOCK;
    const TableJob J = tables_load_job<HF_TABLE_JOBS_PER_BLOCK>(W, job0, tid);
    // the parameter block: in global memory (copied before the launch), or — one region, hf_device.h KParams — in the kernel
    // arguments: rebuilt here in LDS, and by block 0 in global memory for the kernels that follow
    __shared__ double s_params[KARG ? sizeof(DevParams) / 8 : 1];
    const DevParams* __restrict__ P;
    if constexpr (KARG) {
        kparams_expand(kp, s_params, tid, 256);
        if (blockIdx.x == 0) kparams_expand(kp, reinterpret_cast<double*>(P_out), tid, 256);
        __syncthreads();
        P = reinterpret_cast<const DevParams*>(s_params);
    } else P = Pg;
    if (tid < 16) s_base[tid] = P->item_base[tid];
    if (tid < P->n_items) { s_item[0][tid] = P->item_s[tid]; s_item[1][tid] = P->item_u[tid]; s_item[2][tid] = P->item_c[tid]; }
    TabItems It; It.base = s_base; It.s = s_item[0]; It.u = s_item[1]; It.c = s_item[2];
    tables_eval<HF_TABLE_JOBS_PER_BLOCK, 256>(W, P, It, job0, tid, J, s_job, s_val);   // (its first barrier also covers s_base / s_item)
}
