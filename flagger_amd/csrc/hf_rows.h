// hf_rows.h — sufficient statistics aggregated by EMISSION ROW instead of by window (HF_ALGO_SCAN, Gaussian /
// truncated-exponential models; the default statistics path of one-GPU runs, hmm_flagger_hip.h HF_STATS_ROWS).
//
// Every statistic the M-step reads is LINEAR in the pair counts  xi_t[pre][s] = f_{t-1}[pre]·T·e·b_t[s] / terminationProb
// (hmm.c:563-650), with coefficients that depend only on the window's emission row — the table row of its key
// (region, x, x_prev), or the private row of a contig-end window (hf_scan.h): the trunc-exp / Gaussian estimator updates
// (hmm_utils.c:812-839, 1027-1034) multiply the count by functions of (x, x_prev, region, component).  So the counts are
// summed per row first (k_pair_sums: 48 multiplications per window, no division, no component loop) and the component
// arithmetic runs once per ROW (k_row_stats: ~8 k rows on BASELINE configs[2] against 1.5 M windows).  Each count is the
// reference's own product in the reference's operand order; only the order of the additions differs from the per-chunk
// path (hf_chunks.h k_stats_tile), i.e. the results agree to rounding (tests: 1e-12 relative against each other, 1e-9
// against the oracle).  The order is fixed by the plan built in hf_create, so a run is reproducible bit for bit.
//
// Plan (static, hf_create): the pairs (t-1, t), t = 2..T-1 of every chunk, sorted by (region, row of A, t); a GROUP is up to
// 64 consecutive pairs of one row of A (= emission row x transition class), their records consecutive in memory; a ROW SLOT
// is up to 4 consecutive groups of one EMISSION row (a popular row has many slots: linearity again), worked on by four lanes
// of k_row_stats; row slots are padded to 64 per region (four wavefronts of 16 slots: the wavefronts of a block always
// belong to one region).
#pragma once
#include <type_traits>
#include "hf_scan.h"

#define HF_GRP_PAIRS 64
#define HF_ROWSLOT_GROUPS 4

// PairIdx, RowSlot: hf_device.h

// ------------------------------------------------------------------------------------------
// k_pair_sums: per group, sum over its pairs of the counts f[pre]·A[pre][s]·b[s] (before the division), A = T∘e = the
// group's row of A (hf_seg.h): the sum of the outer products f ⊗ b is multiplied by A once per group.  (The reference
// multiplies f·T·e·b per pair, hmm.c:612: same value up to the rounding of the order; rounds 1-2 did that per pair.)
// The records of a group are CONSECUTIVE (hf_create: position = group*64 + i; k_seg_fb scatters whole records), so the
// kernel streams: a wavefront takes 4 consecutive groups = 16 KiB, every load instruction reads 1 KiB contiguous = 16 whole
// records, FOUR lanes per record — lane q of a quad holds 16-byte piece q (f01 f23 b01 b23), takes the f piece and the b
// piece of its 2x2 block of the 4x4 count matrix from its quad by DPP, and accumulates the block; the 16 quads are summed by
// a fixed butterfly at the end of every group.  Positions past a group's last pair are not loaded (never written either).
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v) {   // quad_perm within each group of four lanes
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) k_pair_sums(int n_groups, const int32_t* __restrict__ grp_ar, const int32_t* __restrict__ grp_n,
                                                   const double* __restrict__ lutA, const double* __restrict__ recs,
                                                   double* __restrict__ grp_sums) {
    KSTAMP(2);
    const int wave = (int) ((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int g0 = wave * 4;
    if (g0 >= n_groups) return;
    const int ql = lane & 3, qd = lane >> 2;            // piece of the record, record of the instruction
    const int pi = ql & 1, si = ql >> 1;                // this lane's block: pre in {2pi, 2pi+1}, s in {2si, 2si+1}
    const double2* __restrict__ R2 = reinterpret_cast<const double2*>(recs) + (int64_t) g0 * HF_GRP_PAIRS * 4 + lane;
    int ng[4];
#pragma unroll
    for (int g = 0; g < 4; g++) ng[g] = g0 + g < n_groups ? grp_n[g0 + g] : 0;   // (the arrays are padded to a multiple of 4 groups)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        double2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {                    // records j*16 + qd of group g: four 1 KiB loads in flight (all sixteen of the
            const bool have = j * 16 + qd < ng[g];       // wavefront at once was measured 25 % slower, the next group's four issued ahead 15 % slower, same box)
            v[j] = have ? R2[(g * 4 + j) * 64] : make_double2(0.0, 0.0);   // (non-temporal loads: +1.5 .. 2 us, the records come out of the caches k_seg_fb left them in; profiles/r04i_ab_variants.txt)
        }
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // quad_perm [0,1,0,1]: the f piece of this lane's rows; [2,2,3,3]: the b piece of its columns
            const double f0 = quad_perm_f64<0x44>(v[j].x), f1 = quad_perm_f64<0x44>(v[j].y);
            const double b0 = quad_perm_f64<0xFA>(v[j].x), b1 = quad_perm_f64<0xFA>(v[j].y);
            acc[0] += f0 * b0; acc[1] += f1 * b0; acc[2] += f0 * b1; acc[3] += f1 * b1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {                    // the 16 quads, fixed order
            double x = acc[u];
            x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
            acc[u] = x;
        }
        if (qd == 0 && g0 + g < n_groups) {
            const double* __restrict__ A = lutA + (int64_t) grp_ar[g0 + g] * 16;
            double* __restrict__ dst = grp_sums + (int64_t) (g0 + g) * 16;
#pragma unroll
            for (int u = 0; u < 4; u++) {                // state-major position of entry (pre, s) = (2pi + (u & 1), 2si + (u >> 1))
                const int kk = HF_PS(2 * pi + (u & 1), 2 * si + (u >> 1));
                dst[kk] = acc[u] * A[kk];
            }
        }
    }
}

// k_pair_sums_compact: the same sums over a COMPACT plan (hf_create: most rows of A hold a handful of pairs; the groups'
// records lie back to back, group g at grp_off[g]..grp_off[g+1]).  Four lanes per GROUP: the quad walks its group's records in
// order — lane q holds piece q of the record as in k_pair_sums, the same 2x2 block of the count matrix — four records in flight;
// 16 groups per wavefront, the wavefront runs as long as its largest group (<= 64 pairs).  The quad multiplies by its row of A
// and writes its group's 128 bytes.
__global__ void __launch_bounds__(256) k_pair_sums_compact(int n_groups, const int32_t* __restrict__ grp_ar, const int32_t* __restrict__ grp_off,
                                                           const int32_t* __restrict__ grp_n,
                                                           const double* __restrict__ lutA, const double* __restrict__ recs,
                                                           double* __restrict__ grp_sums) {
    const int wave = (int) ((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int ql = lane & 3, g = wave * 16 + (lane >> 2);
    const int pi = ql & 1, si = ql >> 1;
    int base = 0, n = 0;
    if (g < n_groups) { base = grp_off[g]; n = grp_n[g]; }   // (its own count: behind a sub-pass's last group come that sub-pass's spare records, not the next group)
    const double2* __restrict__ R2 = reinterpret_cast<const double2*>(recs) + (int64_t) base * 4 + ql;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; __any(k0 < n); k0 += 4) {
        double2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = k0 + j < n ? R2[(k0 + j) * 4] : make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double f0 = quad_perm_f64<0x44>(v[j].x), f1 = quad_perm_f64<0x44>(v[j].y);
            const double b0 = quad_perm_f64<0xFA>(v[j].x), b1 = quad_perm_f64<0xFA>(v[j].y);
            acc[0] += f0 * b0; acc[1] += f1 * b0; acc[2] += f0 * b1; acc[3] += f1 * b1;
        }
    }
    if (g < n_groups) {
        const double* __restrict__ A = lutA + (int64_t) grp_ar[g] * 16;
        double* __restrict__ dst = grp_sums + (int64_t) g * 16;
#pragma unroll
        for (int u = 0; u < 4; u += 2) {   // entries (2pi, s) and (2pi + 1, s) are adjacent (state-major): one 16-byte store
            const int kk = HF_PS(2 * pi, 2 * si + (u >> 1));
            const double2 a = *reinterpret_cast<const double2*>(A + kk);
            *reinterpret_cast<double2*>(dst + kk) = make_double2(acc[u] * a.x, acc[u + 1] * a.y);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The total of a pass from the block partials of k_row_stats, by the blocks of that same launch (round 3: one launch and one
// kernel boundary less per pass than the separate k_rows_total of rounds 1-2):
//   * the LAST block of every region to finish sums that region's partials (rows_total_region): in plan order by NQ
//     interleaved accumulators per element (fixed by the launch geometry), expanded into the estimator layout of
//     include/hmm_flagger_hip.h exactly as k_chunk_stats does, assembled in LDS and written once — regions in parallel;
//   * the last of the blocks that sum the chunks' log-likelihoods adds those up in k_reduce's order (the same bits as the
//     per-chunk path) and zero-fills the blocks of regions without a row (rows_total_ll);
//   * the last of THOSE parts to finish writes the flag word and, for a host that polls, a per-region checksum word bound
//     to the pass (out[V+2+r]) and the completion stamp (out[V+1]): see wait_total.
// Every hand-off goes through write-through stores, a drained queue and a ticket (xcu_store / xcu_load below).
// Results go to `out_dev` (V + 1 doubles, the flag word last — or to element 0 of `flag_row`, hf_bind_rank_total) and, when
// the context has a pinned host block, to `out_host`.  rw_off[r]..rw_off[r+1]: the wavefronts of region r.
// ------------------------------------------------------------------------------------------
// cross-CU hand-off of a few doubles without cache-wide fences: write-through stores (sc0 sc1), drained by the producer before
// its ticket, and cache-bypassing loads on the reader's side (MI355X_MICROARCH.md, workgroup dispatch & visibility)
__device__ __forceinline__ void xcu_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double xcu_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// scratch of the hand-offs (hf_ctx.d_done): tickets, then the parts' checksum sums and the log-likelihood for the finalizer
#define HF_DONE_REGION0 1                                    // tickets of region r at HF_DONE_REGION0 + r, r = n_regions: the log-likelihood blocks
#define HF_DONE_PARTS (HF_DONE_REGION0 + HF_MAXREGIONS + 1)  // ticket of the finished parts
#define HF_DONE_WORDS (HF_DONE_PARTS + 1)                    // unsigned words; then (8-byte aligned) HF_MAXREGIONS + 1 doubles
__host__ __device__ __forceinline__ double* done_scratch(unsigned* done) { return reinterpret_cast<double*>(done + ((HF_DONE_WORDS + 1) & ~1)); }
#define HF_DONE_BYTES ((((HF_DONE_WORDS + 1) & ~1) * 4) + (HF_MAXREGIONS + 1) * 8)

#ifndef HF_RS_WPB
#define HF_RS_WPB 8    // wavefronts per block of k_row_stats (a multiple of 4: k_row_stats_nb keeps 4): the block that totals a region then reads
#endif                 // 144 partial vectors instead of 288 — one batch of loads per thread instead of three
static_assert(HF_RS_WPB % 4 == 0 && HF_RS_WPB <= 16, "regions are padded to whole blocks of HF_RS_WPB wavefronts; the negative-binomial kernels use 4");
#ifndef HF_ROWS_TOTAL_INFLIGHT
#define HF_ROWS_TOTAL_INFLIGHT 32   // loads a thread of the totalling block keeps in flight (a wavefront counts at most 63 outstanding)
#endif
// one region; returns (thread 0) the checksum sum of what was written
template <int KT>
__device__ unsigned long long rows_total_region(int r, const int32_t* __restrict__ rw_off, int wpb, const double* __restrict__ blk_stats,
                                                const DevParams* __restrict__ P, int Kctx, double* __restrict__ out_dev,
                                                double* __restrict__ out_host) {
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    constexpr int NQMAX = 16;
    const int tid = threadIdx.x, nt = blockDim.x;
    __shared__ double part[NQMAX][NA];
    __shared__ double red[NA];
    __shared__ double blockv[24 * HF_MAXCOMP + 16];   // the region's block of the vector, assembled in LDS
    __shared__ unsigned long long s_x[16];
    const int ncol = P->ncomp[3];
    const bool te = hf_err_is_truncexp(P);
    const int rstride = 24 * Kctx + 16;
    int nq = nt / NA;                              // interleaved accumulators per element: one (element, accumulator) item per thread
    nq = nq < 1 ? 1 : (nq > NQMAX ? NQMAX : nq);
    const int w0 = rw_off[r] / wpb, w1 = rw_off[r + 1] / wpb;   // rw_off counts wavefronts, a multiple of HF_RS_WPB per region
    for (int v = tid; v < rstride; v += nt) blockv[v] = 0.0;
    for (int w = tid; w < nq * NA; w += nt) {
        const int q = w / NA, i = w - q * NA;
        double v = 0.0;
        constexpr int NF = HF_ROWS_TOTAL_INFLIGHT;
        for (int k = w0 + q; k < w1; k += nq * NF) {   // NF loads in flight (the partials come from other CUs: every load is a miss), adds in plan order
            double xk[NF];
#pragma unroll
            for (int u = 0; u < NF; u++) xk[u] = k + nq * u < w1 ? xcu_load(blk_stats + (int64_t) (k + nq * u) * NA + i) : 0.0;
#pragma unroll
            for (int u = 0; u < NF; u++) if (k + nq * u < w1) v += xk[u];
        }
        part[q][i] = v;
    }
    __syncthreads();
    for (int i = tid; i < NA; i += nt) {
        double v = 0.0;
        for (int u = 0; u < nq; u++) v += part[u][i];
        red[i] = v;
    }
    __syncthreads();
    if (w1 > w0 && tid < 64) {
        double* __restrict__ dst = blockv;
        const StatAcc<KT>* __restrict__ Sa = reinterpret_cast<const StatAcc<KT>*>(red);
        if (tid < 16) dst[24 * Kctx + tid] = Sa->trans[tid];
        if (tid == 16 && te) { dst[(0 * 2 + 0) * Kctx] = Sa->te_num; dst[(0 * 2 + 1) * Kctx] = Sa->te_den; }
        if (tid >= 20 && tid < 23) {
            const int s = tid - 20;
            if (!(s == 0 && te)) {
                double* dd = dst + (s * 3) * 2 * Kctx;
                dd[(0 * 2 + 0) * Kctx] = Sa->g_mnum[s]; dd[(0 * 2 + 1) * Kctx] = Sa->g_den[s];
                dd[(1 * 2 + 0) * Kctx] = Sa->g_vnum[s]; dd[(1 * 2 + 1) * Kctx] = Sa->g_den[s];
                dd[(2 * 2 + 0) * Kctx] = Sa->g_den[s];  dd[(2 * 2 + 1) * Kctx] = Sa->g_den[s];
            }
        }
        if (tid >= 32 && tid < 32 + KT && (tid - 32) < ncol) {
            const int cc = tid - 32;
            double* dd = dst + (3 * 3) * 2 * Kctx;
            dd[(0 * 2 + 0) * Kctx + cc] = Sa->c_mnum[cc]; dd[(0 * 2 + 1) * Kctx + cc] = Sa->c_den[cc];
            dd[(1 * 2 + 0) * Kctx + cc] = Sa->c_vnum[cc]; dd[(1 * 2 + 1) * Kctx + cc] = Sa->c_den[cc];
            dd[(2 * 2 + 0) * Kctx + cc] = Sa->c_den[cc];  dd[(2 * 2 + 1) * Kctx + cc] = Sa->c_wden;
        }
    }
    __syncthreads();
    unsigned long long x = 0ull;           // checksum of what is written (hf_cks_term): the host verifies what it read
    for (int v = tid; v < rstride; v += nt) {
        const double dv = blockv[v];
        const int64_t at = 1 + (int64_t) r * rstride + v;
        out_dev[at] = dv;
        if (out_host) out_host[at] = dv;
        x += hf_cks_term((unsigned long long) __double_as_longlong(dv), at);
    }
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((tid & 63) == 0) s_x[tid >> 6] = x;
    __syncthreads();
    unsigned long long c = 0ull;
    if (tid == 0) for (int w = 0; w < (nt >> 6); w++) c += s_x[w];
    return c;
}

// the log-likelihood (element 0) in k_reduce's order over the chunk list, by the block's first wavefront; zeros for the
// regions that have no row at all.  Returns (thread 0) the log-likelihood.
__device__ double rows_total_ll(const int32_t* __restrict__ rw_off, int nreg, int Kctx, const double* __restrict__ chunk_ll, int64_t C,
                                double* __restrict__ out_dev, double* __restrict__ out_host, double* __restrict__ scratch) {
    const int tid = threadIdx.x, nt = blockDim.x;
    double acc = 0.0;
    if (tid < 64) {
        int64_t c = tid;
        for (; c + 64 * 3 < C; c += 64 * 4) {
            const double x0 = xcu_load(chunk_ll + c), x1 = xcu_load(chunk_ll + c + 64), x2 = xcu_load(chunk_ll + c + 128), x3 = xcu_load(chunk_ll + c + 192);
            acc += x0; acc += x1; acc += x2; acc += x3;
        }
        for (; c < C; c += 64) acc += xcu_load(chunk_ll + c);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (tid == 0) { out_dev[0] = acc; if (out_host) out_host[0] = acc; }
    }
    const int rstride = 24 * Kctx + 16;
    for (int r = 0; r < nreg; r++)
        if (rw_off[r + 1] == rw_off[r]) {
            for (int v = tid; v < rstride; v += nt) { const int64_t at = 1 + (int64_t) r * rstride + v; out_dev[at] = 0.0; if (out_host) out_host[at] = 0.0; }
            if (tid == 0) xcu_store(scratch + r, 0.0);       // checksum sum of a block of zeros
        }
    return acc;
}

// ------------------------------------------------------------------------------------------
// k_row_stats: FOUR lanes per row slot — lane p takes the previous state p: its row of the transition counts and its
// term of every estimator update of k_stats_tile (hmm_utils.c:812-839, 1027-1034), with the slot's summed counts in the
// place of one window's counts — 16 row slots of ONE region per wavefront.  The 64 lanes are summed in lane order out of
// LDS (as k_stats_tile does), the wavefronts of a block in wave order: one partial vector per BLOCK, StatAcc<KT> order.
// The blocks after the first n_rw_blocks do a second job that has to happen once per pass anyway: the log-likelihood of
// every chunk (one wavefront per chunk, the same sum as k_chunk_stats) into element 0 of the chunk's vector.
// The last block of every part of the launch to finish then sums the part's partials: rows_total_region / rows_total_ll above.
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(64 * HF_RS_WPB) k_row_stats(int n_rowwaves, int n_rw_blocks, const int32_t* __restrict__ rw_region,
                                                      const RowSlot* __restrict__ slots, const double* __restrict__ grp_sums,
                                                      const RowSrc S, const DevParams* __restrict__ P, double* __restrict__ blk_stats,
                                                      int C, const int32_t* __restrict__ chunk_tile0,
                                                      const double* __restrict__ tile_ll, double* __restrict__ chunk_stats, int64_t V,
                                                      double* __restrict__ chunk_ll, const int32_t* __restrict__ rw_off, int Kctx,
                                                      double* __restrict__ out_dev, double* __restrict__ out_host, double* __restrict__ flag_row,
                                                      const unsigned* __restrict__ flags, double seq, unsigned* __restrict__ done, int n_parts, int bpw,
                                                      double* __restrict__ part_host) {
    // part_host (round 5, the one-GPU path without a polling host): pinned HOST memory [n_rw_blocks][NA] | [C] | [1].  Every block writes its
    // partial vector (a log-likelihood wavefront its chunk's value, block 0 the flag word) there and is done — the HOST sums them after the
    // pass, in the order rows_total_region / rows_total_ll use (hf_estep.hip host_rows_total: the same bits).  Three dependent global round trips
    // (drained partial -> ticket -> the last block's loads) leave the launch's critical path: k_row_stats 13 -> ~6 us.  The partials also stay in
    // blk_stats / chunk_ll, so that a total on the DEVICE can still be had afterwards (k_rows_total_late: hf_rank_total).
    KSTAMP(3);
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    constexpr int NS = 16 + 9 + 2;
    extern __shared__ __attribute__((aligned(16))) double s_rows[];
    const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ bool s_last;
    if ((int) blockIdx.x >= n_rw_blocks) {
        const int c = ((int) blockIdx.x - n_rw_blocks) * wpb + wave;
        if (c < C) {
            const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
            double s = 0.0;
            for (int k = lane; k < nt; k += 64) s += tile_ll[k0 + k];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if (lane == 0) { chunk_stats[(int64_t) c * V] = s; xcu_store(chunk_ll + c, s); if (part_host) part_host[(int64_t) n_rw_blocks * NA + c] = s; }
        }
    } else {
    const int rw = (int) blockIdx.x * wpb + wave;
    const int ncol = P->ncomp[3];
    const bool te = hf_err_is_truncexp(P);
    const int q = lane & 3;                               // the state (column of the counts) this lane of the quad works on
    constexpr int KQ = (KT + 3) / 4;                      // components q, q + 4, .. of the collapsed state
    const DevRegion* __restrict__ R = &P->reg[rw < n_rowwaves ? rw_region[rw] : 0];
    double aq[4];                                         // alpha[p][q], p = 0..3
#pragma unroll
    for (int p = 0; p < 4; p++) aq[p] = P->alpha[p * 4 + q];
    double om[4], o3[4];                                  // 1 - alpha[p][q], 1 - alpha[p][3]
#pragma unroll
    for (int p = 0; p < 4; p++) { om[p] = 1.0 - aq[p]; o3[p] = 1.0 - P->alpha[p * 4 + 3]; }
    double tr[4] = {0.0, 0.0, 0.0, 0.0};                  // trans[p][q]
    double gm = 0.0, gv = 0.0, gd = 0.0, te_num = 0.0, te_den = 0.0, c_wden = 0.0;
    double cm[KQ], cv[KQ], cd[KQ];
#pragma unroll
    for (int j = 0; j < KQ; j++) { cm[j] = 0.0; cv[j] = 0.0; cd[j] = 0.0; }
    // bpw batches of 16 row slots (hf_create: 1 unless the plan is sparse), accumulated in batch order before the one reduction
    for (int bt = 0; bt < bpw; bt++) {
        RowSlot sl; sl.row = -1; sl.g0 = 0; sl.ng = 0; sl.xpx = 0;
        if (rw < n_rowwaves) sl = slots[((int64_t) rw * bpw + bt) * 16 + (lane >> 2)];
        const bool have = sl.row >= 0;
        // column q of the slot's summed counts (rows and tables are state-major: a column is 32 contiguous bytes), groups in plan order
        double cnt[4] = {0.0, 0.0, 0.0, 0.0};
        {
            const double2* __restrict__ gs = reinterpret_cast<const double2*>(grp_sums + (int64_t) sl.g0 * 16 + q * 4);
            double2 g01[HF_ROWSLOT_GROUPS], g23[HF_ROWSLOT_GROUPS];
#pragma unroll
            for (int g = 0; g < HF_ROWSLOT_GROUPS; g++) {
                const bool on = have && g < sl.ng;
                g01[g] = on ? gs[g * 8] : make_double2(0.0, 0.0);
                g23[g] = on ? gs[g * 8 + 1] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int g = 0; g < HF_ROWSLOT_GROUPS; g++)
                if (g < sl.ng) { cnt[0] += g01[g].x; cnt[1] += g01[g].y; cnt[2] += g23[g].x; cnt[3] += g23[g].y; }
        }
        double Eq[4] = {1.0, 1.0, 1.0, 1.0};
        if (have) {
            const double2* __restrict__ er = reinterpret_cast<const double2*>(S.lutE + (int64_t) sl.row * 16 + q * 4);
            const double2 e01 = er[0], e23 = er[1];
            Eq[0] = e01.x; Eq[1] = e01.y; Eq[2] = e23.x; Eq[3] = e23.y;
        }
        const double x = (double) (sl.xpx & 0xff), px = (double) ((sl.xpx >> 8) & 0xff);
        double adj[4], xa[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            adj[p] = cnt[p] / HF_TERMINATION_PROB;                          // hmm.c:613-614
            xa[p] = aq[p] == 0.0 ? x : (x - aq[p] * px) / om[p];            // hmm_utils.c:812-839
        }
        if (have) {
#pragma unroll
            for (int p = 0; p < 4; p++) tr[p] += adj[p];                    // hmm_utils.c:2010-2015
            if (q == 0 && te) {                                             // hmm_utils.c:1027-1034
#pragma unroll
                for (int p = 0; p < 4; p++) { te_num += adj[p] * x; te_den += adj[p]; }
            } else if (q < 3) {                                             // one component
                const double mu = R->mean[q][0];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const double w = adj[p] * Eq[p] / Eq[p];
                    gm += w * xa[p];
                    const double z = (xa[p] - mu) * om[p];
                    gv += w * z * z;
                    gd += w;
                }
            }
        }
        // the collapsed state: its counts, adjusted coverages and emission values are lane 3's; every lane of the quad takes
        // components q, q + 4, ..: one 32-byte row of the component table [component][previous state] each
        double a3[4], x3[4], E3[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            a3[p] = quad_perm_f64<0xFF>(adj[p]); x3[p] = quad_perm_f64<0xFF>(xa[p]); E3[p] = quad_perm_f64<0xFF>(Eq[p]);
        }
        if (have) {
            const double2* __restrict__ crow = reinterpret_cast<const double2*>(S.lutC + ((int64_t) sl.row * 4) * S.K);
#pragma unroll
            for (int j = 0; j < KQ; j++) {
                const int cc = q + 4 * j;
                if (cc >= ncol) continue;
                const double2 u01 = crow[cc * 2], u23 = crow[cc * 2 + 1];
                const double pc[4] = {u01.x, u01.y, u23.x, u23.y};
                const double mu = R->mean[3][cc];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const double w = a3[p] * pc[p] / E3[p];
                    cm[j] += w * x3[p];
                    const double z = (x3[p] - mu) * o3[p];
                    cv[j] += w * z * z;
                    cd[j] += w;
                    c_wden += w;
                }
            }
        }
    }
    // the 16 quads of the wavefront: a fixed butterfly (lanes 0..3 end up with the sums of their state / their components)
    double* __restrict__ s_blk = s_rows;                  // [wpb][NA] wave sums, StatAcc<KT> order
    double* __restrict__ wsum = s_blk + wave * NA;
    auto quads = [](double v) { v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };
#pragma unroll
    for (int p = 0; p < 4; p++) tr[p] = quads(tr[p]);
    gm = quads(gm); gv = quads(gv); gd = quads(gd); te_num = quads(te_num); te_den = quads(te_den); c_wden = quads(c_wden);
#pragma unroll
    for (int j = 0; j < KQ; j++) { cm[j] = quads(cm[j]); cv[j] = quads(cv[j]); cd[j] = quads(cd[j]); }
    {
        const double w1 = __shfl(c_wden, 1), w2 = __shfl(c_wden, 2), w3 = __shfl(c_wden, 3);
        if (lane == 0) wsum[NS + 3 * KT] = ((c_wden + w1) + w2) + w3;
    }
    if (lane < 4) {
#pragma unroll
        for (int p = 0; p < 4; p++) wsum[p * 4 + q] = tr[p];
        if (q < 3) { wsum[16 + q] = gm; wsum[19 + q] = gv; wsum[22 + q] = gd; }
        if (q == 0) { wsum[25] = te_num; wsum[26] = te_den; }
#pragma unroll
        for (int j = 0; j < KQ; j++) {
            const int cc = q + 4 * j;
            if (cc < KT) { wsum[NS + cc] = cm[j]; wsum[NS + KT + cc] = cv[j]; wsum[NS + 2 * KT + cc] = cd[j]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA; i += blockDim.x) {   // the block's wavefronts in wave order
        double v = 0.0;
        for (int w = 0; w < wpb; w++) v += s_blk[w * NA + i];
        xcu_store(blk_stats + (int64_t) blockIdx.x * NA + i, v);
        if (part_host) part_host[(int64_t) blockIdx.x * NA + i] = v;
    }
    }
    if (part_host) {
        if (blockIdx.x == 0 && threadIdx.x == 0) part_host[(int64_t) n_rw_blocks * NA + C] = (double) (flags ? *flags : 0u);   // (final before this launch: k_seg_fb raised it)
        return;
    }
    // ---- hand-offs: the last block of a part (a region / the log-likelihood blocks) sums the part; the last part writes flag
    // word, checksums and stamp (producer: write-through stores, drained, block barrier, ticket; consumer: cache-bypassing loads) ----
    const int nreg = P->n_regions;
    const int part_id = (int) blockIdx.x < n_rw_blocks ? rw_region[(int) blockIdx.x * wpb] : nreg;
    const unsigned part_blocks = part_id < nreg ? (unsigned) ((rw_off[part_id + 1] - rw_off[part_id]) / wpb) : gridDim.x - (unsigned) n_rw_blocks;
    double* __restrict__ scratch = done_scratch(done);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(done + HF_DONE_REGION0 + part_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == part_blocks - 1;
        if (s_last) done[HF_DONE_REGION0 + part_id] = 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (part_id < nreg) {
        const unsigned long long x = rows_total_region<KT>(part_id, rw_off, wpb, blk_stats, P, Kctx, out_dev, out_host);
        if (seq == 0.0) return;
        if (threadIdx.x == 0) xcu_store(scratch + part_id, __longlong_as_double((long long) x));
    } else {
        const double ll = rows_total_ll(rw_off, nreg, Kctx, chunk_ll, (int64_t) C, out_dev, out_host, scratch);
        if (seq == 0.0) {
            // No host polls the block (the default: completion is the stream's own stamp, hf_estep.hip wait_total): the parts need no common
            // finish.  The flag word is final before this launch starts (k_seg_fb raised it), so this part writes it — one ticket round trip, one
            // load and one store less on the launch's critical path than the round-3/4 "last part" stage (k_row_stats 13.3 -> ~11 us).
            if (threadIdx.x == 0 && flags) {
                const unsigned fl = *flags;
                if (flag_row) flag_row[0] = (double) fl; else out_dev[V] = (double) fl;
                if (out_host) out_host[V] = (double) fl;
            }
            return;
        }
        if (threadIdx.x == 0) xcu_store(scratch + HF_MAXREGIONS, ll);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(done + HF_DONE_PARTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == (unsigned) n_parts - 1;
        if (s_last) done[HF_DONE_PARTS] = 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        const unsigned fl = flags ? *flags : 0u;
        if (flags) {   // the flag word: element V of the vector, or element 0 of a row of an exchange buffer (hf_bind_rank_total)
            if (flag_row) flag_row[0] = (double) fl; else out_dev[V] = (double) fl;
            if (out_host) out_host[V] = (double) fl;
        }
        if (seq != 0.0 && out_host) {   // for a host that polls the pinned block: checksums bound to the pass, then the stamp
            const double ll = xcu_load(scratch + HF_MAXREGIONS);
            for (int r = 0; r < nreg; r++) {
                unsigned long long c = (unsigned long long) __double_as_longlong(seq) + (unsigned long long) __double_as_longlong(xcu_load(scratch + r));
                if (r == 0) c += hf_cks_term((unsigned long long) __double_as_longlong(ll), 0) +
                                 hf_cks_term((unsigned long long) __double_as_longlong((double) fl), V);
                out_host[V + 2 + r] = __longlong_as_double((long long) c);
            }
            __threadfence_system();
            out_host[V + 1] = seq;
            __threadfence_system();
        }
    }
}

// the total on the DEVICE after a pass whose partials went to the host (part_host): one block per region + one for the log-likelihood
template <int KT>
__global__ void __launch_bounds__(512) k_rows_total_late(const int32_t* __restrict__ rw_off, int wpb, const double* __restrict__ blk_stats,
                                                         const DevParams* __restrict__ P, int Kctx, const double* __restrict__ chunk_ll, int64_t C,
                                                         double* __restrict__ out_dev, double* __restrict__ scratch) {
    const int nreg = P->n_regions;
    if ((int) blockIdx.x < nreg) (void) rows_total_region<KT>((int) blockIdx.x, rw_off, wpb, blk_stats, P, Kctx, out_dev, nullptr);
    else (void) rows_total_ll(rw_off, nreg, Kctx, chunk_ll, C, out_dev, nullptr, scratch);
}

// what ranks exchange (hf_rank_total): the total the pass left on the device, without the flag word
__global__ void k_copy_total(const double* __restrict__ src, double* __restrict__ dst, int64_t V) {
    for (int64_t v = threadIdx.x; v < V; v += blockDim.x) dst[v] = src[v];
}

