// hf_chunks.h — per-chunk sufficient statistics (HF_STATS_CHUNKS; the only statistics path of HF_ALGO_SEQ): one estimator
// vector per chunk as EM_runOneIterationForList keeps one EM object per chunk (hmm.c:739-763), reduced over the chunk list in
// list order.  k_stats_tile -> k_chunk_stats -> k_reduce.  The default of one-GPU runs is hf_rows.h.
#pragma once
#include "hf_scan.h"

// Round 6: the per-window arithmetic of k_stats_tile without its IEEE divisions where a prepared reciprocal (divp: within an ulp of the quotient)
// or no division at all gives the same value to rounding — the kernel is bound by exactly this arithmetic (profiles/r05_chunks_path.txt), and
// neither statistics path was ever bit-identical to a sequential run.  -DHF_CHUNKS_FASTDIV=0: the divisions of rounds 1-5.
#ifndef HF_CHUNKS_FASTDIV
#define HF_CHUNKS_FASTDIV 1
#endif

// ------------------------------------------------------------------------------------------
// xi sufficient statistics (A6, A12).  For every pair (i, i+1), i = 1..T-2:
//   xi = f_i[pre] * T * e * b_{i+1}[s] / terminationProb              (hmm.c:563-650)
// Distinct accumulators only (mean.den == var.den == weight.num; weight.den[i] all equal); they are
// expanded into the reference's estimator layout when the chunk vector is written (k_chunk_stats).
// ------------------------------------------------------------------------------------------
template <int KT>
struct StatAcc {
    double trans[16];
    double g_mnum[3], g_vnum[3], g_den[3];      // single-component Gaussian states 0 (Err, gaussian model), 1, 2
    double te_num, te_den;                      // trunc-exp Err
    double c_mnum[KT], c_vnum[KT], c_den[KT], c_wden; // Col components
};

// ------------------------------------------------------------------------------------------
// k_stats_tile: statistics of the pairs of one tile per wavefront; lane l owns the pairs that END at its L windows.
// f, b come from the pass arrays, T from the LDS tables, the emission row and the collapsed state's component
// probabilities from this iteration's rows (k_tables: the table row of (region, x, x_prev), or the window's private
// row at contig ends); the total of the component probabilities is the emission value itself (Ev[pre][Col] is
// the same sum of the same terms).  The 3K per-component accumulators of a lane live in LDS (lane-minor,
// conflict-free) so the component loop stays rolled and the kernel keeps its occupancy; the 27 scalar
// accumulators stay in registers.
// ------------------------------------------------------------------------------------------
struct StatAccSmall {
    double trans[16];
    double g_mnum[3], g_vnum[3], g_den[3];
    double te_num, te_den;
};

template <int KT>
__global__ void __launch_bounds__(256, 2) k_stats_tile(int ntiles, const TileDesc* __restrict__ td,
                                                    const uint32_t* __restrict__ rec, const RowSrc S,
                                                    const DevParams* __restrict__ P,
                                                    const double* __restrict__ F, const double* __restrict__ B,
                                                    const uint64_t* __restrict__ regmask,
                                                    double* __restrict__ tile_stats, const int32_t* __restrict__ slot_of) {
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    constexpr int NS = 16 + 9 + 2;                  // scalar accumulators kept in registers
    constexpr int L = HF_SCAN_L;
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wave;   // 4 wavefronts per block unless LDS forces fewer
    if (tile >= ntiles) return;
    // per-wave accumulator rows [row][lane], row stride 65 doubles: lane-minor accesses and the column sums at the
    // end of a region are both bank-conflict free.  Rows 0..3*ncol-1: component accumulators (q*ncol + cc).
    constexpr int RS = 65;
    const int nrows = 3 * P->ncomp[3] > NS + 1 ? 3 * P->ncomp[3] : NS + 1;
    double* __restrict__ s_row = s_tab + P->n_regions * HF_TAB_STRIDE + wave * (nrows * RS);
    double* __restrict__ s_acc = s_row + lane;
    const TileDesc d = td[tile];
    const int64_t t0 = d.t0, T = d.T, base = d.base;
    const bool te = hf_err_is_truncexp(P);
    const int ncol = P->ncomp[3], nreg = P->n_regions;
    const int64_t a0 = base + (int64_t) lane * L;   // this lane owns windows a0..a0+L-1 and the pairs ending there
    uint32_t rr[L + 1];                             // rr[j+1] = window a0+j, rr[0] = the window before
    rr[0] = load_recs<L>(rec, t0, T, a0, lane, rr + 1);
    int sidx[L];
    tile_slow_index<L>(rr + 1, lane, d.slow0, sidx);
    bool ok[L];
    unsigned long long present = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const int64_t w = a0 + j;                             // pair (w-1, w), w = 2..T-1  (hmm.c:638-642)
        ok[j] = w >= 2 && w <= T - 1;
        if (ok[j]) present |= 1ull << (REC_REGION(rr[j + 1]) & 63u);
    }
    for (int o = 32; o > 0; o >>= 1) present |= __shfl_xor(present, o);
    const unsigned long long in_chunk = regmask[d.chunk];
    double xa[4], om[4];
    const PreDiv term_div = prediv(HF_TERMINATION_PROB);
    for (int r = 0; r < nreg; r++) {
        if (!((in_chunk >> r) & 1ull)) continue;   // k_chunk_stats never reads this slot
        double* __restrict__ dst = tile_stats + ((int64_t) tile * nreg + r) * NA;
        if (!((present >> r) & 1ull)) {            // region occurs in the chunk but not in this tile
            for (int i = lane; i < NA; i += 64) dst[i] = 0.0;
            continue;
        }
        const DevRegion* __restrict__ R = &P->reg[r];
        StatAccSmall a;
#pragma unroll
        for (int i = 0; i < NS; i++) reinterpret_cast<double*>(&a)[i] = 0.0;
        double c_wden = 0.0;
        for (int i = 0; i < 3 * ncol; i++) s_acc[i * RS] = 0.0;
#pragma unroll 1
        for (int j = 0; j < L; j++) {
            if (!(ok[j] && (int) REC_REGION(rr[j + 1]) == r)) continue;
            double Ev[16], Tm[16], f[4], b1[4];
            load_row(row_ptr(S, rr[j + 1], rr[j], sidx[j]), Ev);
            // f of the window before (the previous lane's last one for j == 0), b of the window itself (hf_scan.h fb_slot)
            double2 f01, f23, b01, b23;
            if (slot_of) {      // the pair record {f_{t-1}, b_t} of the segment kernels (hf_seg.h), by slot; F = the records
                const double2* __restrict__ pr = reinterpret_cast<const double2*>(F) + (int64_t) slot_of[t0 + a0 + j] * 4;
                f01 = pr[0]; f23 = pr[1]; b01 = pr[2]; b23 = pr[3];
            } else {            // HF_ALGO_SEQ: f, b tile-major / lane-minor (hf_seq.h)
                const int64_t fs = j > 0 ? fb_slot<L>(tile, lane, j - 1, 0)
                                         : (lane > 0 ? fb_slot<L>(tile, lane - 1, L - 1, 0) : fb_slot<L>(tile - 1, 63, L - 1, 0));
                const int64_t bs = fb_slot<L>(tile, lane, j, 0);
                f01 = reinterpret_cast<const double2*>(F)[fs]; f23 = reinterpret_cast<const double2*>(F)[fs + 64];
                b01 = reinterpret_cast<const double2*>(B)[bs]; b23 = reinterpret_cast<const double2*>(B)[bs + 64];
            }
            const unsigned xw = REC_X(rr[j + 1]), xp = REC_X(rr[j]);
            const double2* __restrict__ crow = crow_ptr(S, rr[j + 1], rr[j], sidx[j]);
            lds_Tm(s_tab, rr[j + 1], Tm);
            f[0] = f01.x; f[1] = f01.y; f[2] = f23.x; f[3] = f23.y;
            b1[0] = b01.x; b1[1] = b01.y; b1[2] = b23.x; b1[3] = b23.y;
            const double x = (double) xw, px = (double) xp;
            // one state (column) at a time, with a scheduling barrier after each: keeps the live set small;
            // state outer / pre inner is also the order of the reference (hmm.c:588-589)
            double adj3[4];
            bool col_fast = false;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                double adj[4];
                bool okc = true;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int k = HF_PS(p, s);                // rows and tables are state-major
                    adj[p] = f[p] * Tm[k] * Ev[k] * b1[s];    // count, hmm.c:612
                    okc &= div_operand_safe(adj[p]);
                }
                if (s == 3) {
#pragma unroll
                    for (int p = 0; p < 4; p++) okc &= Ev[HF_PS(p, 3)] != 0.0 && div_operand_safe(Ev[HF_PS(p, 3)]);
                }
                // count / terminationProb (hmm.c:613-614): the shared-denominator form when every operand is in range
                if (__all(okc)) {
#pragma unroll
                    for (int p = 0; p < 4; p++) adj[p] = divp(adj[p], term_div);
                    if (s == 3) col_fast = true;
                } else {
#pragma unroll
                    for (int p = 0; p < 4; p++) adj[p] = adj[p] / HF_TERMINATION_PROB;
                }
#pragma unroll
                for (int p = 0; p < 4; p++) a.trans[p * 4 + s] += adj[p];   // hmm_utils.c:2010-2015
                if (s == 3) {
#pragma unroll
                    for (int p = 0; p < 4; p++) adj3[p] = adj[p];
                } else if (s == 0 && te) {                    // hmm_utils.c:1027-1034
#pragma unroll
                    for (int p = 0; p < 4; p++) { a.te_num += adj[p] * x; a.te_den += adj[p]; }
                } else {                                      // hmm_utils.c:812-839, one component
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int k = HF_PS(p, s);
                        const double alpha = P->alpha[p * 4 + s];
                        // alpha == 0 (wave-uniform): (x - 0*px) / (1 - 0) is x itself
#if HF_CHUNKS_FASTDIV
                        // (1 - alpha is wave-uniform and the same for every window: its reciprocal is prepared once — the compiler hoists it out of the
                        // window loop; count * prob / totProb of a one-component state is the count itself unless the emission value is zero, where
                        // the reference's 0 * 0 / 0 is a NaN: hmm_utils.c:812-839)
                        const double x_adj = alpha == 0.0 ? x : divp(x - alpha * px, prediv(1.0 - alpha));
                        const double w = Ev[k] == 0.0 ? __builtin_nan("") : adj[p];
#else
                        const double x_adj = alpha == 0.0 ? x : (x - alpha * px) / (1.0 - alpha);
                        const double w = adj[p] * Ev[k] / Ev[k];
#endif
                        a.g_mnum[s] += w * x_adj;
                        const double z = (x_adj - R->mean[s][0]) * (1.0 - alpha);
                        a.g_vnum[s] += w * z * z;
                        a.g_den[s] += w;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // collapsed state, component-major: one 32-byte row per component serves all four pre states
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const double alpha = P->alpha[p * 4 + 3];
#if HF_CHUNKS_FASTDIV
                xa[p] = alpha == 0.0 ? x : divp(x - alpha * px, prediv(1.0 - alpha));
#else
                xa[p] = alpha == 0.0 ? x : (x - alpha * px) / (1.0 - alpha);
#endif
                om[p] = 1.0 - alpha;
            }
            if (col_fast) {   // w = adj3 * pc / E3 with the four reciprocals of E3 prepared once for all components
                PreDiv e3[4];
#pragma unroll
                for (int p = 0; p < 4; p++) e3[p] = prediv(Ev[HF_PS(p, 3)]);
#pragma unroll 1
                for (int cc = 0; cc < ncol; cc++) {
                    const double2 u01 = crow[cc * 2], u23 = crow[cc * 2 + 1];
                    const double mu = R->mean[3][cc];
                    double mnum = s_acc[cc * RS], vnum = s_acc[(ncol + cc) * RS], den = s_acc[(2 * ncol + cc) * RS];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const double pc = p == 0 ? u01.x : p == 1 ? u01.y : p == 2 ? u23.x : u23.y;   // [component][previous state]
                        const double w = divp(adj3[p] * pc, e3[p]);
                        mnum += w * xa[p];
                        const double z = (xa[p] - mu) * om[p];
                        vnum += w * z * z;
                        den += w;
                        c_wden += w;
                    }
                    s_acc[cc * RS] = mnum; s_acc[(ncol + cc) * RS] = vnum; s_acc[(2 * ncol + cc) * RS] = den;
                }
            } else {
#pragma unroll 1
                for (int cc = 0; cc < ncol; cc++) {
                    const double2 u01 = crow[cc * 2], u23 = crow[cc * 2 + 1];
                    const double mu = R->mean[3][cc];
                    double mnum = s_acc[cc * RS], vnum = s_acc[(ncol + cc) * RS], den = s_acc[(2 * ncol + cc) * RS];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const double pc = p == 0 ? u01.x : p == 1 ? u01.y : p == 2 ? u23.x : u23.y;
                        const double w = adj3[p] * pc / Ev[HF_PS(p, 3)];
                        mnum += w * xa[p];
                        const double z = (xa[p] - mu) * om[p];
                        vnum += w * z * z;
                        den += w;
                        c_wden += w;
                    }
                    s_acc[cc * RS] = mnum; s_acc[(ncol + cc) * RS] = vnum; s_acc[(2 * ncol + cc) * RS] = den;
                }
            }
        }
        // sum over the 64 lanes in lane order: accumulator i is summed by lane i out of its LDS row (fixed order),
        // results stored in StatAcc<KT> order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            double v = 0.0;
            if (lane < 3 * ncol) {
                const double* __restrict__ row = s_row + lane * RS;
#pragma unroll 8
                for (int l = 0; l < 64; l++) v += row[l];
            }
            for (int i = lane; i < 3 * KT; i += 64) dst[NS + i] = 0.0;       // slots of components >= ncol
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 3 * ncol) dst[NS + (lane / ncol) * KT + (lane % ncol)] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < NS; i++) s_acc[i * RS] = reinterpret_cast<double*>(&a)[i];
        s_acc[NS * RS] = c_wden;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane <= NS) {
            const double* __restrict__ row = s_row + lane * RS;
            double v = 0.0;
#pragma unroll 8
            for (int l = 0; l < 64; l++) v += row[l];
            dst[lane < NS ? lane : NS + 3 * KT] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// per chunk: log-likelihood = sum of the tiles' partial sums; statistics = sum of the tile partials in tile order,
// expanded into the estimator layout of include/hmm_flagger_hip.h (mean.den == var.den == weight.num;
// weight.den[i] all equal).  Writes the WHOLE chunk vector (zeros where nothing accumulates:
// HMM_resetEstimators, hmm.c:129-134), so no memset is needed between passes.  full == 0: log-likelihood only.
template <int KT>
__global__ void __launch_bounds__(128) k_chunk_stats(const int32_t* __restrict__ chunk_tile0, const int32_t* __restrict__ chunk_ll0,
                                                     const uint64_t* __restrict__ regmask,
                                                     const double* __restrict__ tile_stats, const double* __restrict__ tile_ll,
                                                     const DevParams* __restrict__ P, double* __restrict__ chunk_stats,
                                                     int64_t V, int Kctx, int full) {
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
    const int nreg = P->n_regions, ncol = P->ncomp[3];
    const bool te = hf_err_is_truncexp(P);
    const int64_t rstride = 24 * (int64_t) Kctx + 16;
    const uint64_t present = regmask[c];
    double* __restrict__ vec = chunk_stats + (int64_t) c * V;
    if (tid < 64) {      // the log-likelihood partials of the chunk: per segment (hf_seg.h) or per tile (HF_ALGO_SEQ)
        double s = 0.0;
        const int l0 = chunk_ll0[c], nl = chunk_ll0[c + 1] - l0;
        for (int k = tid; k < nl; k += 64) s += tile_ll[l0 + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (tid == 0) vec[0] = s;
    }
    for (int64_t v = 1 + tid; v < V; v += blockDim.x) vec[v] = 0.0;
    if (!full) return;
    __shared__ double red[NA];
    __syncthreads();
    for (int r = 0; r < nreg; r++) {
        if (!((present >> r) & 1ull)) continue;
        if (tid < NA) {
            double v = 0.0;
            int k = 0;
            for (; k + 8 <= nt; k += 8) {   // 8 loads in flight, adds stay in tile order
                double xk[8];
#pragma unroll
                for (int u = 0; u < 8; u++) xk[u] = tile_stats[((int64_t) (k0 + k + u) * nreg + r) * NA + tid];
#pragma unroll
                for (int u = 0; u < 8; u++) v += xk[u];
            }
            for (; k < nt; k++) v += tile_stats[((int64_t) (k0 + k) * nreg + r) * NA + tid];
            red[tid] = v;
        }
        __syncthreads();
        double* __restrict__ dst = vec + 1 + r * rstride;
        const StatAcc<KT>* __restrict__ Sa = reinterpret_cast<const StatAcc<KT>*>(red);
        if (tid < 16) dst[24 * Kctx + tid] = Sa->trans[tid];
        if (tid == 32 && te) { dst[(0 * 2 + 0) * Kctx] = Sa->te_num; dst[(0 * 2 + 1) * Kctx] = Sa->te_den; }
        if (tid >= 64 && tid < 67) {
            const int s = tid - 64;
            if (!(s == 0 && te)) {
                double* dd = dst + (int64_t) (s * 3) * 2 * Kctx;
                dd[(0 * 2 + 0) * Kctx] = Sa->g_mnum[s]; dd[(0 * 2 + 1) * Kctx] = Sa->g_den[s];
                dd[(1 * 2 + 0) * Kctx] = Sa->g_vnum[s]; dd[(1 * 2 + 1) * Kctx] = Sa->g_den[s];
                dd[(2 * 2 + 0) * Kctx] = Sa->g_den[s];  dd[(2 * 2 + 1) * Kctx] = Sa->g_den[s];
            }
        }
        if (tid >= 96 && tid < 96 + KT && (tid - 96) < ncol) {
            const int cc = tid - 96;
            double* dd = dst + (int64_t) (3 * 3) * 2 * Kctx;
            dd[(0 * 2 + 0) * Kctx + cc] = Sa->c_mnum[cc]; dd[(0 * 2 + 1) * Kctx + cc] = Sa->c_den[cc];
            dd[(1 * 2 + 0) * Kctx + cc] = Sa->c_vnum[cc]; dd[(1 * 2 + 1) * Kctx + cc] = Sa->c_den[cc];
            dd[(2 * 2 + 0) * Kctx + cc] = Sa->c_den[cc];  dd[(2 * 2 + 1) * Kctx + cc] = Sa->c_wden;
        }
        __syncthreads();
    }
}

// sum over chunks (hmm.c:759-763) in a fixed order that depends only on the chunk list: one wavefront per vector
// element, lane l adds chunks l, l+64, ... in list order, then a fixed shuffle tree over the lanes.  The same
// kernel reduces the local chunk list on one GPU and the all-gathered list on N GPUs => identical bits.
// Error flags ride along as element V: this context's flag word, OR-ed with the flag rows of an exchange buffer
// (n_ranks > 0: rank k's word is element 0 of row k*rows_per_rank + flag_row, written by k_flag_row) so that every rank
// of a multi-GPU pass reports the same error.
// seq != 0: `out` is the pinned host block and the host polls out[V+1]: the block that finishes last stamps it.
__global__ void __launch_bounds__(64) k_reduce(const double* __restrict__ chunk_stats, const int32_t* __restrict__ row_index,
                                               int64_t n_chunks, int64_t V, double* __restrict__ out,
                                               const unsigned* __restrict__ flags, int n_ranks, int rows_per_rank, int flag_row,
                                               double seq, unsigned* __restrict__ done, unsigned long long* __restrict__ cks) {
    const int64_t v = blockIdx.x;
    const int lane = threadIdx.x;
    double written = 0.0;
    if (v == V) {
        if (flags) {
            unsigned fl = lane == 0 ? *flags : 0u;
            for (int k = lane; k < n_ranks; k += 64) fl |= (unsigned) chunk_stats[((int64_t) k * rows_per_rank + flag_row) * V];
            for (int o = 32; o > 0; o >>= 1) fl |= __shfl_xor(fl, o);
            written = (double) fl;
            if (lane == 0) out[V] = written;
        }
    } else {
        double acc = 0.0;
        // row_index (multi-GPU): row of global chunk c inside the all-gathered, per-rank padded buffer
        for (int64_t c = lane; c < n_chunks; c += 64) acc += chunk_stats[(row_index ? (int64_t) row_index[c] : c) * V + v];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) out[v] = acc;
        written = acc;
    }
    if (seq != 0.0 && lane == 0) {   // checksum of every word written (hf_cks_term), bound to the pass, then the stamp: see wait_total
        atomicAdd(cks, hf_cks_term((unsigned long long) __double_as_longlong(written), v));
        __threadfence_system();
        if (atomicAdd(done, 1u) == gridDim.x - 1) {
            const unsigned long long c = atomicAdd(cks, 0ull) + (unsigned long long) __double_as_longlong(seq);
            *cks = 0ull; *done = 0u;
            out[V + 2] = __longlong_as_double((long long) c);
            out[V + 1] = seq;
            __threadfence_system();
        }
    }
}

// this rank's error-flag word as element 0 of a row of the exchange buffer (the rest of the row is not read)
__global__ void k_flag_row(const unsigned* __restrict__ flags, double* __restrict__ row) {
    if (threadIdx.x == 0) row[0] = (double) *flags;
}
