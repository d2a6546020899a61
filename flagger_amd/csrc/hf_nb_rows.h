// hf_nb_rows.h — statistics by emission row for the negative_binomial model (HF_STATS_ROWS, hf_rows.h).
//
// The model's statistics go through per-state count data hist[state][min(x, 249)] (hmm.c:615-617, count_data.c:49-57) that
// NegativeBinomial_updateEstimator turns into estimator increments (hmm_utils.c:537-566, 1661-1673); the increments are
// linear in the counts, so one update applied to the sum of every chunk's count data equals the sum of the per-chunk
// updates up to rounding.  The pair counts come from the same k_pair_sums as the Gaussian models (the emission rows hold
// E[x] in every previous-state row: k_tables_nb); then
//   k_row_stats_nb   per row slot: division by the termination probability, transition counts (block partials) and the
//                    slot's contribution to the count data of its (region, x): the sum over previous states, per state
//   k_nb_hist        one wavefront per (region, bin): the slots of the bin in plan order (a static list of hf_create)
//   k_nb_total       one block per region: transition counts, the estimator increments of every (state, component) from
//                    the region's count data, the log-likelihood; published like the total of k_row_stats (checksum + stamp)
#pragma once
#include "hf_rows.h"
#include "hf_nb.h"

// ------------------------------------------------------------------------------------------
// k_row_stats_nb: four lanes per row slot (lane p = previous state p), 16 slots of one region per wavefront — the layout
// of k_row_stats.  slot_h[slot][s] = sum_p counts[p][s] / terminationProb; the 16 transition counts are summed over the
// wavefront's lanes in lane order, over the block's wavefronts in wave order: blk_trans[block][16] ([pre*4 + s]).
// The blocks after the first n_rw_blocks sum the chunks' log-likelihoods as in k_row_stats.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_row_stats_nb(int n_rowwaves, int n_rw_blocks, const RowSlot* __restrict__ slots,
                                                      const double* __restrict__ grp_sums, double* __restrict__ slot_h,
                                                      double* __restrict__ blk_trans, int C, const int32_t* __restrict__ chunk_tile0,
                                                      const double* __restrict__ tile_ll, double* __restrict__ chunk_stats, int64_t V,
                                                      double* __restrict__ chunk_ll, int bpw) {
    __shared__ double s_rows[4][16 * 65];
    __shared__ double s_blk[4][16];
    const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int) blockIdx.x >= n_rw_blocks) {
        const int c = ((int) blockIdx.x - n_rw_blocks) * wpb + wave;
        if (c >= C) return;
        const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
        double s = 0.0;
        for (int k = lane; k < nt; k += 64) s += tile_ll[k0 + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) { chunk_stats[(int64_t) c * V] = s; chunk_ll[c] = s; }
        return;
    }
    const int rw = (int) blockIdx.x * wpb + wave;
    const int p = lane & 3;
    double tr[4] = {0.0, 0.0, 0.0, 0.0};          // the wavefront's transition counts: bpw batches of 16 slots in batch order
    for (int bt = 0; bt < bpw; bt++) {
    RowSlot sl; sl.row = -1; sl.g0 = 0; sl.ng = 0; sl.xpx = 0;
    const int64_t slot = ((int64_t) rw * bpw + bt) * 16 + (lane >> 2);
    if (rw < n_rowwaves) sl = slots[slot];
    double ts[4] = {0.0, 0.0, 0.0, 0.0};          // this slot's
    if (sl.row >= 0) {
        const double* __restrict__ gs = grp_sums + (int64_t) sl.g0 * 16 + p;
        double gv[HF_ROWSLOT_GROUPS][4];
#pragma unroll
        for (int g = 0; g < HF_ROWSLOT_GROUPS; g++)
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) gv[g][s4] = gs[(int64_t) (g < sl.ng ? g : 0) * 16 + s4 * 4];
        double cnt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < HF_ROWSLOT_GROUPS; g++)
            if (g < sl.ng) {
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) cnt[s4] += gv[g][s4];   // groups in plan order
            }
#pragma unroll
        for (int s = 0; s < 4; s++) { ts[s] = cnt[s] / HF_TERMINATION_PROB; tr[s] += ts[s]; }   // hmm.c:613-614
    }
    {   // the slot's count data: previous states in index order (hmm.c:588-589), by the lane of previous state 0
        double h[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const double a1 = __shfl(ts[s], (lane & ~3) | 1), a2 = __shfl(ts[s], (lane & ~3) | 2), a3 = __shfl(ts[s], (lane & ~3) | 3);
            h[s] = ((ts[s] + a1) + a2) + a3;
        }
        if (p == 0 && sl.row >= 0) {
            double2* __restrict__ dst = reinterpret_cast<double2*>(slot_h) + slot * 2;
            dst[0] = make_double2(h[0], h[1]); dst[1] = make_double2(h[2], h[3]);
        }
    }
    }
    double* __restrict__ s_row = s_rows[wave];
#pragma unroll
    for (int i = 0; i < 16; i++) s_row[i * 65 + lane] = (i >> 2) == p ? tr[i & 3] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 16) {
        const double* __restrict__ row = s_row + lane * 65;
        double v = 0.0;
#pragma unroll 8
        for (int l = 0; l < 64; l++) v += row[l];
        s_blk[wave][lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = 0.0;
        for (int w = 0; w < wpb; w++) v += s_blk[w][threadIdx.x];
        blk_trans[(int64_t) blockIdx.x * 16 + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------
// k_nb_hist: one wavefront per (region, bin): H[(region*4 + s)*256 + bin] = sum over the row slots of the bin, in the order
// of the static list (lane l takes entries l, l+64, ..., then a fixed shuffle tree).  Bins without slots are written 0.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nb_hist(int n_bins, const int32_t* __restrict__ bin_off, const int32_t* __restrict__ bin_list,
                                                 const double* __restrict__ slot_h, double* __restrict__ H) {
    const int bin = (int) blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (bin >= n_bins) return;
    const int a = bin_off[bin], b = bin_off[bin + 1];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = a + lane; k < b; k += 64) {
        const double2* __restrict__ src = reinterpret_cast<const double2*>(slot_h) + (int64_t) bin_list[k] * 2;
        const double2 u = src[0], v = src[1];
        acc[0] += u.x; acc[1] += u.y; acc[2] += v.x; acc[3] += v.y;
    }
#pragma unroll
    for (int s = 0; s < 4; s++)
        for (int o = 32; o > 0; o >>= 1) acc[s] += __shfl_down(acc[s], o);
    if (lane == 0) {
        const int region = bin >> 8, x = bin & 255;
#pragma unroll
        for (int s = 0; s < 4; s++) H[((int64_t) region * 4 + s) * 256 + x] = acc[s];
    }
}

// ------------------------------------------------------------------------------------------
// k_nb_total: one block of 1024 threads per region (block 0 also: log-likelihood, flag word).  The region's block of the
// estimator vector as k_chunk_stats_nb builds a chunk's: transition counts, then for every (state, component) the
// theta / lambda / weight increments over the coverage values (hmm_utils.c:537-566; one wavefront per (state, component), lanes over x),
// the weight denominators shared by the
// components of a state (hmm_utils.c:66-74).  Published with the checksum and the stamp of k_row_stats' total (hf_rows.h).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_nb_total(const int32_t* __restrict__ rw_off, int wpb, const double* __restrict__ blk_trans,
                                                  const double* __restrict__ H, const DevParams* __restrict__ P, const NbTables nb,
                                                  const double* __restrict__ chunk_ll, int64_t C, int64_t V, int K,
                                                  double* __restrict__ out, const unsigned* __restrict__ flags, double seq,
                                                  unsigned* __restrict__ done) {
    const int tid = threadIdx.x;
    __shared__ double counts[4][256];
    __shared__ double tpart[60][16];
    __shared__ double blockv[24 * HF_MAXCOMP + 16];
    __shared__ double s_ll;
    __shared__ unsigned long long s_x[16];
    const unsigned fl = (tid == 0 && flags) ? *flags : 0u;
    if (blockIdx.x == 0 && tid >= 960) {   // k_reduce's order over the chunk list
        const int lane = tid - 960;
        double acc = 0.0;
        for (int64_t c = lane; c < C; c += 64) acc += chunk_ll[c];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) { out[0] = acc; s_ll = acc; }
    }
    const int nreg = P->n_regions;
    const int rstride = 24 * K + 16;
    for (int r = blockIdx.x; r < nreg; r += gridDim.x) {
        for (int v = tid; v < rstride; v += 1024) blockv[v] = 0.0;
        for (int i = tid; i < 4 * 256; i += 1024) counts[i >> 8][i & 255] = H[(int64_t) r * 4 * 256 + i];
        __syncthreads();
        const int w0 = rw_off[r] / wpb, w1 = rw_off[r + 1] / wpb;
        {   // transition counts: the block partials of k_row_stats_nb in plan order, 60 interleaved accumulators per entry
            const int q = tid >> 4, i = tid & 15;
            if (q < 60) {
                double v = 0.0;
                for (int k = w0 + q; k < w1; k += 60) v += blk_trans[(int64_t) k * 16 + i];
                tpart[q][i] = v;
            }
        }
        __syncthreads();
        if (tid < 16) {
            double v = 0.0;
            for (int q = 0; q < 60; q++) v += tpart[q][tid];
            blockv[24 * K + tid] = v;
        }
        // one wavefront per (state, component): lane l takes the coverage values l, l+64, .. (coalesced table reads), then a
        // fixed shuffle tree over the lanes — theta / lambda / weight numerators and the first two denominators
        {
            const int wave = tid >> 6, nwaves = (int) (blockDim.x >> 6), lane = tid & 63;
            const int n0 = P->ncomp[0], n1 = P->ncomp[1], n2 = P->ncomp[2], n3 = P->ncomp[3];
            for (int q = wave; q < n0 + n1 + n2 + n3; q += nwaves) {   // (state, component) pairs over the wavefronts
                const int s = q < n0 ? 0 : q < n0 + n1 ? 1 : q < n0 + n1 + n2 ? 2 : 3;
                const int cc = q - (s == 0 ? 0 : s == 1 ? n0 : s == 2 ? n0 + n1 : n0 + n1 + n2);
                const int64_t pc = ((int64_t) r * 4 + s) * K + cc;
                const double* __restrict__ Pc = nb.P + pc * HF_NB_NX;
                const double* __restrict__ Dg = nb.dig + pc * HF_NB_NX;
                const double* __restrict__ Ex = nb.E + ((int64_t) r * 4 + s) * HF_NB_NX;
                const double rr = nb.r[pc], beta = nb.beta[pc], d0 = Dg[0];
                double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // th_num, th_den, la_num, la_den, w_num
                for (int x = lane; x < HF_NB_MAX_COVERAGE; x += 64) {
                    const double count = counts[s][x];
                    if (0 < count) {
                        const double w = count * Pc[x] / Ex[x];
                        const double delta = rr * (Dg[x] - d0);
                        acc[2] += w * delta; acc[3] += w;
                        acc[0] += w * delta * beta; acc[1] += w * delta * beta + w * (x - delta);
                        acc[4] += w;
                    }
                }
#pragma unroll
                for (int q = 0; q < 5; q++)
                    for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o);
                if (lane == 0) {
                    blockv[((s * 3 + 0) * 2 + 0) * K + cc] = acc[0]; blockv[((s * 3 + 0) * 2 + 1) * K + cc] = acc[1];
                    blockv[((s * 3 + 1) * 2 + 0) * K + cc] = acc[2]; blockv[((s * 3 + 1) * 2 + 1) * K + cc] = acc[3];
                    blockv[((s * 3 + 2) * 2 + 0) * K + cc] = acc[4];
                }
            }
        }
        __syncthreads();
        if (tid < 4) {
            const int nc = P->ncomp[tid];
            double den = 0.0;
            for (int c2 = 0; c2 < nc; c2++) den += blockv[((tid * 3 + 2) * 2 + 0) * K + c2];
            for (int c2 = 0; c2 < nc; c2++) blockv[((tid * 3 + 2) * 2 + 1) * K + c2] = den;
        }
        __syncthreads();
        unsigned long long x = 0ull;           // checksum of what this block writes (hf_cks_term)
        for (int v = tid; v < rstride; v += 1024) {
            const double d = blockv[v];
            const int64_t at = 1 + (int64_t) r * rstride + v;
            out[at] = d;
            x += hf_cks_term((unsigned long long) __double_as_longlong(d), at);
        }
        if (seq != 0.0) {
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            if ((tid & 63) == 0) s_x[tid >> 6] = x;
            __syncthreads();
            if (tid == 0) {
                unsigned long long c = (unsigned long long) __double_as_longlong(seq);
                for (int w = 0; w < 16; w++) c += s_x[w];
                if (r == 0) c += hf_cks_term((unsigned long long) __double_as_longlong(s_ll), 0) +
                                 hf_cks_term((unsigned long long) __double_as_longlong((double) fl), V);
                out[V + 2 + r] = __longlong_as_double((long long) c);
            }
        }
        __syncthreads();
    }
    if (tid == 0 && flags && blockIdx.x == 0) out[V] = (double) fl;
    if (seq != 0.0) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0 && atomicAdd(done, 1u) == gridDim.x - 1) { *done = 0u; out[V + 1] = seq; __threadfence_system(); }
    }
}
