// hf_nb.h — the negative_binomial model on the device.
//
// In that model every emission quantity depends on the coverage x alone (NegativeBinomial_getProb, hmm_utils.c:480-520:
// no alpha, no previous coverage, no contig-end factor), and the statistics go through per-state count data
// (hmm.c:615-617, count_data.c:49-57) that is turned into estimator increments once per chunk
// (EmissionDistSeries_updateAllEstimatorsUsingCountData, hmm_utils.c:1661-1673; NegativeBinomial_updateEstimator,
// :537-566).  The caller tabulates E(x), the component probabilities and the digamma table with its own libm
// (hf_params.nb_*), so the device only gathers, multiplies and sums:
//   k_tables_nb       emission rows of the occurring keys / contig-end windows from E[r][s][x]
//   (k_prod_tile, k_carry, k_fb_tile: unchanged — they only consume rows)
//   k_stats_tile_nb   per tile: xi transition counts + count data hist[state][min(x,249)]
//   k_chunk_stats_nb  per chunk: tile partials in tile order -> estimator increments in the standard vector layout
//                     (parameter 0 = theta, 1 = lambda, 2 = weight)
#pragma once
#include "hf_scan.h"

#define HF_NB_NX (HF_NB_MAX_COVERAGE + 1)
#define HF_NB_TILE_VEC (16 + 4 * 256)     // per (tile, region): trans[16], hist[4][256] (bins >= 250 stay zero)

struct NbTables {                          // device copies of hf_params.nb_*
    const double* E;      // [R][4][NX]
    const double* P;      // [R][4][K][NX]
    const double* dig;    // [R][4][K][NX]
    const double* r;      // [R][4][K]
    const double* beta;   // [R][4][K]
};

__global__ void __launch_bounds__(256) k_tables_nb(int n_keys, const int32_t* __restrict__ keys, int n_slow,
                                                   const int64_t* __restrict__ slow_w, const uint32_t* __restrict__ rec,
                                                   int M, const double* __restrict__ nbE, double* __restrict__ lutE,
                                                   double* __restrict__ Es, unsigned* __restrict__ flags) {
    const int job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job == 0) *flags = 0u;   // first kernel of every pass
    if (job >= n_keys + n_slow) return;
    int r, x;
    bool first = false;
    double* dst;
    if (job < n_keys) {
        const int64_t key = keys[job];
        const int64_t MM = (int64_t) M * M;
        r = (int) (key / MM);
        x = (int) ((key % MM) / M);
        dst = lutE + key * 16;
    } else {
        const int k = job - n_keys;
        const uint32_t rw = rec[slow_w[k]];
        r = (int) REC_REGION(rw); x = (int) REC_X(rw); first = REC_FIRST(rw) != 0;
        dst = Es + (int64_t) k * 16;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const double e = nbE[((int64_t) r * 4 + s) * HF_NB_NX + x];
#pragma unroll
        for (int p = 0; p < 4; p++) dst[HF_PS(p, s)] = (first && p != 0) ? 0.0 : e;   // chunk-first: row pre = 0 only (hmm.c:338-352)
    }
}

// One wavefront per tile.  Per step j every lane computes the 16 xi values of the pair ending at its j-th window and
// parks them in LDS; lanes 0..15 add the 64 records' values to their transition accumulator, and the lane that owns
// bin min(x,249) (owner = bin % 64) adds the record to its count-data bins — records are visited in a fixed order, so
// the sums are reproducible.
template <int L>
__global__ void __launch_bounds__(256) k_stats_tile_nb(int ntiles, const TileDesc* __restrict__ td,
                                                       const uint32_t* __restrict__ rec, const RowSrc S,
                                                       const DevParams* __restrict__ P, const double* __restrict__ F,
                                                       const double* __restrict__ B, const uint64_t* __restrict__ regmask,
                                                       double* __restrict__ tile_hist) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wave;   // 4 wavefronts per block unless LDS forces fewer
    if (tile >= ntiles) return;
    // per wave: records [64][16], bins of the records [64] (as doubles), histogram [4 states][4 bins][64 lanes]
    double* __restrict__ s_rec = s_tab + P->n_regions * HF_TAB_STRIDE + wave * (64 * 16 + 64 + 16 * 64);
    double* __restrict__ s_bin = s_rec + 64 * 16;
    double* __restrict__ s_hist = s_bin + 64;
    const TileDesc d = td[tile];
    const int64_t t0 = d.t0, T = d.T, base = d.base;
    const int nreg = P->n_regions;
    const int64_t a0 = base + (int64_t) lane * L;
    uint32_t rr[L + 1];
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
    for (int r = 0; r < nreg; r++) {
        if (!((in_chunk >> r) & 1ull)) continue;
        double* __restrict__ dst = tile_hist + ((int64_t) tile * nreg + r) * HF_NB_TILE_VEC;
        if (!((present >> r) & 1ull)) {
            for (int i = lane; i < HF_NB_TILE_VEC; i += 64) dst[i] = 0.0;
            continue;
        }
        double trans_acc = 0.0;                              // lanes 0..15: entry pre*4+s = lane
#pragma unroll
        for (int i = 0; i < 16; i++) s_hist[i * 64 + lane] = 0.0;
#pragma unroll 1
        for (int j = 0; j < L; j++) {
            const bool mine = ok[j] && (int) REC_REGION(rr[j + 1]) == r;
            double adj[16];
#pragma unroll
            for (int k = 0; k < 16; k++) adj[k] = 0.0;
            if (mine) {
                double Ev[16], Tm[16];
                load_row(row_ptr(S, rr[j + 1], rr[j], sidx[j]), Ev);
                // f of the window before (the previous lane's last one for j == 0), b of the window itself
                const int64_t fs = j > 0 ? fb_slot<L>(tile, lane, j - 1, 0)
                                         : (lane > 0 ? fb_slot<L>(tile, lane - 1, L - 1, 0) : fb_slot<L>(tile - 1, 63, L - 1, 0));
                const int64_t bs = fb_slot<L>(tile, lane, j, 0);
                const double2 f01 = reinterpret_cast<const double2*>(F)[fs], f23 = reinterpret_cast<const double2*>(F)[fs + 64];
                const double2 b01 = reinterpret_cast<const double2*>(B)[bs], b23 = reinterpret_cast<const double2*>(B)[bs + 64];
                lds_Tm(s_tab, rr[j + 1], Tm);
                const double f[4] = {f01.x, f01.y, f23.x, f23.y};
                const double b1[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const double count = f[p] * Tm[HF_PS(p, s)] * Ev[HF_PS(p, s)] * b1[s];
                        adj[s * 4 + p] = count / HF_TERMINATION_PROB;     // hmm.c:613-614
                    }
            }
#pragma unroll
            for (int k = 0; k < 16; k++) s_rec[lane * 16 + k] = adj[k];   // state-major: [s*4 + pre]
            {
                const unsigned x = REC_X(rr[j + 1]);
                s_bin[lane] = mine ? (double) (x < HF_NB_MAX_COVERAGE ? x : HF_NB_MAX_COVERAGE - 1) : -1.0;   // count_data.c:49-57
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 16) {                                  // hmm_utils.c:2010-2015; lane = pre*4 + s reads record entry s*4 + pre
                const int e = (lane & 3) * 4 + (lane >> 2);
#pragma unroll 8
                for (int k = 0; k < 64; k++) trans_acc += s_rec[k * 16 + e];
            }
#pragma unroll 1
            for (int k = 0; k < 64; k++) {
                const int xb = (int) s_bin[k];
                if (xb >= 0 && (xb & 63) == lane) {
                    const int bin = xb >> 6;
#pragma unroll
                    for (int s = 0; s < 4; s++) {             // state outer, pre inner: the reference's order (hmm.c:588-589)
                        double h = s_hist[(s * 4 + bin) * 64 + lane];
#pragma unroll
                        for (int p = 0; p < 4; p++) h += s_rec[k * 16 + s * 4 + p];
                        s_hist[(s * 4 + bin) * 64 + lane] = h;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane < 16) dst[lane] = trans_acc;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int b = 0; b < 4; b++) dst[16 + s * 256 + b * 64 + lane] = s_hist[(s * 4 + b) * 64 + lane];
    }
}

// per chunk: log-likelihood, transition counts, count data (tile partials in tile order), then the estimator
// increments of NegativeBinomial_updateEstimator for every (state, x < 250) with a positive count, x ascending
// (hmm_utils.c:1661-1673, 537-566), written in the standard chunk-vector layout.
__global__ void __launch_bounds__(256) k_chunk_stats_nb(const int32_t* __restrict__ chunk_tile0, const uint64_t* __restrict__ regmask,
                                                        const double* __restrict__ tile_hist, const double* __restrict__ tile_ll,
                                                        const DevParams* __restrict__ P, const NbTables nb,
                                                        double* __restrict__ chunk_stats, int64_t V, int K, int full) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
    const int nreg = P->n_regions;
    const int64_t rstride = 24 * (int64_t) K + 16;
    const uint64_t present = regmask[c];
    double* __restrict__ vec = chunk_stats + (int64_t) c * V;
    if (tid < 64) {
        double s = 0.0;
        for (int k = tid; k < nt; k += 64) s += tile_ll[k0 + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (tid == 0) vec[0] = s;
    }
    for (int64_t v = 1 + tid; v < V; v += blockDim.x) vec[v] = 0.0;
    if (!full) return;
    __shared__ double counts[4][256];
    __syncthreads();
    for (int r = 0; r < nreg; r++) {
        if (!((present >> r) & 1ull)) continue;
        double* __restrict__ dst = vec + 1 + r * rstride;
        for (int i = tid; i < HF_NB_TILE_VEC; i += blockDim.x) {
            double v = 0.0;
            for (int k = 0; k < nt; k++) v += tile_hist[((int64_t) (k0 + k) * nreg + r) * HF_NB_TILE_VEC + i];
            if (i < 16) dst[24 * K + i] = v;
            else counts[(i - 16) >> 8][(i - 16) & 255] = v;
        }
        __syncthreads();
        // thread (s, cc): theta / lambda / weight numerators and the first two denominators of component cc of state s
        const int s = tid >> 4, cc = tid & 15;
        if (tid < 64 && cc < P->ncomp[s]) {
            const int64_t pc = ((int64_t) r * 4 + s) * K + cc;
            const double* __restrict__ Pc = nb.P + pc * HF_NB_NX;
            const double* __restrict__ Dg = nb.dig + pc * HF_NB_NX;
            const double* __restrict__ Ex = nb.E + ((int64_t) r * 4 + s) * HF_NB_NX;
            const double rr = nb.r[pc], beta = nb.beta[pc], d0 = Dg[0];
            double th_num = 0.0, th_den = 0.0, la_num = 0.0, la_den = 0.0, w_num = 0.0;
            for (int x = 0; x < HF_NB_MAX_COVERAGE; x++) {
                const double count = counts[s][x];
                if (0 < count) {
                    const double w = count * Pc[x] / Ex[x];
                    const double delta = rr * (Dg[x] - d0);
                    la_num += w * delta; la_den += w;
                    th_num += w * delta * beta; th_den += w * delta * beta + w * (x - delta);
                    w_num += w;
                }
            }
            dst[((s * 3 + 0) * 2 + 0) * K + cc] = th_num; dst[((s * 3 + 0) * 2 + 1) * K + cc] = th_den;
            dst[((s * 3 + 1) * 2 + 0) * K + cc] = la_num; dst[((s * 3 + 1) * 2 + 1) * K + cc] = la_den;
            dst[((s * 3 + 2) * 2 + 0) * K + cc] = w_num;
        }
        // thread 64 + s: the weight estimator's denominator, shared by all components of state s (hmm_utils.c:66-74):
        // every w of every component, x outer / component inner
        if (tid >= 64 && tid < 68) {
            const int s2 = tid - 64, nc = P->ncomp[s2];
            const double* __restrict__ Ex = nb.E + ((int64_t) r * 4 + s2) * HF_NB_NX;
            double den = 0.0;
            for (int x = 0; x < HF_NB_MAX_COVERAGE; x++) {
                const double count = counts[s2][x];
                if (0 < count)
                    for (int c2 = 0; c2 < nc; c2++) den += count * nb.P[(((int64_t) r * 4 + s2) * K + c2) * HF_NB_NX + x] / Ex[x];
            }
            for (int c2 = 0; c2 < nc; c2++) dst[((s2 * 3 + 2) * 2 + 1) * K + c2] = den;
        }
        __syncthreads();
    }
}
