// hf_nb.h — the negative_binomial model on the device.
//
// In that model every emission quantity depends on the coverage x alone (NegativeBinomial_getProb, hmm_utils.c:480-520:
// no alpha, no previous coverage, no contig-end factor), and the statistics go through per-state count data
// (hmm.c:615-617, count_data.c:49-57) that is turned into estimator increments once per chunk
// (EmissionDistSeries_updateAllEstimatorsUsingCountData, hmm_utils.c:1661-1673; NegativeBinomial_updateEstimator,
// :537-566).  The caller tabulates E(x), the component probabilities and the digamma table with its own libm
// (hf_params.nb_*), so the device only gathers, multiplies and sums:
//   k_tables_nb       emission rows of the occurring keys / contig-end windows from E[r][s][x]
//   (the segment kernels of hf_seg.h: unchanged — they only consume rows; k_arows makes their rows of A)
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

// One wavefront per tile, lane l owns the pairs that end at its L windows.  The 16 xi values of a pair go to the lane's
// transition accumulators (registers) and to the wave-private count data hist[state][min(x,249)] in LDS.  Lanes whose
// pairs fall into the same bin are serialised in ascending lane order: every pending lane bids for its bin with an LDS
// atomic MIN of its lane number (the result does not depend on the order of the bids), the winner adds its four
// values per state (state outer / pre inner, hmm.c:588-589) and retires; rounds repeat until no lane is pending — as
// many rounds as the most crowded bin has pairs.  A fixed order, so the sums are reproducible.
#define HF_NB_WAVE_LDS (16 * 65 + 4 * 256 + 128)     // doubles per wavefront: lane-sum rows, count data, bids (256 x u32)
template <int L>
__global__ void __launch_bounds__(256) k_stats_tile_nb(int ntiles, const TileDesc* __restrict__ td,
                                                       const uint32_t* __restrict__ rec, const RowSrc S,
                                                       const DevParams* __restrict__ P, const double* __restrict__ F,
                                                       const double* __restrict__ B, const uint64_t* __restrict__ regmask,
                                                       double* __restrict__ tile_hist, const int32_t* __restrict__ slot_of) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wave;   // 4 wavefronts per block unless LDS forces fewer
    if (tile >= ntiles) return;
    double* __restrict__ s_row = s_tab + P->n_regions * HF_TAB_STRIDE + wave * HF_NB_WAVE_LDS;   // [16][65]
    double* __restrict__ s_hist = s_row + 16 * 65;                                                // [4][256]
    unsigned* __restrict__ s_bid = reinterpret_cast<unsigned*>(s_hist + 4 * 256);                 // [256]
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
    for (int i = lane; i < 256; i += 64) s_bid[i] = 0xffffffffu;
    for (int r = 0; r < nreg; r++) {
        if (!((in_chunk >> r) & 1ull)) continue;
        double* __restrict__ dst = tile_hist + ((int64_t) tile * nreg + r) * HF_NB_TILE_VEC;
        if (!((present >> r) & 1ull)) {
            for (int i = lane; i < HF_NB_TILE_VEC; i += 64) dst[i] = 0.0;
            continue;
        }
        double tr[16];                                        // transition counts of this lane's pairs, [pre*4 + s]
#pragma unroll
        for (int k = 0; k < 16; k++) tr[k] = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) s_hist[i * 64 + lane] = 0.0;   // hist[s][bin]: s*256 + bin
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (int j = 0; j < L; j++) {
            const bool mine = ok[j] && (int) REC_REGION(rr[j + 1]) == r;
            double adj[16];                                   // state-major: [s*4 + pre]
#pragma unroll
            for (int k = 0; k < 16; k++) adj[k] = 0.0;
            if (mine) {
                double Ev[16], Tm[16];
                load_row(row_ptr(S, rr[j + 1], rr[j], sidx[j]), Ev);
                // f of the window before (the previous lane's last one for j == 0), b of the window itself
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
                lds_Tm(s_tab, rr[j + 1], Tm);
                const double f[4] = {f01.x, f01.y, f23.x, f23.y};
                const double b1[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const double count = f[p] * Tm[HF_PS(p, s)] * Ev[HF_PS(p, s)] * b1[s];
                        adj[s * 4 + p] = count / HF_TERMINATION_PROB;     // hmm.c:613-614
                        tr[p * 4 + s] += adj[s * 4 + p];                  // hmm_utils.c:2010-2015
                    }
            }
            const unsigned x = REC_X(rr[j + 1]);
            const int bin = (int) (x < HF_NB_MAX_COVERAGE ? x : HF_NB_MAX_COVERAGE - 1);          // count_data.c:49-57
            bool pending = mine;
            while (__any(pending)) {
                if (pending) atomicMin(&s_bid[bin], (unsigned) lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool win = pending && s_bid[bin] == (unsigned) lane;
                if (win) {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        double h = s_hist[s * 256 + bin];
#pragma unroll
                        for (int p = 0; p < 4; p++) h += adj[s * 4 + p];
                        s_hist[s * 256 + bin] = h;
                    }
                    s_bid[bin] = 0xffffffffu;
                    pending = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        // transition counts: sums over the 64 lanes through LDS rows, one lane per entry, lane order
#pragma unroll
        for (int k = 0; k < 16; k++) s_row[k * 65 + lane] = tr[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16) {
            const double* __restrict__ row = s_row + lane * 65;
            double v = 0.0;
#pragma unroll 8
            for (int l = 0; l < 64; l++) v += row[l];
            dst[lane] = v;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) dst[16 + i * 64 + lane] = s_hist[i * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// per chunk: log-likelihood, transition counts, count data (tile partials in tile order), then the estimator
// increments of NegativeBinomial_updateEstimator for every (state, x < 250) with a positive count, x ascending
// (hmm_utils.c:1661-1673, 537-566), written in the standard chunk-vector layout.
__global__ void __launch_bounds__(256) k_chunk_stats_nb(const int32_t* __restrict__ chunk_tile0, const int32_t* __restrict__ chunk_ll0,
                                                        const uint64_t* __restrict__ regmask,
                                                        const double* __restrict__ tile_hist, const double* __restrict__ tile_ll,
                                                        const DevParams* __restrict__ P, const NbTables nb,
                                                        double* __restrict__ chunk_stats, int64_t V, int K, int full) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
    const int nreg = P->n_regions;
    const int64_t rstride = 24 * (int64_t) K + 16;
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
    __shared__ double counts[4][256];
    __syncthreads();
    for (int r = 0; r < nreg; r++) {
        if (!((present >> r) & 1ull)) continue;
        double* __restrict__ dst = vec + 1 + r * rstride;
        for (int i = tid; i < HF_NB_TILE_VEC; i += blockDim.x) {
            double v = 0.0;
            int k = 0;
            for (; k + 8 <= nt; k += 8) {   // 8 loads in flight, adds stay in tile order
                double xk[8];
#pragma unroll
                for (int u = 0; u < 8; u++) xk[u] = tile_hist[((int64_t) (k0 + k + u) * nreg + r) * HF_NB_TILE_VEC + i];
#pragma unroll
                for (int u = 0; u < 8; u++) v += xk[u];
            }
            for (; k < nt; k++) v += tile_hist[((int64_t) (k0 + k) * nreg + r) * HF_NB_TILE_VEC + i];
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
        // the weight estimator's denominator is shared by all components of a state (hmm_utils.c:66-74): the sum of every
        // component's w — taken here as the sum over components (index order) of their own sums over x
        __syncthreads();
        if (tid >= 64 && tid < 68) {
            const int s2 = tid - 64, nc = P->ncomp[s2];
            double den = 0.0;
            for (int c2 = 0; c2 < nc; c2++) den += dst[((s2 * 3 + 2) * 2 + 0) * K + c2];
            for (int c2 = 0; c2 < nc; c2++) dst[((s2 * 3 + 2) * 2 + 1) * K + c2] = den;
        }
        __syncthreads();
    }
}
