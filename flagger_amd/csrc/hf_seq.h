// hf_seq.h — HF_ALGO_SEQ: the on-device cross-check of the scan path.  Every window's emission row by direct evaluation
// (k_emit_rows), then one wavefront per chunk visits the windows in order with the reference's exact operation order
// (k_fwd_seq, k_bwd_seq: hmm.c:333-545); the statistics come from the per-chunk kernels of hf_chunks.h.  ~30x slower than
// the scan path and never the default.
#pragma once
#include "hf_device.h"

// HF_ALGO_SEQ: emission row of every window by direct evaluation, E[t][16] (A8-A10); chunk-first windows hold
// e_s(x_0; alpha=0, preX=0) in row pre=0 (hmm.c:338-352)
__global__ void __launch_bounds__(256) k_emit_rows(int64_t N, const uint32_t* __restrict__ rec, const double* __restrict__ beta,
                                                   const DevParams* __restrict__ P, const double* __restrict__ nbE,
                                                   double* __restrict__ E, unsigned* __restrict__ flags) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const uint32_t r = rec[t];
    const bool first = REC_FIRST(r) != 0;
    const double x = (double) REC_X(r), px = first ? 0.0 : (double) REC_X(rec[t - 1]);
    unsigned nan = 0;
    double out[16];
    if (nbE) {   // negative_binomial: e_s(x) from the caller's table (hf_params.nb_E), the same in every row
        for (int s = 0; s < 4; s++) {
            const double e = nbE[((int64_t) REC_REGION(r) * 4 + s) * (HF_NB_MAX_COVERAGE + 1) + REC_X(r)];
            if (e != e) nan |= HF_FLAG_NAN;
            for (int p = 0; p < 4; p++) out[p * 4 + s] = (first && p != 0) ? 0.0 : e;
        }
    } else
    hf_emit_values(P, &P->reg[REC_REGION(r)], x, px, first, beta[t], out, &nan);
    double2* dst = reinterpret_cast<double2*>(E) + t * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = make_double2(out[2 * k], out[2 * k + 1]);
    if (nan) atomicOr(flags, nan);
}

__device__ __forceinline__ void load_E_window(const double* __restrict__ E, int64_t t, double* Ev) {
    const double2* __restrict__ src = reinterpret_cast<const double2*>(E) + t * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = src[k]; Ev[2 * k] = v.x; Ev[2 * k + 1] = v.y; }
}

// ------------------------------------------------------------------------------------------
// HF_ALGO_SEQ: one wavefront per chunk, windows visited in order with the reference's exact
// operation order; tiles of 64 windows are staged through LDS with coalesced loads/stores.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fwd_seq(const int64_t* __restrict__ off, const int32_t* __restrict__ chunk_tile0,
                                                const uint32_t* __restrict__ rec,
                                                const double* __restrict__ E, const DevParams* __restrict__ P,
                                                double* __restrict__ F, double* __restrict__ scale,
                                                double* __restrict__ tile_ll, unsigned* __restrict__ flags) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int64_t t0 = off[c], T = off[c + 1] - t0;
    __shared__ double Es[64][17];
    __shared__ double Fs[64][5];
    __shared__ uint32_t rs[64];
    double f[4] = {0.0, 0.0, 0.0, 0.0};
    double ll = 0.0;
    unsigned bad = 0;
    for (int64_t base = 0; base < T; base += 64) {
        const int n = (int) ((T - base) < 64 ? (T - base) : 64);
        if (lane < n) {
            const int64_t t = t0 + base + lane;
            rs[lane] = rec[t];
            load_E_window(E, t, &Es[lane][0]);
        }
        __syncthreads();
        for (int j = 0; j < n; j++) {
            const uint32_t r = rs[j];
            double nf[4], sc = 0.0;
            if (base + j == 0) { // hmm.c:333-364
                const DevRegion* __restrict__ R = &P->reg[REC_REGION(r)];
#pragma unroll
                for (int s = 0; s < 4; s++) { nf[s] = Es[j][s] * R->trans[4][s]; sc += nf[s]; }
            } else {             // hmm.c:366-420
                double Tm[16];
                load_T(P, r, Tm);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    double acc = 0.0;
#pragma unroll
                    for (int p = 0; p < 4; p++) acc += (f[p] * Tm[p * 4 + s] * Es[j][p * 4 + s]);
                    nf[s] = acc;
                    sc += acc;
                }
                if (sc < 1e-50) bad |= HF_FLAG_SCALE;
            }
#pragma unroll
            for (int s = 0; s < 4; s++) { f[s] = nf[s] / sc; }
            ll += log(sc);
            if (lane == 0) { Fs[j][0] = f[0]; Fs[j][1] = f[1]; Fs[j][2] = f[2]; Fs[j][3] = f[3]; Fs[j][4] = sc; }
        }
        __syncthreads();
        if (lane < n) {
            const int64_t t = t0 + base + lane;
            reinterpret_cast<double2*>(F)[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], base + lane, 0)] = make_double2(Fs[lane][0], Fs[lane][1]);
            reinterpret_cast<double2*>(F)[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], base + lane, 1)] = make_double2(Fs[lane][2], Fs[lane][3]);
            scale[t] = Fs[lane][4];
        }
        __syncthreads();
    }
    // the chunk's log-likelihood goes through the same per-tile slots as the scan path (k_chunk_stats sums them)
    const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
    for (int k = lane; k < nt; k += 64) tile_ll[k0 + k] = k == 0 ? ll : 0.0;
    if (lane == 0 && bad) atomicOr(flags, bad);
}

__global__ void __launch_bounds__(64) k_bwd_seq(const int64_t* __restrict__ off, const int32_t* __restrict__ chunk_tile0,
                                                const uint32_t* __restrict__ rec,
                                                const double* __restrict__ E, const DevParams* __restrict__ P,
                                                const double* __restrict__ F, const double* __restrict__ scale,
                                                double* __restrict__ B, int8_t* __restrict__ label,
                                                unsigned* __restrict__ flags) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int64_t t0 = off[c], T = off[c + 1] - t0;
    if (T <= 0) return;
    __shared__ double Es[64][17];   // E of window i+1
    __shared__ double Fs[64][5];    // f_i[0..3], scale_i
    __shared__ double Bs[64][4];
    __shared__ uint32_t rs[64];     // rec of window i+1
    __shared__ int8_t Ls[64];
    double b[4];
    unsigned bad = 0;
    { // last column, hmm.c:452-467
        const int64_t t = t0 + T - 1;
        const DevRegion* __restrict__ R = &P->reg[REC_REGION(rec[t])];
        const double sc = scale[t];
        double f[4];
        {
            const double2* __restrict__ F2 = reinterpret_cast<const double2*>(F);
            const double2 f01 = F2[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], T - 1, 0)], f23 = F2[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], T - 1, 1)];
            f[0] = f01.x; f[1] = f01.y; f[2] = f23.x; f[3] = f23.y;
        }
        for (int s = 0; s < 4; s++) b[s] = R->trans[s][4] / sc;
        if (lane == 0) {
            reinterpret_cast<double2*>(B)[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], T - 1, 0)] = make_double2(b[0], b[1]);
            reinterpret_cast<double2*>(B)[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], T - 1, 1)] = make_double2(b[2], b[3]);
            label[t] = (int8_t) posterior_label(f, b, sc);
        }
    }
    // columns T-2 .. 0 in tiles; tile covers i in [lo, lo+n)
    for (int64_t hi = T - 1; hi > 0; hi -= 64) {
        const int64_t lo = hi >= 64 ? hi - 64 : 0;
        const int n = (int) (hi - lo);
        if (lane < n) {
            const int64_t t = t0 + lo + lane; // window i
            rs[lane] = rec[t + 1];
            load_E_window(E, t + 1, &Es[lane][0]);
            {
                const double2* __restrict__ F2 = reinterpret_cast<const double2*>(F);
                const double2 f01 = F2[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], lo + lane, 0)], f23 = F2[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], lo + lane, 1)];
                Fs[lane][0] = f01.x; Fs[lane][1] = f01.y; Fs[lane][2] = f23.x; Fs[lane][3] = f23.y;
            }
            Fs[lane][4] = scale[t];
        }
        __syncthreads();
        for (int j = n - 1; j >= 0; j--) { // hmm.c:470-529
            double Tm[16];
            load_T(P, rs[j], Tm);
            double nb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int p = 0; p < 4; p++) nb[p] += Tm[p * 4 + s] * Es[j][p * 4 + s] * b[s];
            const double sc = Fs[j][4];
            if (sc < 1e-50) bad |= HF_FLAG_SCALE;
            double f[4];
#pragma unroll
            for (int s = 0; s < 4; s++) { b[s] = nb[s] / sc; f[s] = Fs[j][s]; }
            const int lab = posterior_label(f, b, sc);
            if (lane == 0) { Bs[j][0] = b[0]; Bs[j][1] = b[1]; Bs[j][2] = b[2]; Bs[j][3] = b[3]; Ls[j] = (int8_t) lab; }
        }
        __syncthreads();
        if (lane < n) {
            const int64_t t = t0 + lo + lane;
            reinterpret_cast<double2*>(B)[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], lo + lane, 0)] = make_double2(Bs[lane][0], Bs[lane][1]);
            reinterpret_cast<double2*>(B)[fb_slot_w<HF_SCAN_L>(chunk_tile0[c], lo + lane, 1)] = make_double2(Bs[lane][2], Bs[lane][3]);
            label[t] = Ls[lane];
        }
        __syncthreads();
    }
    if (lane == 0 && bad) atomicOr(flags, bad);
}

