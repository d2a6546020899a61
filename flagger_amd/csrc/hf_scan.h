// hf_scan.h — HF_ALGO_SCAN: tile-parallel forward/backward for 4-state chains on gfx950.
//
// The scaled forward recurrence f_t ∝ f_{t-1}·A_t (A_t[pre][s] = T_t(pre,s)·e_t(pre,s), hmm.c:366-420)
// is a product of 4x4 non-negative matrices, hence associative.  A chunk is cut into tiles of 64·L
// windows, one wavefront per tile, every lane owning L consecutive windows:
//   k_emit_tile  emission rows + lane products + the product of every tile (all tiles of all chunks at once)
//   k_carry      per chunk: sequential sweep over its <= T/(64L) tile products -> carried-in forward
//                vector and carried-in backward direction of every tile
//   k_fwd_tile   per tile: Kogge-Stone scan of the 64 lane products (DPP/ds_bpermute shuffles, power-of-two
//                renormalisation after every product: exact, nothing underflows) gives each lane the product
//                of everything before it; the lane then REPLAYS its L windows with the reference's exact
//                operation order from the carried-in, normalised vector
//   k_bwd_tile   mirror image with suffix products (hmm.c:470-529) + posterior argmax (hmm.c:671-692); the
//                absolute magnitude of a carried-in b vector is recovered from the invariant
//                sum_s f_t[s]·b_t[s]·scale_t = terminationProb of the scaled forward-backward
// Every f_t / b_t / scale_t is therefore produced by the reference's own arithmetic; only the carried-in
// vectors differ from a purely sequential run, in the last ulp.
#pragma once
#include "hf_device.h"

struct M4 { double m[16]; };

// Conditional-transition and start tables of every region, staged once per block in LDS (136 doubles per region:
// tcond[8][16], start[4], pad): the per-window lookup no longer waits on global memory behind the window record.
#define HF_TAB_STRIDE 136
__device__ __forceinline__ void fill_tab(const DevParams* __restrict__ P, double* __restrict__ s_tab) {
    const int n = P->n_regions * HF_TAB_STRIDE;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / HF_TAB_STRIDE, k = i % HF_TAB_STRIDE;
        s_tab[i] = k < 128 ? P->reg[r].tcond[k >> 4][k & 15] : (k < 132 ? P->reg[r].trans[4][k - 128] : 0.0);
    }
    __syncthreads();
}

__device__ __forceinline__ void lds_Tm(const double* __restrict__ s_tab, uint32_t r, double Tm[16]) {
    if (REC_FIRST(r)) {                       // chunk-first window: the start row for every pre
        const double* __restrict__ s = s_tab + REC_REGION(r) * HF_TAB_STRIDE + 128;
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = s[k & 3];
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

// Scan-mode layout of the emission stash: tile-major, lane-minor, so that the 64 lanes of the wavefront that owns a
// tile read/write 1 KiB contiguous per instruction: E2[((tile*L + i)*8 + k2)*64 + lane] = (E[2*k2], E[2*k2+1]) of
// the window `i` of `lane` (window index in chunk = tile_base + lane*L + i).
template <int L>
__device__ __forceinline__ int64_t e_slot(int tile, int lane, int i) { return (((int64_t) tile * L + i) * 8) * 64 + lane; }

template <int L>
__device__ __forceinline__ void load_E(const double* __restrict__ E, int tile, int lane, int i, double Ev[16]) {
    const double2* __restrict__ src = reinterpret_cast<const double2*>(E) + e_slot<L>(tile, lane, i);
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = src[k * 64]; Ev[2 * k] = v.x; Ev[2 * k + 1] = v.y; }
}

template <int L>
__device__ __forceinline__ void store_E(double* __restrict__ E, int tile, int lane, int i, const double Ev[16]) {
    double2* __restrict__ dst = reinterpret_cast<double2*>(E) + e_slot<L>(tile, lane, i);
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k * 64] = make_double2(Ev[2 * k], Ev[2 * k + 1]);
}

// T of one window; chunk-first windows use the start row for every pre (k_emit puts e_s(x_0) in row 0)
__device__ __forceinline__ void load_Tm(const DevParams* __restrict__ P, uint32_t r, double Tm[16]) {
    if (REC_FIRST(r)) {
        const DevRegion* __restrict__ R = &P->reg[REC_REGION(r)];
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = R->trans[4][k & 3];
    } else {
        load_T(P, r, Tm);
    }
}

// ------------------------------------------------------------------------------------------
// k_lut: this iteration's emission tables for interior windows (beta == beta_star), per region and per
// (x, x_prev) in [0, M)^2 — the same device functions as the direct evaluation, so the values are identical:
//   lutE[r][x*M+px][16]       the emission row E[pre][s]
//   lutC[r][x*M+px][K][4]     component probabilities of the collapsed state, [component][u-th distinct alpha]
// NaNs are stored, not reported: only a window that actually uses the entry raises HF_E_NAN.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lut(int M, int K, const DevParams* __restrict__ P, double* __restrict__ lutE,
                                             double* __restrict__ lutC, unsigned* __restrict__ flags) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *flags = 0u;   // first kernel of every pass
    const int r = blockIdx.y;
    const int64_t MM = (int64_t) M * M;
    const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= MM) return;
    const double x = (double) (idx / M), px = (double) (idx % M);
    const DevRegion* __restrict__ R = &P->reg[r];
    const double bs = P->beta_star;
    unsigned nan = 0;
    double out[16];
    hf_emit_values<true>(P, R, x, px, false, bs, out, &nan);
    double2* dst = reinterpret_cast<double2*>(lutE) + ((int64_t) r * MM + idx) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = make_double2(out[2 * k], out[2 * k + 1]);
    const int ncol = P->ncomp[3], nu = P->nuniq[3];
    double* __restrict__ c = lutC + (((int64_t) r * MM + idx) * 4) * K;
    for (int cc = 0; cc < ncol; cc++)
        for (int u = 0; u < 4; u++)
            c[cc * 4 + u] = u < nu ? hf_gauss_comp_star(R->m1[3][u][cc], R->gvar[3][cc], R->gnorm[3][cc], x, px,
                                                        P->ualpha[3][u], bs, &nan) : 0.0;
}

// ------------------------------------------------------------------------------------------
// k_emit_tile: one wavefront per tile.  Each lane evaluates the emission rows of its L windows (written to the
// stash E, 128 B per window), multiplies them into its lane product Q_l (written to Qs: the forward AND the
// backward tile kernels start from it instead of re-reading E for a first pass), and an ordered shuffle tree
// over the lanes gives the tile product Pt.  Chunk-first windows are excluded from Q_l and Pt.
// ------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256) k_emit_tile(int ntiles, const int32_t* __restrict__ tile_chunk,
                                                   const int64_t* __restrict__ tile_base,
                                                   const int64_t* __restrict__ off, const uint32_t* __restrict__ rec,
                                                   const double* __restrict__ beta, const DevParams* __restrict__ P,
                                                   const double* __restrict__ lutE, int M,
                                                   double* __restrict__ E, double* __restrict__ Qs,
                                                   double* __restrict__ Pt, unsigned* __restrict__ flags) {
    const int64_t MM = (int64_t) M * M;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const int c = tile_chunk[tile];
    const int64_t t0 = off[c], T = off[c + 1] - t0, base = tile_base[tile];
    const int64_t a = base + (int64_t) lane * L;
    unsigned nan = 0;
    M4 Q;
    m4_identity(Q);
#pragma unroll 1
    for (int i = 0; i < L; i++) {
        if (a + i < T) {
            const int64_t t = t0 + a + i;
            double Ev[16];
            const uint32_t r = rec[t];
            const double bt = beta[t];
            const unsigned xi = REC_X(r), reg = REC_REGION(r);
            const unsigned pxi = REC_FIRST(r) ? 0u : REC_X(rec[t - 1]);
            // wave-uniform choice: every active lane's window is an interior, non-first one => its emission row is a
            // function of (region, x, x_prev) only and comes from this iteration's table (k_lut)
            if (__all(bt == P->beta_star && !REC_FIRST(r))) {
                const double2* __restrict__ src = reinterpret_cast<const double2*>(lutE) + (((int64_t) reg * MM + xi * M + pxi) * 8);
#pragma unroll
                for (int k = 0; k < 8; k++) { const double2 v = src[k]; Ev[2 * k] = v.x; Ev[2 * k + 1] = v.y; }
                bool isnan = false;
#pragma unroll
                for (int k = 0; k < 16; k++) isnan |= Ev[k] != Ev[k];
                if (isnan) nan |= HF_FLAG_NAN;
            } else {
                hf_emit_values<false>(P, &P->reg[reg], (double) xi, (double) pxi, REC_FIRST(r) != 0, bt, Ev, &nan);
            }
            store_E<L>(E, tile, lane, i, Ev);
            if (!REC_FIRST(r)) {
                double Tm[16];
                load_T(P, r, Tm);
                M4 A, R;
#pragma unroll
                for (int k = 0; k < 16; k++) A.m[k] = Tm[k] * Ev[k];
                m4_mul(R, Q, A);
                Q = R;
                m4_renorm(Q);
            }
        }
    }
    {
        double2* dst = reinterpret_cast<double2*>(Qs + ((int64_t) tile * 64 + lane) * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    // ordered tree product over lanes: after step d, lane l (l % 2d == 0) holds the product of lanes l..l+2d-1
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        M4 Rgt, R;
        m4_shfl_down(Rgt, Q, d);
        if ((lane & (2 * d - 1)) == 0) { m4_mul(R, Q, Rgt); Q = R; m4_renorm(Q); }
    }
    if (lane == 0) {
        double2* dst = reinterpret_cast<double2*>(Pt + (int64_t) tile * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = make_double2(Q.m[2 * k], Q.m[2 * k + 1]);
    }
    if (nan) atomicOr(flags, nan);
}

__device__ __forceinline__ void load_lane_product(M4& Q, const double* __restrict__ Qs, int tile, int lane) {
    const double2* __restrict__ src = reinterpret_cast<const double2*>(Qs + ((int64_t) tile * 64 + lane) * 16);
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = src[k]; Q.m[2 * k] = v.x; Q.m[2 * k + 1] = v.y; }
}

// ------------------------------------------------------------------------------------------
// k_carry: per chunk, sweep the tile products.  cf[tile] = normalised forward vector entering the tile
// (tile 0 of a chunk: unused, the tile kernel starts from window 0); cb[tile] = direction of b at the
// first window after BACKWARD tile `tile` (backward tile k owns windows base_k-1 .. base_k+64L-2).
// Wave 0 sweeps forward, wave 1 backward; each stages the tile products through its half of LDS in batches of
// HF_CARRY_BATCH tiles (coalesced loads by all 64 lanes), then lane 0 runs the dependent chain out of LDS.
// ------------------------------------------------------------------------------------------
#define HF_CARRY_BATCH 128
template <int L>
__global__ void __launch_bounds__(128) k_carry(const int64_t* __restrict__ off, const int32_t* __restrict__ chunk_tile0,
                                               const uint32_t* __restrict__ rec, const double* __restrict__ E,
                                               const DevParams* __restrict__ P, const double* __restrict__ Pt,
                                               double* __restrict__ cf, double* __restrict__ cb) {
    __shared__ __attribute__((aligned(16))) double s_pt[2][HF_CARRY_BATCH * 16];
    __shared__ double s_out[2][HF_CARRY_BATCH * 4];
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t0 = off[c], T = off[c + 1] - t0;
    if (T <= 0) return;
    const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
    double* __restrict__ sp = s_pt[wave];
    double* __restrict__ so = s_out[wave];
    double v[4];
    if (wave == 0) {
        const uint32_t r0 = rec[t0];
        const DevRegion* __restrict__ R = &P->reg[REC_REGION(r0)];
        double sv = 0.0, E0[16];
        load_E<L>(E, k0, 0, 0, E0);
#pragma unroll
        for (int s = 0; s < 4; s++) { v[s] = E0[s] * R->trans[4][s]; sv += v[s]; }
#pragma unroll
        for (int s = 0; s < 4; s++) v[s] /= sv;
    } else {
        const DevRegion* __restrict__ R = &P->reg[REC_REGION(rec[t0 + T - 1])];
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
        __builtin_amdgcn_s_waitcnt(0);   // the wave's own LDS stores have landed (single wave per half: no barrier needed)
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            if (wave == 0) {
                for (int k = 0; k < n; k++) {
                    so[k * 4 + 0] = v[0]; so[k * 4 + 1] = v[1]; so[k * 4 + 2] = v[2]; so[k * 4 + 3] = v[3];
                    const double* __restrict__ M = sp + k * 16;
                    double u[4], su = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        double s = v[0] * M[j];
                        s = fma(v[1], M[4 + j], s); s = fma(v[2], M[8 + j], s); s = fma(v[3], M[12 + j], s);
                        u[j] = s; su += s;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = u[j] / su;
                }
            } else {
                for (int k = n - 1; k >= 0; k--) {
                    so[k * 4 + 0] = v[0]; so[k * 4 + 1] = v[1]; so[k * 4 + 2] = v[2]; so[k * 4 + 3] = v[3];
                    const double* __restrict__ M = sp + k * 16;
                    double u[4], su = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        double s = M[i * 4] * v[0];
                        s = fma(M[i * 4 + 1], v[1], s); s = fma(M[i * 4 + 2], v[2], s); s = fma(M[i * 4 + 3], v[3], s);
                        u[i] = s; su += s;
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = u[i] / su;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        {
            double* __restrict__ dst = (wave == 0 ? cf : cb) + (int64_t) (k0 + b0) * 4;
            for (int i = lane; i < n * 4; i += 64) dst[i] = so[i];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// k_fwd_tile: one wavefront per tile
// ------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256) k_fwd_tile(int ntiles, const TileDesc* __restrict__ td,
                                                  const uint32_t* __restrict__ rec,
                                                  const double* __restrict__ E, const double* __restrict__ Qs,
                                                  const DevParams* __restrict__ P,
                                                  const double* __restrict__ cf, double* __restrict__ F,
                                                  double* __restrict__ scale, double* __restrict__ tile_ll,
                                                  unsigned* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const TileDesc d = td[tile];
    const int64_t t0 = d.t0, T = d.T, base = d.base;
    const int64_t a = base + (int64_t) lane * L;
    // everything phase 3 needs is requested now, so its latency hides behind the scan
    uint32_t rr[L];
#pragma unroll
    for (int i = 0; i < L; i++) rr[i] = (a + i < T) ? rec[t0 + a + i] : 0u;
    double Ecur[16];
    load_E<L>(E, tile, lane, 0, Ecur);
    double carry[4];
    if (base == 0) { carry[0] = 1.0; carry[1] = 0.0; carry[2] = 0.0; carry[3] = 0.0; }  // (1,0,0,0)·A_first = start∘e
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) carry[j] = cf[(int64_t) tile * 4 + j];
    }
    unsigned bad = 0;
    // phase 1 + 2: exclusive prefix product over lanes
    M4 Q;
    load_lane_product(Q, Qs, tile, lane);
    if (base == 0 && lane == 0) {   // the chunk's first window is not in Q_l: prepend A_first = start∘e (row 0 only)
        double Tm[16];
        lds_Tm(s_tab, rr[0], Tm);
        M4 A, R;
#pragma unroll
        for (int k = 0; k < 16; k++) A.m[k] = Tm[k] * Ecur[k];
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
    double f[4];
    {
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
    // phase 3: replay this lane's windows in the reference's operation order (hmm.c:333-420)
    double ll = 0.0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        double Enext[16];
        if (i + 1 < L) load_E<L>(E, tile, lane, i + 1, Enext);     // next window's row is in flight during this one
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
                for (int p = 0; p < 4; p++) acc += (f[p] * Tm[p * 4 + s] * Ecur[p * 4 + s]);
                nf[s] = acc;
                sc += acc;
            }
            if (!REC_FIRST(r) && sc < 1e-50) bad |= HF_FLAG_SCALE;   // hmm.c:412-415
#pragma unroll
            for (int s = 0; s < 4; s++) f[s] = nf[s] / sc;
            ll += log(sc);                                            // hmm.c:428
            reinterpret_cast<double2*>(F + t * 4)[0] = make_double2(f[0], f[1]);
            reinterpret_cast<double2*>(F + t * 4)[1] = make_double2(f[2], f[3]);
            scale[t] = sc;
        }
        if (i + 1 < L) {
#pragma unroll
            for (int k = 0; k < 16; k++) Ecur[k] = Enext[k];
        }
    }
    for (int o = 32; o > 0; o >>= 1) ll += __shfl_down(ll, o);
    if (lane == 0) tile_ll[tile] = ll;
    if (bad) atomicOr(flags, bad);
}

// chunk log-likelihood = sum of its tiles' partial sums (fixed order)
__global__ void __launch_bounds__(64) k_chunk_ll(const int32_t* __restrict__ chunk_tile0, const double* __restrict__ tile_ll,
                                                 double* __restrict__ chunk_stats, int64_t V) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int k0 = chunk_tile0[c], nt = chunk_tile0[c + 1] - k0;
    double s = 0.0;
    for (int k = lane; k < nt; k += 64) s += tile_ll[k0 + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    // the chunk's statistics vector starts from zero every pass (HMM_resetEstimators, hmm.c:129-134)
    for (int64_t v = 1 + lane; v < V; v += 64) chunk_stats[(int64_t) c * V + v] = 0.0;
    if (lane == 0) chunk_stats[(int64_t) c * V] = s;
}

// ------------------------------------------------------------------------------------------
// k_bwd_tile: backward tile k owns windows i = base_k-1 .. base_k+64L-2 (i >= 0, i <= T-2); window i
// uses G_i = A_{i+1}:  b_i = G_i·b_{i+1} / scale_i.  The chunk's last window b_{T-1} = term/scale_{T-1} is
// written by the tile that contains window T-2 (or by tile 0 when T == 1).
// ------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256) k_bwd_tile(int ntiles, const TileDesc* __restrict__ td,
                                                  const uint32_t* __restrict__ rec,
                                                  const double* __restrict__ E, const double* __restrict__ Qs,
                                                  const DevParams* __restrict__ P,
                                                  const double* __restrict__ cb, const double* __restrict__ F,
                                                  const double* __restrict__ scale, double* __restrict__ B,
                                                  int8_t* __restrict__ label, unsigned* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    const TileDesc d = td[tile];
    const int64_t t0 = d.t0, T = d.T, base = d.base;
    const int64_t Tm1 = T - 1;                              // windows 0..T-2 have a recurrence step
    const int64_t a = base - 1 + (int64_t) lane * L;        // first window of this lane (may be -1)
    const int64_t tend = base - 1 + 64 * (int64_t) L;       // first window after the tile
    // requests for phase 3 go out first: window records of t+1, scales and the first emission row
    uint32_t rr[L];
    double scv[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
        const bool ok = a + i >= 0 && a + i < Tm1;
        rr[i] = ok ? rec[t0 + a + i + 1] : 0u;
        scv[i] = ok ? scale[t0 + a + i] : 1.0;
    }
    double Ecur[16];
    load_E<L>(E, tile, lane, L - 1, Ecur);
    unsigned bad = 0;
    const uint32_t rlast = rec[t0 + T - 1];
    const DevRegion* __restrict__ Rl = &P->reg[REC_REGION(rlast)];
    const double term = Rl->trans[0][4];
    double carry[4];
    const bool last_tile = tend >= Tm1;
    if (last_tile) {  // hmm.c:452-467: b_{T-1}[s] = M[s][End] / scale_{T-1}
        const double sc = scale[t0 + T - 1];
#pragma unroll
        for (int s = 0; s < 4; s++) carry[s] = Rl->trans[s][4] / sc;
        if (lane == 0) {
            const int64_t t = t0 + T - 1;
            double f[4];
#pragma unroll
            for (int s = 0; s < 4; s++) f[s] = F[t * 4 + s];
            reinterpret_cast<double2*>(B + t * 4)[0] = make_double2(carry[0], carry[1]);
            reinterpret_cast<double2*>(B + t * 4)[1] = make_double2(carry[2], carry[3]);
            label[t] = (int8_t) posterior_label(f, carry, sc);
        }
    } else {          // direction from k_carry, magnitude from sum_s f·b·scale = terminationProb at window `tend`
        const int64_t t = t0 + tend;
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < 4; s++) { carry[s] = cb[(int64_t) tile * 4 + s]; dot += F[t * 4 + s] * carry[s]; }
        const double k = term / (scale[t] * dot);
#pragma unroll
        for (int s = 0; s < 4; s++) carry[s] *= k;
    }
    // phase 1 + 2: exclusive SUFFIX product over lanes; lane windows i own A_{i+1}
    double b[4];
    {
        M4 Q;
        load_lane_product(Q, Qs, tile, lane);               // = product over windows a+1 .. a+L (window 0 excluded)
#pragma unroll
        for (int d2 = 1; d2 < 64; d2 <<= 1) {
            M4 Rgt, R;
            m4_shfl_down(Rgt, Q, d2);
            if (lane + d2 < 64) { m4_mul(R, Q, Rgt); Q = R; m4_renorm(Q); }
        }
        M4 X;
        m4_shfl_down(X, Q, 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double s = X.m[i * 4] * carry[0];
            s = fma(X.m[i * 4 + 1], carry[1], s); s = fma(X.m[i * 4 + 2], carry[2], s); s = fma(X.m[i * 4 + 3], carry[3], s);
            b[i] = s;
        }
    }
    const int64_t nxt = a + L;                              // window whose b this lane starts from
    if (lane == 63 || nxt >= Tm1) {
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] = carry[i];        // the carried vector itself
    } else {
        const int64_t t = t0 + nxt;
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < 4; s++) dot += F[t * 4 + s] * b[s];
        const double k = term / (scale[t] * dot);
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] *= k;
    }
    // phase 3: replay this lane's windows (decreasing i) in the reference's operation order (hmm.c:470-529)
    double fcur[4] = {0.0, 0.0, 0.0, 0.0};
    if (a + L - 1 >= 0 && a + L - 1 < Tm1) {
        const double2* __restrict__ fp = reinterpret_cast<const double2*>(F + (t0 + a + L - 1) * 4);
        const double2 f01 = fp[0], f23 = fp[1];
        fcur[0] = f01.x; fcur[1] = f01.y; fcur[2] = f23.x; fcur[3] = f23.y;
    }
#pragma unroll
    for (int i = L - 1; i >= 0; i--) {
        double Enext[16], fnext[4] = {0.0, 0.0, 0.0, 0.0};
        if (i > 0) {                                          // previous window's row and f are in flight during this one
            load_E<L>(E, tile, lane, i - 1, Enext);
            if (a + i - 1 >= 0 && a + i - 1 < Tm1) {
                const double2* __restrict__ fp = reinterpret_cast<const double2*>(F + (t0 + a + i - 1) * 4);
                const double2 f01 = fp[0], f23 = fp[1];
                fnext[0] = f01.x; fnext[1] = f01.y; fnext[2] = f23.x; fnext[3] = f23.y;
            }
        }
        if (a + i >= 0 && a + i < Tm1) {
            const int64_t t = t0 + a + i;
            double Tm[16];
            lds_Tm(s_tab, rr[i], Tm);                         // window t+1 = base + lane*L + i of this tile
            double nb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int p = 0; p < 4; p++) nb[p] += Tm[p * 4 + s] * Ecur[p * 4 + s] * b[s];
            const double sc = scv[i];
            if (sc < 1e-50) bad |= HF_FLAG_SCALE;             // hmm.c:521-524
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = nb[s] / sc;
            reinterpret_cast<double2*>(B + t * 4)[0] = make_double2(b[0], b[1]);
            reinterpret_cast<double2*>(B + t * 4)[1] = make_double2(b[2], b[3]);
            label[t] = (int8_t) posterior_label(fcur, b, sc);
        }
        if (i > 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) Ecur[k] = Enext[k];
#pragma unroll
            for (int k = 0; k < 4; k++) fcur[k] = fnext[k];
        }
    }
    if (bad) atomicOr(flags, bad);
}
