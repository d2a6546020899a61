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
// path (hf_estep.hip k_stats_tile), i.e. the results agree to rounding (tests: 1e-12 relative against each other, 1e-9
// against the oracle).  The order is fixed by the plan built in hf_create, so a run is reproducible bit for bit.
//
// Plan (static, hf_create): the pairs (t-1, t), t = 2..T-1 of every chunk, sorted by (region, row, t); a GROUP is up to 64
// consecutive pairs of one row, worked on by 16 lanes; a ROW SLOT is up to 4 consecutive groups of one row (a
// popular row has many slots: linearity again), worked on by one lane of k_row_stats; row slots are padded to whole
// wavefronts per region.
#pragma once
#include "hf_scan.h"

#define HF_GRP_PAIRS 64
#define HF_ROWSLOT_GROUPS 4

// PairIdx, RowSlot: hf_device.h

// ------------------------------------------------------------------------------------------
// k_pair_sums: sum over a group's pairs of the counts f[pre]·T[pre][s]·e[pre][s]·b[s] (before the division).
// 16 lanes per group (4 groups per wavefront), FOUR lanes per pair: lane q of a quad loads 16-byte piece q of the pair
// record (k_fb_tile RECS: f01 f23 b01 b23) — one load instruction covers 16 whole records — takes the f piece and the b
// piece of its 2x2 block of the 4x4 count matrix from its quad by DPP, and accumulates the block; the four quads of a
// group take the pairs round-robin and are summed by a fixed butterfly at the end.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v) {   // quad_perm within each group of four lanes
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) k_pair_sums(int n_groups, const PairIdx* __restrict__ pairs, const int32_t* __restrict__ grp_row,
                                                   const double* __restrict__ lutE, const DevParams* __restrict__ P,
                                                   const double* __restrict__ recs, double* __restrict__ grp_sums) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    fill_tab(P, s_tab);
    const int grp = (int) ((blockIdx.x * 256u + threadIdx.x) >> 4);
    if (grp >= n_groups) return;   // whole groups leave together; DPP and the butterfly stay inside a group
    const int ql = threadIdx.x & 3, qd = (threadIdx.x >> 2) & 3;
    const int pi = ql & 1, si = ql >> 1;                // this lane's block: pre in {2pi, 2pi+1}, s in {2si, 2si+1}
    int kk[4];                                          // state-major positions of the block's four entries
#pragma unroll
    for (int u = 0; u < 4; u++) kk[u] = HF_PS(2 * pi + (u & 1), 2 * si + (u >> 1));
    double ev[4];
    {
        const double* __restrict__ er = lutE + (int64_t) grp_row[grp] * 16;
#pragma unroll
        for (int u = 0; u < 4; u++) ev[u] = er[kk[u]];
    }
    const PairIdx* __restrict__ pp = pairs + (int64_t) grp * HF_GRP_PAIRS + qd;
    const double2* __restrict__ R2 = reinterpret_cast<const double2*>(recs);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int it0 = 0; it0 < HF_GRP_PAIRS / 4; it0 += 4) {
        PairIdx q[4];
        double2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            q[j] = pp[(it0 + j) * 4];
            v[j] = R2[(int64_t) (q[j].t < 0 ? 0 : q[j].t) * 4 + ql];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // quad_perm [0,1,0,1]: the f piece of this lane's rows; [2,2,3,3]: the b piece of its columns
            const double f0 = quad_perm_f64<0x44>(v[j].x), f1 = quad_perm_f64<0x44>(v[j].y);
            const double b0 = quad_perm_f64<0xFA>(v[j].x), b1 = quad_perm_f64<0xFA>(v[j].y);
            if (q[j].t >= 0) {
                const uint32_t r = q[j].rec;
                double tm[4];
                if (REC_REGCHG(r)) {                      // region change => 1/(S+1), hmm.c:398-400
#pragma unroll
                    for (int u = 0; u < 4; u++) tm[u] = 1.0 / (HF_NSTATES + 1);
                } else {
                    const double* __restrict__ tt = s_tab + REC_REGION(r) * HF_TAB_STRIDE + REC_VMASK(r) * 16;
#pragma unroll
                    for (int u = 0; u < 4; u++) tm[u] = tt[kk[u]];
                }
                acc[0] += f0 * tm[0] * ev[0] * b0;        // count before the division, hmm.c:612
                acc[1] += f1 * tm[1] * ev[1] * b0;
                acc[2] += f0 * tm[2] * ev[2] * b1;
                acc[3] += f1 * tm[3] * ev[3] * b1;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        double v = acc[u];
        v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16);
        acc[u] = v;
    }
    if (qd == 0) {
        double* __restrict__ dst = grp_sums + (int64_t) grp * 16;
#pragma unroll
        for (int u = 0; u < 4; u++) dst[kk[u]] = acc[u];
    }
}

// ------------------------------------------------------------------------------------------
// k_row_stats: one lane per row slot, one wavefront per 64 row slots of ONE region; the estimator updates of
// k_stats_tile with the slot's summed counts in the place of one window's counts.  Output: one partial vector per
// wavefront in StatAcc<KT> order (the format k_stats_tile writes per tile).
// The blocks after the first n_rw_blocks do a second job that has to happen once per pass anyway: the log-likelihood of
// every chunk (one wavefront per chunk, the same sum as k_chunk_stats) into element 0 of the chunk's vector.
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(256, 2) k_row_stats(int n_rowwaves, int n_rw_blocks, const int32_t* __restrict__ rw_region,
                                                      const RowSlot* __restrict__ slots, const double* __restrict__ grp_sums,
                                                      const RowSrc S, const DevParams* __restrict__ P, double* __restrict__ rw_stats,
                                                      int C, const int32_t* __restrict__ chunk_tile0,
                                                      const double* __restrict__ tile_ll, double* __restrict__ chunk_stats, int64_t V,
                                                      double* __restrict__ chunk_ll) {
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    constexpr int NS = 16 + 9 + 2;
    constexpr int RS = 65;
    extern __shared__ __attribute__((aligned(16))) double s_rows[];
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
    if (rw >= n_rowwaves) return;
    const int ncol = P->ncomp[3];
    const bool te = hf_err_is_truncexp(P);
    const int nrows = 3 * ncol > NS + 1 ? 3 * ncol : NS + 1;
    double* __restrict__ s_row = s_rows + wave * (nrows * RS);
    double* __restrict__ s_acc = s_row + lane;
    const DevRegion* __restrict__ R = &P->reg[rw_region[rw]];
    const RowSlot sl = slots[(int64_t) rw * 64 + lane];
    StatAccSmall a;
#pragma unroll
    for (int i = 0; i < NS; i++) reinterpret_cast<double*>(&a)[i] = 0.0;
    double c_wden = 0.0;
    for (int i = 0; i < 3 * ncol; i++) s_acc[i * RS] = 0.0;
    if (sl.row >= 0) {
        double cnt[16];
#pragma unroll
        for (int k = 0; k < 16; k++) cnt[k] = 0.0;
        const double2* __restrict__ gs = reinterpret_cast<const double2*>(grp_sums) + (int64_t) sl.g0 * 8;
        double2 gv[HF_ROWSLOT_GROUPS][8];          // all loads first, then the additions in plan order
#pragma unroll
        for (int g = 0; g < HF_ROWSLOT_GROUPS; g++)
#pragma unroll
            for (int k = 0; k < 8; k++) gv[g][k] = gs[(int64_t) (g < sl.ng ? g : 0) * 8 + k];
#pragma unroll
        for (int g = 0; g < HF_ROWSLOT_GROUPS; g++)
            if (g < sl.ng) {
#pragma unroll
                for (int k = 0; k < 8; k++) { cnt[2 * k] += gv[g][k].x; cnt[2 * k + 1] += gv[g][k].y; }
            }
        double Ev[16];
        load_row(reinterpret_cast<const double2*>(S.lutE) + (int64_t) sl.row * 8, Ev);
        const double2* __restrict__ crow = reinterpret_cast<const double2*>(S.lutC + ((int64_t) sl.row * 4) * S.K);
        const double x = (double) (sl.xpx & 0xff), px = (double) ((sl.xpx >> 8) & 0xff);
        double adj3[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            double adj[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                adj[p] = cnt[HF_PS(p, s)] / HF_TERMINATION_PROB;      // hmm.c:613-614
                a.trans[p * 4 + s] += adj[p];                         // hmm_utils.c:2010-2015
            }
            if (s == 3) {
#pragma unroll
                for (int p = 0; p < 4; p++) adj3[p] = adj[p];
            } else if (s == 0 && te) {                                // hmm_utils.c:1027-1034
#pragma unroll
                for (int p = 0; p < 4; p++) { a.te_num += adj[p] * x; a.te_den += adj[p]; }
            } else {                                                  // hmm_utils.c:812-839, one component
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int k = HF_PS(p, s);
                    const double alpha = P->alpha[p * 4 + s];
                    const double x_adj = alpha == 0.0 ? x : (x - alpha * px) / (1.0 - alpha);
                    const double w = adj[p] * Ev[k] / Ev[k];
                    a.g_mnum[s] += w * x_adj;
                    const double z = (x_adj - R->mean[s][0]) * (1.0 - alpha);
                    a.g_vnum[s] += w * z * z;
                    a.g_den[s] += w;
                }
            }
        }
        double xa[4], om[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const double alpha = P->alpha[p * 4 + 3];
            xa[p] = alpha == 0.0 ? x : (x - alpha * px) / (1.0 - alpha);
            om[p] = 1.0 - alpha;
        }
        double2 cu[KT][2];                        // every component's probabilities first: no load on the loop's path
#pragma unroll
        for (int cc = 0; cc < KT; cc++)
            if (cc < ncol) { cu[cc][0] = crow[cc * 2]; cu[cc][1] = crow[cc * 2 + 1]; }
#pragma unroll
        for (int cc = 0; cc < KT; cc++) {         // collapsed state, [component][previous state]
            if (cc >= ncol) continue;
            const double2 u01 = cu[cc][0], u23 = cu[cc][1];
            const double mu = R->mean[3][cc];
            double mnum = 0.0, vnum = 0.0, den = 0.0;
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
    // sums over the 64 lanes in lane order, as k_stats_tile: accumulator i is summed by lane i out of its LDS row
    double* __restrict__ dst = rw_stats + (int64_t) rw * NA;
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
        for (int i = lane; i < 3 * KT; i += 64) dst[NS + i] = 0.0;
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
}

// ------------------------------------------------------------------------------------------
// k_rows_total: one block.  Element 0 = sum of the chunks' log-likelihoods (k_row_stats) in k_reduce's order (the same bits as the
// per-chunk path); per region, the wavefront partials of k_row_stats summed in plan order by 1024/NA interleaved
// accumulators per element (fixed), expanded into the estimator layout of include/hmm_flagger_hip.h exactly as
// k_chunk_stats does; the vector is assembled in device memory and then copied to `out` (the pinned host block).
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(1024) k_rows_total(const int32_t* __restrict__ rw_off, const double* __restrict__ rw_stats,
                                                    const DevParams* __restrict__ P, const double* __restrict__ chunk_ll, int64_t C,
                                                    int64_t V, int Kctx, double* __restrict__ total, double* __restrict__ out,
                                                    const unsigned* __restrict__ flags) {
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    const int tid = threadIdx.x;
    constexpr int NQ = 960 / NA;                  // interleaved accumulators per element (the last wavefront sums the log-likelihoods)
    __shared__ double part[NQ][NA];
    __shared__ double red[NA];
    for (int64_t v = 1 + tid; v < V; v += 1024) total[v] = 0.0;
    if (tid >= 960) {   // k_reduce's order over the chunk list
        const int lane = tid - 960;
        double acc = 0.0;
        int64_t c = lane;
        for (; c + 64 * 3 < C; c += 64 * 4) {
            const double x0 = chunk_ll[c], x1 = chunk_ll[c + 64], x2 = chunk_ll[c + 128], x3 = chunk_ll[c + 192];
            acc += x0; acc += x1; acc += x2; acc += x3;
        }
        for (; c < C; c += 64) acc += chunk_ll[c];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) total[0] = acc;
    }
    __syncthreads();
    const int nreg = P->n_regions, ncol = P->ncomp[3];
    const bool te = hf_err_is_truncexp(P);
    const int64_t rstride = 24 * (int64_t) Kctx + 16;
    for (int r = 0; r < nreg; r++) {
        const int w0 = rw_off[r], w1 = rw_off[r + 1];
        if (w1 == w0) continue;
        const int q = tid / NA, i = tid % NA;
        if (q < NQ) {
            double v = 0.0;
            int k = w0 + q;
            for (; k + NQ * 3 < w1; k += NQ * 4) {   // 4 loads in flight, adds in plan order
                double xk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) xk[u] = rw_stats[(int64_t) (k + NQ * u) * NA + i];
#pragma unroll
                for (int u = 0; u < 4; u++) v += xk[u];
            }
            for (; k < w1; k += NQ) v += rw_stats[(int64_t) k * NA + i];
            part[q][i] = v;
        }
        __syncthreads();
        if (tid < NA) {
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < NQ; u++) v += part[u][tid];
            red[tid] = v;
        }
        __syncthreads();
        double* __restrict__ dst = total + 1 + r * rstride;
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
    __threadfence();
    __syncthreads();
    if (out != total)
        for (int64_t v = tid; v < V; v += 1024) out[v] = total[v];
    if (tid == 0 && flags) out[V] = (double) *flags;
}
