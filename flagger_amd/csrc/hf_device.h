// hf_device.h — device-side data layout and per-window math shared by all E-step kernels.
// gfx950 only.  Compiled with -ffp-contract=off: every fp64 operation below is written in the
// operand order of the reference (programs/submodules/hmm_utils/hmm_utils.c) so that the only
// differences from the CPU path are the last-ulp behaviour of exp()/log().
#pragma once
// -DHF_KSTAMP (profiling builds only): every kernel of the default pass leaves the wall clock (s_memrealtime, 100 MHz) of its block 0 in a ring
// of 64 passes — start-to-start distances of a pass's launches and of consecutive passes without a profiler in the way; hf_destroy appends the
// ring to $HF_KSTAMP_FILE (profiles/tools/r06_kstamp.py)
#ifdef HF_KSTAMP
__device__ unsigned long long g_kstamp[64 * 4 + 1];
#define KSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long p_ = g_kstamp[256]; if ((i) == 0) { p_ += 1; g_kstamp[256] = p_; } \
                                                                 g_kstamp[(p_ & 63) * 4 + (i)] = wall_clock64(); } } while (0)
#else
#define KSTAMP(i)
#endif
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hmm_flagger_hip.h"
#include "hf_exp.h"

#define HF_PI 3.14159              // common.h:15 (sic)
#define HF_TERMINATION_PROB 1e-4   // hmm_utils.c:2112

// device error flag bits
#define HF_FLAG_SCALE 1u
#define HF_FLAG_NAN 2u
#define HF_FLAG_REGION 4u
#define HF_FLAG_SYNC 8u     // one-launch segment kernel: a segment waited too long for another segment's product (hf_seg.h)

// ---- packed window record (4 B/window), built once by k_setup ----
//   bits 0..7  x      = (uint8_t) coverage             hmm.c:345,384
//   bits 8..15 region = annotation_flag >> 58          ptBlock.c:294-298
//   bits 16..18 validity mask: Dup / Col / End(Msj)    hmm_utils.c:2229-2254
//   bit 19     first window of its chunk
//   bit 20     region differs from the previous window hmm.c:398
//   bit 21     "slow": chunk-first, or contig-end factor beta_t != beta_star => private emission row (hf_scan.h)
#define REC_X(r) ((r) & 0xffu)
#define REC_REGION(r) (((r) >> 8) & 0xffu)
#define REC_VMASK(r) (((r) >> 16) & 0x7u)
#define REC_FIRST(r) (((r) >> 19) & 1u)
#define REC_REGCHG(r) (((r) >> 20) & 1u)
#define REC_SLOW(r) (((r) >> 21) & 1u)

// one record per tile of the scan kernels (built once in hf_create): removes the tile -> chunk -> offsets chain
// of dependent loads
// slow0 = position in the slow list of the tile's first slow window
struct TileDesc { long long t0; int T; int base; int chunk; int slow0; };
#define HF_AROW_CLASSES 9   // transition classes of an interior window: the 8 validity masks, 8 = region change (first window: 9)
// one segment of a chunk = one workgroup of the segment kernels (hf_seg.h), built once in hf_create
struct SegDesc {
    long long t0;            // global index of the segment's first window
    int n, L;                // windows of the segment, windows per lane
    int slot0, next_slot;    // first slot (scales, lane products: slot0 + step*64 + lane); first slot of the chunk's next segment
    int trash_pos;           // record position of the segment's own spare record: where lanes without a window store (hf_seg.h)
    int chunk_slow0;         // row of A (hf_seg.h) of the chunk's first window: start∘e
    int seg0, k, nseg;       // first segment of the chunk, this segment's position in it, segments of the chunk
    int reg_first, reg_last; // region of the chunk's first / last window
    int chunk;               // chunk index
    int spare_pos;           // record position whose f half takes f of the CHUNK's last window
    int ident_row;           // the table's identity row (behind the rows of A): what a lane multiplies by past its last window
};
static_assert(sizeof(SegDesc) == 64, "SegDesc is one 64-byte load");
// One row of the per-pass emission tables (hf_scan.h): static per context, built once by k_build_jobs.
// flags: 1 star (table key: the per-iteration constants of beta_star apply), 2 first (chunk-first window), 4 active; bits 8..: transition class of
// the job's row of A (hf_seg.h); row: index of the row in lutE / lutC units; bt: the window's own beta (beta_star for a table key)
struct TableJob { double x, px, bt; int64_t row; int32_t r; int32_t flags; };
// what the table work reads and writes: static per context (k_tables takes it by value)
struct TabWork {
    int32_t n_jobs, K;
    const TableJob* jobs;
    double *lutE, *lutC;
    double* lutA;                          // statistics by emission row / segment kernels: job j also writes row j of A (null: no rows of A)
};
// statistics by emission row (hf_rows.h): one pair (t-1, t) of the plan, one row slot
struct RowSlot { int32_t row, g0, ng, xpx; };                    // row < 0: padding; xpx = x | x_prev << 8

// Layout of the forward / backward arrays f, b (double2 units): tile-major, lane-minor —
//   slot(tile, lane, j, h) = ((tile*L + j)*2 + h)*64 + lane     window = tile base + lane*L + j, h: states (0,1) / (2,3)
// so that the 64 lanes of the wavefront that owns a tile read / write 1 KiB contiguous per instruction.
template <int L>
__device__ __forceinline__ int64_t fb_slot(int64_t tile, int lane, int j, int h) { return ((tile * L + j) * 2 + h) * 64 + lane; }
// the same for window w (index inside its chunk) of a chunk whose first tile is tile0
template <int L>
__device__ __forceinline__ int64_t fb_slot_w(int tile0, int64_t w, int h) {
    const int64_t TW = 64 * L;
    const int rem = (int) (w % TW);
    return fb_slot<L>(tile0 + w / TW, rem / L, rem % L, h);
}

// ---- divisions that share a denominator ---------------------------------------------------------------------
// a / d as the compiler emits it for fp64 (LowerFDIV64: v_div_scale x2, v_rcp_f64, two Newton steps, quotient,
// residual, v_div_fmas, v_div_fixup) spends most of its instructions on the denominator.  When many numerators
// meet the same denominator, the reciprocal refinement is done once (prediv) and each quotient costs a multiply and
// two FMAs (divp) — the SAME operations in the same order, so the result is bit-identical to a / d whenever
// v_div_scale would not rescale and v_div_fixup would not intervene: d and a normal, far from the exponent limits.
// div_operand_safe() is the guard on EACH operand (0 or 2^-380 <= v <= 2^380, v >= 0): then the exponents differ by
// less than the 768 at which v_div_scale steps in and the quotient stays normal; callers fall back to a / d otherwise.
struct PreDiv { double d, r; };
__device__ __forceinline__ PreDiv prediv(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = fma(-d, r0, 1.0);
    const double r1 = fma(r0, e0, r0);
    const double e1 = fma(-d, r1, 1.0);
    PreDiv p;
    p.d = d;
    p.r = fma(r1, e1, r1);
    return p;
}
__device__ __forceinline__ double divp(double a, const PreDiv& p) {
    const double q0 = a * p.r;
    const double res = fma(-p.d, q0, a);
    return fma(res, p.r, q0);
}
__device__ __forceinline__ bool div_operand_safe(double v) {
    const unsigned h = (unsigned) __double2hiint(v);
    return v == 0.0 || (h - 0x28300000u) <= (0x57B00000u - 0x28300000u);
}

#define HF_TABLE_MAX_ITEMS 80   // 1 + 4 + 4 + 4*HF_MAXCOMP = 73 at most

struct DevRegion {
    double trans[5][5];                 // Transition.matrix (row 4 Start, column 4 End)
    double tcond[8][16];                // Transition_getProbConditional per validity mask, [pre*4+s]
    double lambda, trunc_point;         // TruncExponential
    double mean[HF_NSTATES][HF_MAXCOMP];
    double var[HF_NSTATES][HF_MAXCOMP];
    double weight[HF_NSTATES][HF_MAXCOMP];
    // iteration constants for windows whose contig-end factor is the dominant value beta_star (all interior
    // windows): each is the reference's own sub-expression, evaluated once on the host instead of per window
    double m1[HF_NSTATES][4][HF_MAXCOMP];   // (1 - alpha_u) * mean_c                 hmm_utils.c:775
    double gvar[HF_NSTATES][HF_MAXCOMP];    // var_c * beta_star                      hmm_utils.c:777-778
    double gnorm[HF_NSTATES][HF_MAXCOMP];   // w_c / sqrt(gvar * 2 * PI)              hmm_utils.c:781
    double te_lam, te_den;                  // lambda / beta_star, 1 - exp(-te_lam * beta_star * trunc)  hmm_utils.c:942-946
};

struct DevParams {
    int32_t model_type, n_regions;
    int32_t ncomp[HF_NSTATES];
    int32_t nuniq[HF_NSTATES];          // distinct alpha values in column s
    int32_t umap[16];                   // [pre*4+s] -> index of alpha[pre][s] among column s' distinct values
    double ualpha[HF_NSTATES][4];       // [s][u]
    double alpha[16];                   // [pre*4+s]
    double beta_star;                   // value of beta_t for every window away from contig ends (hmm.c:301-316)
    // k_tables' work list of one emission row: item i evaluates state item_s[i] for its item_u[i]-th distinct alpha,
    // component item_c[i] (one exp each); item_base[s*4 + u] = the item of (s, u, component 0)
    int32_t n_items;
    int32_t item_base[16];
    uint8_t item_s[HF_TABLE_MAX_ITEMS], item_u[HF_TABLE_MAX_ITEMS], item_c[HF_TABLE_MAX_ITEMS];
    DevRegion reg[1];                   // n_regions entries
};

// ---- the parameter block of a one-region model through the KERNEL ARGUMENTS of the pass's first kernel ----
// Per EM step the host used to enqueue a copy of the parameter block (a blit kernel: 2.5-3 us, a kernel boundary and one more
// API call on the step's critical path) before k_tables.  For one region the bytes that are in use (the components in use, not
// HF_MAXCOMP of them) fit the 4 KiB of kernel arguments: the host sends every 8-byte word in use with its place in the DevParams image,
// every block of k_tables rebuilds the image in LDS, block 0 also in global memory for the kernels after it.
#define HF_KP_MAX_WORDS 384
struct KParams {
    int32_t n_words, pad;
    uint16_t idx[HF_KP_MAX_WORDS];     // where word i goes: 8-byte words of the DevParams image
    double data[HF_KP_MAX_WORDS];
};
static_assert(sizeof(KParams) <= 3860, "k_tables' other arguments need the rest of the 4 KiB");
static_assert(sizeof(DevParams) % 8 == 0 && offsetof(DevParams, reg) % 8 == 0, "the image is copied in 8-byte words");

// all threads of a block: the image out of the kernel arguments (dst: LDS, or global memory)
__device__ __forceinline__ void kparams_expand(const KParams& kp, double* __restrict__ dst, int tid, int nthreads) {
    for (int i = tid; i < kp.n_words; i += nthreads) dst[kp.idx[i]] = kp.data[i];
}

__device__ __forceinline__ bool hf_err_is_truncexp(const DevParams* __restrict__ P) {
    return P->model_type == HF_MODEL_TRUNC_EXP_GAUSSIAN;
}

// exp() of the emission densities.  Round 6 restated glibc's algorithm bit for bit (hf_exp.h: 0 of 63 M arguments differ from
// the host's libm, on the device too) to test the standing explanation of the --accelerate residue — "the device's exp is 1 ulp from glibc's
// for 6 % of the arguments and SQUAREM amplifies it".  It is NOT the cause: of 800 accelerated fuzz runs 67 differ from the oracle command
// line with the host's exp against 69 with the device library's, the same seeds (profiles/r06_exp_fuzz.txt) — what SQUAREM amplifies is the
// order of the additions in the scans, which no exp can change.  The restatement costs k_tables 0.5-5 us per pass (a table look-up per
// value; profiles/r06_ab_exp.txt) and buys nothing, so the device library's exp stays; -DHF_EXP_OCML=0 builds the other one
// (profiles/tools/build_variants.sh), and the self-test hook keeps checking it on the device.
#ifndef HF_EXP_OCML
#define HF_EXP_OCML 1
#endif
__device__ __forceinline__ double hf_emit_exp(double x) {
#if HF_EXP_OCML
    return exp(x);
#else
    return hf_exp(x);
#endif
}

// hmm_utils.c:941-947 TruncExponential_getProb
__device__ __forceinline__ double hf_trunc_exp(double lambda, double trunc_point, double x, double beta) {
    double lam = lambda / beta;
    double b = beta * trunc_point;
    if (trunc_point < x) return 0.0;
    return lam * hf_emit_exp(-lam * x) / (1 - hf_emit_exp(-lam * b));
}

// one mixture component, hmm_utils.c:775-790; sets *nan when the reference would exit
__device__ __forceinline__ double hf_gauss_comp(double mu, double var_c, double w, double x, double pre_x,
                                                double alpha, double beta, unsigned* nan) {
    double mean = (1 - alpha) * mu + alpha * pre_x;
    mean *= beta;
    double var = var_c * beta;
    double d = x - mean;
    double p = w / (sqrt(var * 2 * HF_PI)) * hf_emit_exp(-0.5 * (d * d) / var);
    if (p != p) *nan |= HF_FLAG_NAN;
    if (p < 1e-40) p = 1e-40;
    return p;
}

// same component with the beta_star constants: m1 = (1-alpha)*mu, gvar = var*beta, gnorm = w/sqrt(gvar*2*PI)
__device__ __forceinline__ double hf_gauss_comp_star(double m1, double gvar, double gnorm, double x, double pre_x,
                                                     double alpha, double beta, unsigned* nan) {
    double mean = m1 + alpha * pre_x;
    mean *= beta;
    const double d = x - mean;
    double p = gnorm * hf_emit_exp(-0.5 * (d * d) / gvar);
    if (p != p) *nan |= HF_FLAG_NAN;
    if (p < 1e-40) p = 1e-40;
    return p;
}

__device__ __forceinline__ double hf_trunc_exp_star(const DevRegion* __restrict__ R, double x) {
    if (R->trunc_point < x) return 0.0;
    return R->te_lam * hf_emit_exp(-R->te_lam * x) / R->te_den;
}

// Gaussian_getProb, hmm_utils.c:753-758 (sum over components in index order)
__device__ __forceinline__ double hf_gauss_sum(const DevRegion* __restrict__ R, int s, int ncomp, double x,
                                               double pre_x, double alpha, double beta, unsigned* nan) {
    double tot = 0.0;
    for (int c = 0; c < ncomp; c++)
        tot += hf_gauss_comp(R->mean[s][c], R->var[s][c], R->weight[s][c], x, pre_x, alpha, beta, nan);
    return tot;
}

// E[pre][s] = e_s(x | px, alpha[pre][s], bt) for region parameters R (A8-A10), evaluated once per distinct alpha
// of a column; Err as trunc-exp ignores alpha.  `first` = chunk-first window: e_s(x; alpha = 0, preX = 0) in row
// pre = 0 and zeros elsewhere (hmm.c:338-352).  Direct per-window evaluation (HF_ALGO_SEQ); the scan path builds the
// same values per table row in k_tables.
__device__ __forceinline__ void hf_emit_values(const DevParams* __restrict__ P, const DevRegion* __restrict__ R, double x,
                                               double px, bool first, double bt, double out[16], unsigned* nan) {
    const bool te = hf_err_is_truncexp(P);
    if (first) {
#pragma unroll
        for (int k = 0; k < 16; k++) out[k] = 0.0;
        out[0] = te ? hf_trunc_exp(R->lambda, R->trunc_point, x, bt)
                    : hf_gauss_sum(R, 0, P->ncomp[0], x, 0.0, 0.0, bt, nan);
        for (int s = 1; s < 4; s++) out[s] = hf_gauss_sum(R, s, P->ncomp[s], x, 0.0, 0.0, bt, nan);
    } else {
        for (int s = 0; s < 4; s++) {
            double val[4];
            if (s == 0 && te) {
                const double v = hf_trunc_exp(R->lambda, R->trunc_point, x, bt);
                val[0] = v; val[1] = v; val[2] = v; val[3] = v;
            } else {
                const int nu = P->nuniq[s], nc = P->ncomp[s];
                val[0] = val[1] = val[2] = val[3] = 0.0;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (u < nu)
                        val[u] = hf_gauss_sum(R, s, nc, x, px, P->ualpha[s][u], bt, nan);
            }
#pragma unroll
            for (int pre = 0; pre < 4; pre++) {
                const int u = (s == 0 && te) ? 0 : P->umap[pre * 4 + s];
                out[pre * 4 + s] = u == 0 ? val[0] : u == 1 ? val[1] : u == 2 ? val[2] : val[3];
            }
        }
    }
}

// transition row table for one window: region change => 1/(S+1) (hmm.c:398-400)
__device__ __forceinline__ void load_T(const DevParams* __restrict__ P, uint32_t r, double Tm[16]) {
    if (REC_REGCHG(r)) {
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = 1.0 / (HF_NSTATES + 1);
    } else {
        const double* __restrict__ src = P->reg[REC_REGION(r)].tcond[REC_VMASK(r)];
#pragma unroll
        for (int k = 0; k < 16; k++) Tm[k] = src[k];
    }
}

__device__ __forceinline__ int posterior_label(const double f[4], const double b[4], double sc) {
    // hmm.c:671-692 + common.c:292-304 (strict >, first maximum wins)
    double p[4], total = 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) { p[s] = f[s] * b[s] * sc; total += p[s]; }
#pragma unroll
    for (int s = 0; s < 4; s++) p[s] /= total;
    double mx = p[0]; int idx = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) if (mx < p[s]) { mx = p[s]; idx = s; }
    return idx;
}

// Checksum of a polled result block (hf_estep.hip wait_total): wrapping sum of bit pattern x position weight.  Position-
// dependent on purpose: the estimator layout repeats values (mean.den == var.den == weight.num), and a plain XOR is blind
// to two equal stale words.
__host__ __device__ inline unsigned long long hf_cks_term(unsigned long long bits, long long index) {
    return bits * ((0x9E3779B97F4A7C15ull * (unsigned long long) (index + 1)) | 1ull);
}
