// hf_exp.h — exp() for double as the HOST computes it (round 6).  The reference's emission densities call libm's exp
// (hmm_utils.c:782, 945); the device's own exp (ocml) is a different algorithm and lands 1 ulp away from glibc's for ~6 % of the arguments
// of a pass.  Rounds 4-5 blamed that for the last-digit differences of ~1 in 10 --accelerate runs; this file was written to test it, and
// the answer is no (hf_device.h, profiles/r06_exp_fuzz.txt) — it is an opt-in build (-DHF_EXP_OCML=0) and a self-test.  It is a
// restatement of the algorithm glibc >= 2.28 uses (sysdeps/ieee754/dbl-64/e_exp.c, from ARM's Optimized Routines:
//     x = k ln2 / N + r, N = 128;   exp(x) = 2^(k/N) exp(r);   2^(k/N) from a table of 128 {rest, value} pairs;   exp(r) - 1 ~ degree-5 polynomial)
// with a fused multiply-add EXACTLY where the image's libm fuses — glibc selects its FMA build of the routine (__exp_fma, compiled with
// -mfma: the compiler contracted eight of its multiply-adds and left the others apart) on every CPU that has FMA: read off the
// instruction sequence of libm.so.6 (glibc 2.35) of this image, operation by operation, below.  One function for host and device (-ffp-contract=off on
// both sides: nothing is fused that is not written as fma): tests/test_exp_cpu.py compares the host build with libm's exp bit for bit
// over tens of millions of arguments, tests/test_estep_gpu.py the device's results with the host's.  On a host without FMA glibc runs the
// same algorithm unfused and the oracle's bits move with it; the product is then as far from that oracle as the device's own exp was (1 ulp).
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include "hf_exp_table.h"

#if defined(__HIPCC__) || defined(__CUDACC__)
#define HF_HD __host__ __device__
#else
#define HF_HD
#endif

HF_HD static inline uint64_t hf_exp_bits(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }
HF_HD static inline double hf_exp_double(uint64_t u) { double x; std::memcpy(&x, &u, 8); return x; }

HF_HD static inline double hf_exp(double x) {
    static const uint64_t T[256] = {HF_EXP_TABLE_WORDS};
    constexpr double InvLn2N = 0x1.71547652b82fep0 * 128, Shift = 0x1.8p52;
    constexpr double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    uint32_t abstop = (uint32_t) (hf_exp_bits(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {                     // |x| < 2^-54, or |x| >= 512, or not finite
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;                        // tiny: 1 + x is the correctly rounded result
        if (abstop >= 0x409u) {                                                    // |x| >= 1024
            if (hf_exp_bits(x) == hf_exp_bits(-INFINITY)) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;                                  // inf, NaN
            return (hf_exp_bits(x) >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;   // underflow / overflow (glibc: __math_uflow / __math_oflow)
        }
        abstop = 0;                                     // 512 <= |x| < 1024: the scale may leave the exponent range — handled at the end
    }
    // kd = round(x N / ln2) through the shift; r = x - kd ln2 / N in two pieces: three FMAs in libm's build
    const double z = std::fma(x, InvLn2N, Shift);
    const uint64_t ki = hf_exp_bits(z);
    const double kd = z - Shift;
    double r = std::fma(kd, NegLn2hiN, x);
    r = std::fma(kd, NegLn2loN, r);
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = ki << (52 - 7);
    const double tail = hf_exp_double(T[idx]);
    const uint64_t sbits = T[idx + 1] + top;
    // tmp = tail + r + r2 (C2 + r C3) + r2 r2 (C4 + r C5): libm's build fuses (C2 + r C3), (C4 + r C5), the product with r2 onto (tail + r)
    // and the product of r2 r2 onto that; tail + r, r r and r2 r2 stay plain operations
    const double r2 = r * r;
    const double p23 = std::fma(r, C3, C2);
    const double p45 = std::fma(r, C5, C4);
    const double lo = std::fma(p23, r2, tail + r);
    const double tmp = std::fma(r2 * r2, p45, lo);
    if (abstop == 0) {                                  // e_exp.c specialcase(): the result is huge or (sub)normal-small
        if ((ki & 0x80000000ull) == 0) {                // k > 0: the exponent of scale might have overflowed by <= 460
            const double scale = hf_exp_double(sbits - (1009ull << 52));
            return 0x1p1009 * std::fma(scale, tmp, scale);
        }
        const double scale = hf_exp_double(sbits + (1022ull << 52));   // k < 0: special care in the subnormal range
        const double st = scale * tmp;                  // (libm: a plain product here — it is used twice — and a plain sum)
        double y = scale + st;
        if (y < 1.0) {                                  // round to the right precision before scaling into the subnormal range
            double l2 = scale - y + st;
            const double hi = 1.0 + y;
            l2 = 1.0 - hi + y + l2;
            y = (hi + l2) - 1.0;
            if (y == 0.0) y = 0.0;                      // (no -0.0)
        }
        return 0x1p-1022 * y;
    }
    const double scale = hf_exp_double(sbits);
    return std::fma(scale, tmp, scale);                 // scale + scale tmp, fused in libm's build
}
