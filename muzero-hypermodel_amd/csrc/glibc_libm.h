// glibc_libm.h -- glibc's double-precision `log` and `pow`, restated operation for operation so that the GPU
// reproduces the values numpy's legacy gamma sampler gets from libm on the host.
//
// Why: root exploration noise is numpy.random.dirichlet (reference self_play.py:468-477), i.e. legacy_standard_gamma,
// i.e. `log` and `pow` of libm.so.6 on random arguments.  MuZero's priors after noise feed fp64 UCB scores that are
// compared with ==, so the noise has to be the reference's to the last bit -- an accurate logarithm is not enough, it
// has to be glibc's.  glibc >= 2.28 uses the ARM optimized-routines algorithms (sysdeps/ieee754/dbl-64/e_log.c,
// e_pow.c, e_exp.c): table-driven range reduction + polynomial.  On x86-64 CPUs with FMA (every host of an MI355X)
// the dynamic loader picks the `__log_fma` / `__pow_fma` builds, compiled with -mfma and GCC's default
// -ffp-contract=fast, so WHICH multiply-adds are fused is part of the function.  The fma() calls below are exactly
// those of that build (read off GCC 11's optimised GIMPLE of the same source; every other operation is a separate
// IEEE operation, and this library is compiled with -ffp-contract=off).  tests/test_glibc_libm.py pins both the
// tables (glibc_libm_tables.inc, tools/extract_libm_tables.py) and the operation order to the libm of the machine
// the tests run on: bit-equal on tens of millions of arguments over the sampler's domain; tests/test_gpu_dirichlet.py
// does the same for the device build.
//
// Scope: finite positive arguments, which is all the sampler produces (log: 1 - U, (1 - U) / shape, U, r2; pow: bases
// in [0, inf), exponents 1 / shape > 0).  Other arguments return NaN (never reached; flagged by the callers' tests).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MZ_LIBM_HD __host__ __device__
#else
#define MZ_LIBM_HD
#endif

namespace mz {
namespace libm {

#define MZ_LIBM_TABLE(type, name, n) static const type host_##name[n]
#include "glibc_libm_tables.inc"
#undef MZ_LIBM_TABLE
#if defined(__HIPCC__)
#define MZ_LIBM_TABLE(type, name, n) static __device__ const type dev_##name[n]
#include "glibc_libm_tables.inc"
#undef MZ_LIBM_TABLE
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MZ_LIBM(name) dev_##name
#else
#define MZ_LIBM(name) host_##name
#endif

MZ_LIBM_HD inline uint64_t bits_of(double x) { return __builtin_bit_cast(uint64_t, x); }
MZ_LIBM_HD inline double double_of(uint64_t u) { return __builtin_bit_cast(double, u); }
MZ_LIBM_HD inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// e_log.c __log (the __FP_FAST_FMA build)
MZ_LIBM_HD inline double glibc_log(double x) {
    const double* A = MZ_LIBM(log_poly);
    const double* B = MZ_LIBM(log_poly1);
    uint64_t ix = bits_of(x);
    const uint32_t top = static_cast<uint32_t>(ix >> 48);
    const uint64_t lo_bound = 0x3fee000000000000ull;   // 1.0 - 0x1p-4
    const uint64_t hi_bound = 0x3ff1090000000000ull;   // 1.0 + 0x1.09p-4
    if (ix - lo_bound < hi_bound - lo_bound) {
        // close to 1: a degree-12 polynomial in r = x - 1 with the leading terms in double-double
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double p0 = fma_(B[3], r2, fma_(B[2], r, B[1]));
        const double p1 = fma_(B[6], r2, fma_(B[5], r, B[4]));
        const double p2 = fma_(B[10], r3, fma_(B[9], r2, fma_(B[8], r, B[7])));
        const double p = fma_(fma_(p2, r3, p1), r3, p0);
        const double big = fma_(r, 0x1p27, r);
        const double rhi = fma_(-r, 0x1p27, big);
        const double rlo = r - rhi;
        const double sq = rhi * rhi;
        const double hi = fma_(sq, B[0], r);
        double lo = fma_(sq, B[0], r - hi);
        lo = fma_(B[0] * rlo, r + rhi, lo);
        return hi + fma_(p, r3, lo);
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        // x < 0x1p-1022, inf or nan: only a subnormal positive x is a valid argument here
        if (ix * 2 == 0 || (top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return double_of(0x7ff8000000000000ull);
        ix = bits_of(x * 0x1p52);
        ix -= 52ull << 52;
    }
    // x = 2^k z, z in [OFF, 2 OFF); the table entry holds 1/c and log(c) for c near the centre of z's subinterval
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = static_cast<int>((tmp >> (52 - 7)) & 127u);
    const int k = static_cast<int>(static_cast<int64_t>(tmp) >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = MZ_LIBM(log_tab)[2 * i];
    const double logc = MZ_LIBM(log_tab)[2 * i + 1];
    const double z = double_of(iz);
    const double r = fma_(z, invc, -1.0);
    const double kd = static_cast<double>(k);
    const double w = fma_(MZ_LIBM(log_ln2)[0], kd, logc);
    const double hi = r + w;
    const double lo = fma_(MZ_LIBM(log_ln2)[1], kd, (w - hi) + r);
    const double r2 = r * r;
    const double q = fma_(fma_(A[4], r, A[3]), r2, fma_(A[2], r, A[1]));
    return fma_(r * r2, q, fma_(A[0], r2, lo)) + hi;
}

// e_pow.c log_inline: log(x) as hi + *tail with about 68 bits
MZ_LIBM_HD inline double pow_log(uint64_t ix, double* tail) {
    const double* A = MZ_LIBM(pow_poly);
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = static_cast<int>((tmp >> (52 - 7)) & 127u);
    const int k = static_cast<int>(static_cast<int64_t>(tmp) >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double z = double_of(iz);
    const double kd = static_cast<double>(k);
    const double invc = MZ_LIBM(pow_tab)[4 * i];
    const double logc = MZ_LIBM(pow_tab)[4 * i + 2];
    const double logctail = MZ_LIBM(pow_tab)[4 * i + 3];
    const double r = fma_(z, invc, -1.0);
    const double t1 = fma_(kd, MZ_LIBM(pow_ln2)[0], logc);
    const double t2 = r + t1;
    const double lo1 = fma_(kd, MZ_LIBM(pow_ln2)[1], logctail);
    const double lo2 = r + (t1 - t2);
    const double ar = r * A[0];
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = fma_(ar, r, -ar2);
    const double lo4 = ar2 + (t2 - hi);
    const double p = fma_(ar2, fma_(ar2, fma_(r, A[6], A[5]), fma_(r, A[4], A[3])), fma_(r, A[2], A[1]));
    const double lo = fma_(ar3, p, lo4 + (lo3 + (lo1 + lo2)));
    const double y = hi + lo;
    *tail = (hi - y) + lo;
    return y;
}

// e_pow.c exp_inline + specialcase: exp(x + xtail) for the pow kernel (sign_bias = 0)
MZ_LIBM_HD inline double pow_exp(double x, double xtail) {
    uint32_t abstop = static_cast<uint32_t>(bits_of(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {   // |x| < 2^-54 or >= 512
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
        if (abstop >= 0x409u)                   // |x| >= 1024: underflow / overflow
            return (bits_of(x) >> 63) ? 0.0 : double_of(0x7ff0000000000000ull);
        abstop = 0;                             // large |x|: the scale needs care, below
    }
    const double shift = MZ_LIBM(exp_head)[1];
    double kd = fma_(x, MZ_LIBM(exp_head)[0], shift);
    const uint64_t ki = bits_of(kd);
    kd = kd - shift;
    double r = fma_(kd, MZ_LIBM(exp_head)[3], fma_(kd, MZ_LIBM(exp_head)[2], x));
    r = xtail + r;
    const uint64_t idx = 2 * (ki & 127u);
    const uint64_t top = ki << (52 - 7);
    const double tail = double_of(MZ_LIBM(exp_tab)[idx]);
    const uint64_t sbits = MZ_LIBM(exp_tab)[idx + 1] + top;
    const double r2 = r * r;
    const double* C = MZ_LIBM(exp_poly);
    const double low = fma_(r2, fma_(r, C[1], C[0]), r + tail);
    const double tmp = fma_(r2 * r2, fma_(r, C[3], C[2]), low);
    if (abstop == 0) {
        if ((ki & 0x80000000u) == 0) {          // k > 0: the exponent of scale may have overflowed
            const double scale = double_of(sbits - (1009ull << 52));
            return fma_(tmp, scale, scale) * 0x1p1009;
        }
        // k < 0: the result may be subnormal; round once, at the right precision
        const uint64_t sb = sbits + (1022ull << 52);
        const double scale = double_of(sb);
        const double prod = tmp * scale;
        double y = scale + prod;
        const double mag = y < 0.0 ? -y : y;
        if (mag < 1.0) {
            const double one = y < 0.0 ? -1.0 : 1.0;
            double lo = prod + (scale - y);
            const double hi = y + one;
            lo = lo + (y + (one - hi));
            y = (hi + lo) - one;
            if (y == 0.0) y = double_of(sb & 0x8000000000000000ull);
        }
        return y * 0x1p-1022;
    }
    const double scale = double_of(sbits);
    return fma_(tmp, scale, scale);
}

// e_pow.c __pow for x >= 0 finite, y > 0 finite with 2^-65 <= y < 2^63
MZ_LIBM_HD inline double glibc_pow(double x, double y) {
    uint64_t ix = bits_of(x);
    const uint32_t topx = static_cast<uint32_t>(ix >> 52);
    const uint32_t topy = static_cast<uint32_t>(bits_of(y) >> 52);
    if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) {
        if (ix == 0 && topy - 0x3beu < 0x43eu - 0x3beu) return 0.0;    // pow(+0, y > 0)
        if (topx != 0 || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) return double_of(0x7ff8000000000000ull);
        ix = bits_of(x * 0x1p52);                                      // subnormal x
        ix &= 0x7fffffffffffffffull;
        ix -= 52ull << 52;
    }
    double lo;
    const double hi = pow_log(ix, &lo);
    const double ehi = y * hi;
    const double elo = fma_(y, lo, fma_(y, hi, -ehi));
    return pow_exp(ehi, elo);
}

}  // namespace libm
}  // namespace mz
