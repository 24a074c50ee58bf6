// Scalar math helpers for the track-likelihood kernels (MI355X / gfx950, fp64).
//
// The recursion is carried in the LINEAR domain with extended-range weights
//     value = m * 2^e        (m: double mantissa, not necessarily normalised; e: int32)
// instead of the reference's log domain (extrack/tracking.py:109-318 keeps log-probabilities and
// pays one exp per sequence and one log per fused sequence per step).  The only transcendental
// left per (sequence, step) is the exponential of the Gaussian quadratic form (xt_exp_tab), whose
// power-of-two part goes straight into the integer exponent, so nothing under- or overflows.
//
// Every function is usable from host code too: tests/emul runs the very same kernel body on CPU
// threads (test infrastructure only; the product never does).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define XT_HD __host__ __device__ __forceinline__
#else
#define XT_HD inline
#endif

#if defined(__clang__)
#define XT_UNROLL _Pragma("unroll")  // loops over register arrays: an index that is not a compile-time constant puts the array into scratch
#else
#define XT_UNROLL
#endif

#define XT_INL __attribute__((always_inline))  // lambdas of a kernel body: a closure that is not inlined lives in scratch memory

#define XT_EMIN (-(1 << 30))          // exponent of an exactly-zero weight
#define XT_LN2 0.693147180559945309417232121458
#define XT_LOG2PI 1.83787706640934548356065947281

#if defined(__HIP_DEVICE_COMPILE__)
XT_HD double xt_rcp(double x)
{
    // v_rcp_f64 seed + two Newton steps (4 FMA): <= 1 ulp for normal x, no scaling needed because
    // the arguments (variances, weight sums) are far from the fp64 range limits.
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
XT_HD double xt_frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }
XT_HD int xt_frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
XT_HD double xt_ldexp(double x, int e) { return __builtin_amdgcn_ldexp(x, e); }
XT_HD double xt_rint(double x) { return __builtin_rint(x); }
#else
XT_HD double xt_rcp(double x) { return 1.0 / x; }
XT_HD double xt_frexp_mant(double x)
{
    int e;
    return frexp(x, &e);
}
XT_HD int xt_frexp_exp(double x)
{
    int e;
    frexp(x, &e);
    return x == 0.0 ? 0 : e;
}
XT_HD double xt_ldexp(double x, int e) { return ldexp(x, e); }
XT_HD double xt_rint(double x) { return nearbyint(x); }
#endif

XT_HD double xt_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1 / x with ONE Newton step on the hardware seed (measured 2.2e-15 relative, tools/ubench/rcp_accuracy.hip): for quantities that end in a
// normalised posterior (tolerance 1e-9), not in the likelihood.
XT_HD double xt_rcp_fast(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / x;
#endif
}

// Keeps the instruction scheduler from interleaving the code before and after this point (independent per-direction blocks scheduled
// into each other multiply the live temporaries and spill).
XT_HD void xt_sched_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// Round-to-nearest-integer of |v| < 2^31 by the magic-number add: the integer lands in the low 32 bits of the double,
// so the int conversion is free and the fp64 rounding instruction is saved (v_rndne_f64 + v_cvt_i32_f64 -> one add).
#define XT_MAGIC 6755399441055744.0  // 1.5 * 2^52
XT_HD int xt_lo32(double t)
{
    union {
        double d;
        long long i;
    } u;
    u.d = t;
    return (int)u.i;
}

// exp(x), x <= 0, table-driven: x = (32 e + j) ln2/32 + r, |r| <= ln2/64, exp(x) = 2^e * T[j] * p with p = P5(r).
// T[j] = 2^(j/32), j < 32, is part of the model blob (xt_tables.h; the first 32 of its 64 table slots).  32 entries of 8 bytes cover every LDS
// bank exactly once (64 banks x 4 B), so the table lookups of a wavefront never conflict - equal indices broadcast.  With the 64-entry
// table of rounds 1 - 3 entries j and j + 32 shared a bank pair: 29 % of the headline kernel's LDS cycles were conflicts and a conflict-free
// timing experiment ran 4 % faster (round 4).  P5: Taylor coefficients with the r^6 term folded into c2 / c4 (Chebyshev economisation on
// |r| <= ln2/64): |rel err| < 1.5e-16 in exact arithmetic, 2.5e-16 evaluated in fp64 - the figures of the 64-entry Taylor version.
// A NaN argument stays NaN.
// Clamp of the exponent argument: n = 32 x / ln2 must fit int32 AND the sum of a sequence's exponent (>= XT_EMIN = -2^30) and n
// must not wrap: |n| < 2^30 <=> x > -2.3e7.  A Gaussian exponent below -1.1e7 is a jump of more than 4 600 standard deviations
// in one frame; such a sequence's weight is 2^(-1.5e7) instead of its true (even smaller) value.
#define XT_TCLAMP (-1.1e7)
#define XT_EXP_SCALE 46.16624130844683       // 32 / ln2
#define XT_EXP_HI (-0.02166084898635745)     // -ln2 / 32, high part (trailing zero bits: n * HI is exact) ...
#define XT_EXP_LO (-4.06140840434059e-10)    // ... and the rest
#define XT_EXP_MASK 31
#define XT_EXP_SHIFT 5
#define XT_EXP_C5 8.33333333333333333333e-03
#define XT_EXP_C4 0.041666911037706464       // 1/24 + (3/2) a^2 / 720,  a = ln2 / 64
#define XT_EXP_C3 1.66666666666666666667e-01
#define XT_EXP_C2 0.4999999999892509         // 1/2 - (9/16) a^4 / 720
XT_HD void xt_exp_tab(double x, double& p, int& j, int& e)
{
    x = x < XT_TCLAMP ? XT_TCLAMP : x;
    const double tk = xt_fma(x, XT_EXP_SCALE, XT_MAGIC);
    const double kf = tk - XT_MAGIC;
    double r = xt_fma(kf, XT_EXP_HI, x);
    r = xt_fma(kf, XT_EXP_LO, r);
    double q = XT_EXP_C5;
    q = xt_fma(q, r, XT_EXP_C4);
    q = xt_fma(q, r, XT_EXP_C3);
    q = xt_fma(q, r, XT_EXP_C2);
    q = xt_fma(q, r, 1.0);
    p = xt_fma(q, r, 1.0);
    const int n = xt_lo32(tk);
    j = n & XT_EXP_MASK;
    e = n >> XT_EXP_SHIFT;
}

// The same with a degree-4 polynomial (r^5 folded into c1 / c3: |rel err| < 8e-14): posterior weights only.
XT_HD void xt_exp_tab_fast(double x, double& p, int& j, int& e)
{
    x = x < XT_TCLAMP ? XT_TCLAMP : x;
    const double tk = xt_fma(x, XT_EXP_SCALE, XT_MAGIC);
    const double kf = tk - XT_MAGIC;
    double r = xt_fma(kf, XT_EXP_HI, x);
    r = xt_fma(kf, XT_EXP_LO, r);
    double q = 4.16666666666666666667e-02;
    q = xt_fma(q, r, 0.16666788852186565);   // 1/6 + (5/4) a^2 / 120
    q = xt_fma(q, r, 0.5);
    q = xt_fma(q, r, 0.9999999999641697);    // 1 - (5/16) a^4 / 120
    p = xt_fma(q, r, 1.0);
    const int n = xt_lo32(tk);
    j = n & XT_EXP_MASK;
    e = n >> XT_EXP_SHIFT;
}

// den^(-D/2) for a scalar variance (K == 1) given r = 1/den.
template <int D>
XT_HD double xt_pow_half(double r)
{
    if (D == 1) return sqrt(r);
    if (D == 2) return r;
    if (D == 3) return r * sqrt(r);
    double o = 1.0;
    for (int d = 0; d < D / 2; ++d) o *= r;
    return (D & 1) ? o * sqrt(r) : o;
}

// Extended-range accumulator: sum of terms m_i * 2^e_i, kept as m * 2^e with e = max e_i seen.
struct XtAcc {
    double m;
    int e;
    XT_HD void clear()
    {
        m = 0.0;
        e = XT_EMIN;
    }
    XT_HD void add(double m2, int e2)
    {
        if (m2 == 0.0) return;
        if (e2 > e) {
            m = xt_ldexp(m, e - e2) + m2;  // e - e2 may be hugely negative: ldexp saturates to 0
            e = e2;
        } else {
            m += xt_ldexp(m2, e2 - e);
        }
    }
};
