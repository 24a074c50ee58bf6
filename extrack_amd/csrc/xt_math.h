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

// exp(x), x <= 0, table-driven: x = (64 e + j) ln2/64 + r, |r| <= ln2/128, exp(x) = 2^e * T64[j] * p with p = P5(r).
// T64[j] = 2^(j/64) is part of the model blob (xt_tables.h).  |rel err| < 3e-16.  A NaN argument stays NaN.
// (Round 4 tried a 32-entry table - one entry per LDS bank pair, no conflicts - with the r^6 term folded into the coefficients: same speed
// in every kernel, but the two extra non-inline constants cost the wave-uniform threshold-fusion apply kernel 4 VGPRs, i.e. its second
// workgroup per CU: reverted.)
// Clamp of the exponent argument: n = 64 x / ln2 must fit int32 AND the sum of a sequence's exponent (>= XT_EMIN = -2^30) and n
// must not wrap: |n| < 2^30 <=> x > -1.16e7.  A Gaussian exponent below -1.1e7 is a jump of more than 4 600 standard deviations
// in one frame; such a sequence's weight is 2^(-1.5e7) instead of its true (even smaller) value.
#define XT_TCLAMP (-1.1e7)
XT_HD void xt_exp_tab(double x, double& p, int& j, int& e)
{
    x = x < XT_TCLAMP ? XT_TCLAMP : x;
    const double tk = xt_fma(x, 92.33248261689366, XT_MAGIC);
    const double kf = tk - XT_MAGIC;
    double r = xt_fma(kf, -0.010830424493178725, x);
    r = xt_fma(kf, -2.030704202170295e-10, r);
    double q = 8.33333333333333333333e-03;
    q = xt_fma(q, r, 4.16666666666666666667e-02);
    q = xt_fma(q, r, 1.66666666666666666667e-01);
    q = xt_fma(q, r, 0.5);
    q = xt_fma(q, r, 1.0);
    p = xt_fma(q, r, 1.0);
    const int n = xt_lo32(tk);
    j = n & 63;
    e = n >> 6;
}

// The same with a degree-4 polynomial (|rel err| < 5e-14): posterior weights only.
XT_HD void xt_exp_tab_fast(double x, double& p, int& j, int& e)
{
    x = x < XT_TCLAMP ? XT_TCLAMP : x;
    const double tk = xt_fma(x, 92.33248261689366, XT_MAGIC);
    const double kf = tk - XT_MAGIC;
    double r = xt_fma(kf, -0.010830424493178725, x);
    r = xt_fma(kf, -2.030704202170295e-10, r);
    double q = 4.16666666666666666667e-02;
    q = xt_fma(q, r, 1.66666666666666666667e-01);
    q = xt_fma(q, r, 0.5);
    q = xt_fma(q, r, 1.0);
    p = xt_fma(q, r, 1.0);
    const int n = xt_lo32(tk);
    j = n & 63;
    e = n >> 6;
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
