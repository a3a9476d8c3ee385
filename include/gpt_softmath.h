/*
 * gpt_softmath.h — the transcendental functions of the path-tracing hot path,
 * written out so that the host (x86-64, gcc/clang) and the device (gfx950,
 * hipcc) evaluate them with the SAME sequence of IEEE-754 operations.
 *
 * Why: the reference calls CUDA's libm (sinf, cosf, tan, atan, acos, __powf —
 * reference src/wrap.h:26-85, src/pathtracer.cu:107-138,187-197,
 * src/infinite.h:17-59).  Those implementations are not reproducible outside
 * nvcc, and glibc/ocml differ from each other in the last bit, which flips
 * discrete path decisions.  Every function below takes and
 * returns float, computes in double with only + - * / sqrt and integer bit
 * moves (all correctly rounded on both targets when FMA contraction is off),
 * and rounds once to float at the end, which makes the result the correctly
 * rounded float in all but ~1e-8 of cases and bit-identical on both targets in
 * all of them.
 *
 * Build rule: every translation unit that includes this header is compiled
 * with -ffp-contract=off.
 *
 * C99 / C++ / HIP compatible.
 */
#ifndef GPT_SOFTMATH_H
#define GPT_SOFTMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define GPT_HD __host__ __device__
#else
#define GPT_HD
#endif
#define GPT_INL static inline __attribute__((always_inline))

GPT_HD GPT_INL uint64_t gpt_d2u(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
GPT_HD GPT_INL double gpt_u2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
GPT_HD GPT_INL uint32_t gpt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
GPT_HD GPT_INL float gpt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

GPT_HD GPT_INL int gpt_isnanf(float x) { return (gpt_f2u(x) & 0x7fffffffu) > 0x7f800000u; }
GPT_HD GPT_INL int gpt_isinff(float x) { return (gpt_f2u(x) & 0x7fffffffu) == 0x7f800000u; }
GPT_HD GPT_INL float gpt_fabsf(float x) { return gpt_u2f(gpt_f2u(x) & 0x7fffffffu); }
GPT_HD GPT_INL double gpt_fabs(double x) { return gpt_u2d(gpt_d2u(x) & 0x7fffffffffffffffull); }

/* fminf/fmaxf with the IEEE minNum/maxNum rule CUDA uses: a NaN operand loses. */
GPT_HD GPT_INL float gpt_fminf(float a, float b) { return gpt_isnanf(a) ? b : (gpt_isnanf(b) ? a : (a < b ? a : b)); }
GPT_HD GPT_INL float gpt_fmaxf(float a, float b) { return gpt_isnanf(a) ? b : (gpt_isnanf(b) ? a : (a > b ? a : b)); }

/* ---- sin / cos kernels on |r| <= pi/4 (Taylor, truncation < 1e-16) ---- */
GPT_HD GPT_INL double gpt_ksin(double r)
{
    double z = r * r;
    double p = 7.6471637318198164759e-13;             /*  1/15! */
    p = p * z + -1.6059043836821614599e-10;           /* -1/13! */
    p = p * z + 2.5052108385441718775e-08;            /*  1/11! */
    p = p * z + -2.7557319223985890653e-06;           /* -1/9!  */
    p = p * z + 1.9841269841269841270e-04;            /*  1/7!  */
    p = p * z + -8.3333333333333333333e-03;           /* -1/5!  */
    p = p * z + 1.6666666666666666667e-01;            /*  1/3!  */
    return r - (r * z) * p;
}
GPT_HD GPT_INL double gpt_kcos(double r)
{
    double z = r * r;
    double p = 4.7794773323873852974e-14;             /*  1/16! */
    p = p * z + -1.1470745597729724714e-11;           /* -1/14! */
    p = p * z + 2.0876756987868098979e-09;            /*  1/12! */
    p = p * z + -2.7557319223985890653e-07;           /* -1/10! */
    p = p * z + 2.4801587301587301587e-05;            /*  1/8!  */
    p = p * z + -1.3888888888888888889e-03;           /* -1/6!  */
    p = p * z + 4.1666666666666666667e-02;            /*  1/4!  */
    return (1.0 - 0.5 * z) + (z * z) * p;
}

/* x = k*pi/2 + r, |r| <= pi/4 (+ rounding); valid for |x| < ~1e6, the hot
 * path only ever passes |x| < 8 (TWOPI*u, atan()+TWOPI). */
GPT_HD GPT_INL double gpt_rem_pio2(double x, int *quadrant)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;   /* 33 significant bits */
    const double pio2_lo = 6.07710050650619224932e-11;
    const double shifter = 6755399441055744.0;           /* 1.5 * 2^52 */
    double t = x * two_over_pi + shifter;                /* round-to-nearest-even integer */
    double k = t - shifter;
    /* k mod 2^32 sits in the low word of t (two's complement, |k| < 2^51): read it from there rather than convert k, so that
     * arguments far outside the supported range (and inf / NaN) still give the SAME bits on host and device - a float -> integer
     * conversion that overflows is undefined in C and saturates on gfx950.  Only bits 0 and 1 are ever looked at. */
    *quadrant = (int)(uint32_t)gpt_d2u(t);
    return (x - k * pio2_hi) - k * pio2_lo;
}

GPT_HD GPT_INL float gpt_sinf(float x)
{
    int q;
    double r = gpt_rem_pio2((double)x, &q);
    double s = (q & 1) ? gpt_kcos(r) : gpt_ksin(r);
    return (float)((q & 2) ? -s : s);
}

GPT_HD GPT_INL float gpt_cosf(float x)
{
    int q;
    double r = gpt_rem_pio2((double)x, &q);
    double c = (q & 1) ? gpt_ksin(r) : gpt_kcos(r);
    return (float)(((q + 1) & 2) ? -c : c);
}

GPT_HD GPT_INL float gpt_tanf(float x)
{
    int q;
    double r = gpt_rem_pio2((double)x, &q);
    double s = gpt_ksin(r), c = gpt_kcos(r);
    return (float)((q & 1) ? (-c / s) : (s / c));
}

/* atan on double; odd polynomial on |t| <= tan(pi/8), max rel. error 2e-15 */
GPT_HD GPT_INL double gpt_atan_d(double x)
{
    const double pio2 = 1.57079632679489661923;
    const double pio4 = 0.78539816339744830962;
    double ax = gpt_fabs(x);
    int inv = ax > 1.0;
    if (inv) ax = 1.0 / ax;                 /* 1/inf = 0 */
    int shift = ax > 0.41421356237309503;   /* tan(pi/8) */
    double t = shift ? (ax - 1.0) / (ax + 1.0) : ax;
    double z = t * t;
    double p = -0.025356815993101172;
    p = p * z + 0.050314584348260305;
    p = p * z + -0.06509439380222083;
    p = p * z + 0.07674395519100947;
    p = p * z + -0.09089648136299565;
    p = p * z + 0.11111058157195652;
    p = p * z + -0.14285713071163197;
    p = p * z + 0.19999999987550235;
    p = p * z + -0.33333333333302867;
    p = p * z + 1.0000000000000002;
    double a = t * p;
    if (shift) a = pio4 + a;
    if (inv) a = pio2 - a;
    return (gpt_d2u(x) >> 63) ? -a : a;
}

GPT_HD GPT_INL float gpt_atanf(float x) { return (float)gpt_atan_d((double)x); }

/* acos(x) = 2*atan(sqrt((1-x)/(1+x))); NaN outside [-1,1] like libm */
GPT_HD GPT_INL float gpt_acosf(float x)
{
    double d = (double)x;
    if (!(d >= -1.0 && d <= 1.0)) return gpt_u2f(0x7fc00000u);
    double q = (1.0 - d) / (1.0 + d);       /* x=-1 -> +inf -> atan = pi/2 */
    return (float)(2.0 * gpt_atan_d(__builtin_sqrt(q)));
}

/* natural log of a positive finite double (used by powf) */
GPT_HD GPT_INL double gpt_log_d(double x)
{
    const double ln2 = 6.93147180559945286227e-01;
    uint64_t u = gpt_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = gpt_u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
    if (m > 1.41421356237309515) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);       /* |s| <= 0.1716 */
    double z = s * s;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * ln2 + 2.0 * (s * p);
}

/* exp of a double with |y| < 700 (used by powf) */
GPT_HD GPT_INL double gpt_exp_d(double y)
{
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double shifter = 6755399441055744.0;
    double t = y * inv_ln2 + shifter;
    double k = t - shifter;
    double r = (y - k * ln2_hi) - k * ln2_lo;            /* |r| <= 0.3466 */
    double p = 1.0 / 6227020800.0;                        /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int64_t ki = (int64_t)k;
    return p * gpt_u2d((uint64_t)(ki + 1023) << 52);
}

/* powf for x > 0 (the gamma tonemap clamps x >= 1e-5 first,
 * reference src/pathtracer.cu:187-197) */
GPT_HD GPT_INL float gpt_powf(float x, float y)
{
    return (float)gpt_exp_d((double)y * gpt_log_d((double)x));
}

/* expf / logf for the participating-media code (src/medium.h:15,41-43, src/common.h:81-86, src/wrap.h:158-160).
 * expf: underflows to 0 below -104 (0x1p-150 rounds to 0), overflows to +inf above 88.73; NaN stays NaN.
 * logf: log(0) = -inf, log(x < 0) = NaN, log(+inf) = +inf; denormal floats are exact in double. */
GPT_HD GPT_INL float gpt_expf(float x)
{
    if (gpt_isnanf(x)) return x;
    if (x > 88.8f) return gpt_u2f(0x7f800000u);
    if (x < -104.f) return 0.f;
    return (float)gpt_exp_d((double)x);
}
GPT_HD GPT_INL float gpt_logf(float x)
{
    if (gpt_isnanf(x) || x < 0.f) return gpt_u2f(0x7fc00000u);
    if (x == 0.f) return gpt_u2f(0xff800000u);
    if (gpt_isinff(x)) return x;
    return (float)gpt_log_d((double)x);
}

#endif /* GPT_SOFTMATH_H */
