// detmath.h — deterministic FP64 elementary functions shared by host and gfx950 code.
//
// Why this exists: the reference filter calls glibc cos/sin/exp/log/fmod
// (slam/raycast.cpp:147-148, slam/mcl.cpp:60-61,73,75, slam/util.cpp:11,47).  glibc and the ROCm
// device libm differ in the last bit, and a 1-ulp change of cos/sin can move a ray sample across a
// cell edge.  Every function here is built ONLY from IEEE-754 correctly rounded operations
// (+ - * / sqrt fma) in a fixed order, so the same source gives the same bits on x86 (g++,
// -ffp-contract=off) and on gfx950 (hipcc, -ffp-contract=off).  Accuracy is < 1 ulp
// (tests/test_detmath.py measures it against 80-bit long double).
//
// Algorithms: the classic argument-reduction + minimax-polynomial schemes published with Sun's
// fdlibm (coefficients are the published fdlibm constants, written here as bit patterns), in the
// branch-light arrangement also used by musl/FreeBSD.  No code is taken from the reference repo.
//
// All translation units that include this header MUST be compiled with -ffp-contract=off.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DM_HD __host__ __device__ static inline
#else
#define DM_HD static inline
#endif

DM_HD uint64_t dm_bits(double x) { return __builtin_bit_cast(uint64_t, x); }
DM_HD double dm_from_bits(uint64_t b) { return __builtin_bit_cast(double, b); }
DM_HD double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
DM_HD double dm_sqrt(double x) { return __builtin_sqrt(x); }  // IEEE correctly rounded on both sides
DM_HD double dm_fabs(double x) { return __builtin_fabs(x); }
DM_HD uint32_t dm_hi(double x) { return (uint32_t)(dm_bits(x) >> 32); }

// DM_KC(bits, dep): the constant with these bits.  On the device it is materialised where it is used (two 32-bit moves
// that formally depend on `dep`, a value of the surrounding computation) instead of being hoisted out of every loop into a
// register pair for the whole kernel: the polynomial constants of this header otherwise occupy 32 vector registers of
// the ray kernels, which costs them a wavefront per SIMD.  Same value on both sides.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DM_NO_REMAT)
template <uint64_t B>
__device__ static inline double dm_kc(double dep)
{
    uint32_t lo, hi;
    asm("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=&v"(hi) : "n"((uint32_t)(B & 0xffffffffull)), "n"((uint32_t)(B >> 32)), "v"(dep));
    return dm_from_bits(((uint64_t)hi << 32) | lo);
}
#define DM_KC(bits, dep) dm_kc<bits>(dep)
// DM_TIE(var, dep): no instruction; tells the compiler that var changes whenever dep does, so that values derived from
// var are computed where they are used instead of being hoisted (and kept in registers) wherever dep is not invariant.
#define DM_TIE(var, dep) asm("" : "+v"(var) : "v"(dep))
#else
#define DM_KC(bits, dep) dm_from_bits(bits)
#define DM_TIE(var, dep) ((void)0)
#endif

// DM_ANY(cond): cond on the host; on the device "cond holds in some lane of the wavefront", a scalar branch the compiler
// cannot turn into straight-line selects.  For rare, expensive cases: the body must still test cond per lane.
// DM_COLD(): first statement of such a body; keeps the compiler from executing the body speculatively anyway.
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_ANY(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
#define DM_COLD() asm volatile("; rare path")
#else
#define DM_ANY(cond) (cond)
#define DM_COLD() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// x * 2^n, exact except at the subnormal edge (same operation sequence on every platform).
DM_HD double dm_scalbn(double x, int n)
{
    double y = x;
    if (n > 1023)
    {
        y *= dm_from_bits(0x7FE0000000000000ull);  // 2^1023
        n -= 1023;
        if (n > 1023)
        {
            y *= dm_from_bits(0x7FE0000000000000ull);
            n -= 1023;
            if (n > 1023) n = 1023;
        }
    }
    else if (n < -1022)
    {
        y *= dm_from_bits(0x0360000000000000ull);  // 2^-1022 * 2^53 = 2^-969
        n += 1022 - 53;
        if (n < -1022)
        {
            y *= dm_from_bits(0x0360000000000000ull);
            n += 1022 - 53;
            if (n < -1022) n = -1022;
        }
    }
    return y * dm_from_bits((uint64_t)(0x3ff + n) << 52);
}

// ---------------------------------------------------------------------------------------------
// fmod(x, y) for y > 0.  IEEE fmod is exact, so this agrees bit-for-bit with glibc's fmod
// (slam/util.cpp:47 uses it with y = 2*pi).  Each subtraction below satisfies t <= r < 2t and is
// therefore exact (Sterbenz).
DM_HD double dm_fmod_pos(double x, double y)
{
    double r = dm_fabs(x);
    if (!(r >= y)) return x;  // also returns NaN unchanged
    if (dm_hi(r) >= 0x7ff00000u) return x - x;  // inf -> NaN
    const int ey = (int)((dm_bits(y) >> 52) & 0x7ff);
    while (r >= y)
    {
        const int er = (int)((dm_bits(r) >> 52) & 0x7ff);
        double t = dm_scalbn(y, er - ey);
        if (t > r) t *= 0.5;
        r -= t;
    }
    return (dm_bits(x) >> 63) ? -r : r;
}

// ---------------------------------------------------------------------------------------------
// Argument reduction by pi/2 (Cody-Waite, pi/2 carried to 33+33+53 bits, a third 33-bit piece when
// the argument is within 2^-49 (relative) of a multiple of pi/2).  Returns n (quadrant count) and
// y0 + y1 = |x| - n*pi/2 for the ABSOLUTE value of x; the caller applies the sign.  Written without
// data-dependent branches on the common path (GPU lanes stay converged).
DM_HD int dm_rem_pio2_abs(double t, double* y0, double* y1)
{
    const double invpio2 = dm_from_bits(0x3FE45F306DC9C883ull);  // 6.36619772367581382433e-01
    const double pio2_1 = dm_from_bits(0x3FF921FB54400000ull);   // first 33 bits of pi/2
    const double pio2_2 = dm_from_bits(0x3DD0B4611A600000ull);   // second 33 bits
    const double pio2_2t = dm_from_bits(0x3BA3198A2E037073ull);  // pi/2 - (pio2_1+pio2_2)
    const double pio2_3 = dm_from_bits(0x3BA3198A2E000000ull);   // third 33 bits
    const double pio2_3t = dm_from_bits(0x397B839A252049C1ull);  // pi/2 - (pio2_1+pio2_2+pio2_3)

    if (!(t < 1048576.0))
    {
        // |x| >= 2^20: outside the range the filter produces (normalize_angle keeps headings in
        // (-2pi, pi]).  Deterministic but reduced-accuracy fallback: exact fmod by the double 2*pi.
        if (!(t < dm_from_bits(0x7FF0000000000000ull)))
        {
            *y0 = t - t;  // inf/nan -> nan
            *y1 = 0.0;
            return 0;
        }
        t = dm_fmod_pos(t, dm_from_bits(0x401921FB54442D18ull));
    }
    const int n = (int)(t * invpio2 + 0.5);
    const double fn = (double)n;
    // two exact 33-bit pieces (fn*pio2_1 and fn*pio2_2 are exact products), then the tail
    double r = (t - fn * pio2_1);
    double w = fn * pio2_2;
    double t2 = r;
    r = t2 - w;
    w = fn * pio2_2t - ((t2 - r) - w);
    double a = r - w;
    const int j = (int)((dm_bits(t) >> 52) & 0x7ff);
    const int i = j - (int)((dm_bits(a) >> 52) & 0x7ff);
    if (DM_ANY(i > 49))
    {  // heavy cancellation (the argument within 2^-49 of a multiple of pi/2): third piece, 151 bits of pi/2
        DM_COLD();
        const double t3 = r;
        const double w3 = fn * pio2_3;
        const double r3 = t3 - w3;
        const double w4 = fn * pio2_3t - ((t3 - r3) - w3);
        if (i > 49)
        {
            r = r3;
            w = w4;
            a = r3 - w4;
        }
    }
    *y0 = a;
    *y1 = (r - a) - w;
    return n;
}

// sin on [-pi/4, pi/4] of x + y (y = tail of the reduced argument).
DM_HD double dm_ksin(double x, double y)
{
    const double S1 = DM_KC(0xBFC5555555555549ull, x);  // -1.66666666666666324348e-01
    const double S2 = DM_KC(0x3F8111111110F8A6ull, x);  //  8.33333333332248946124e-03
    const double S3 = DM_KC(0xBF2A01A019C161D5ull, x);  // -1.98412698298579493134e-04
    const double S4 = DM_KC(0x3EC71DE357B1FE7Dull, x);  //  2.75573137070700676789e-06
    const double S5 = DM_KC(0xBE5AE5E68A2B9CEBull, x);  // -2.50507602534068634195e-08
    const double S6 = DM_KC(0x3DE5D93A5ACFD57Cull, x);  //  1.58969099521155010221e-10
    const double z = x * x;
    const double w = z * z;
    const double r = dm_fma(z, dm_fma(z, S4, S3), S2) + (z * w) * dm_fma(z, S6, S5);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// cos on [-pi/4, pi/4] of x + y.
DM_HD double dm_kcos(double x, double y)
{
    const double C1 = DM_KC(0x3FA555555555554Cull, x);  //  4.16666666666666019037e-02
    const double C2 = DM_KC(0xBF56C16C16C15177ull, x);  // -1.38888888888741095749e-03
    const double C3 = DM_KC(0x3EFA01A019CB1590ull, x);  //  2.48015872894767294178e-05
    const double C4 = DM_KC(0xBE927E4F809C52ADull, x);  // -2.75573143513906633035e-07
    const double C5 = DM_KC(0x3E21EE9EBDB4B1C4ull, x);  //  2.08757232129817482790e-09
    const double C6 = DM_KC(0xBDA8FAE9BE8838D4ull, x);  // -1.13596475577881948265e-11
    const double z = x * x;
    const double w = z * z;
    const double r = z * dm_fma(z, dm_fma(z, C3, C2), C1) + (w * w) * dm_fma(z, dm_fma(z, C6, C5), C4);
    const double hz = 0.5 * z;
    const double q = 1.0 - hz;
    return q + (((1.0 - q) - hz) + (z * r - x * y));
}

DM_HD void dm_sincos(double x, double* s, double* c)
{
    double y0, y1;
    const int n = dm_rem_pio2_abs(dm_fabs(x), &y0, &y1);
    const double sn = dm_ksin(y0, y1);
    const double cs = dm_kcos(y0, y1);
    // quadrant of |x|: (sin, cos) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn)
    double ss = (n & 1) ? cs : sn;
    double cc = (n & 1) ? sn : cs;
    if (n & 2) ss = -ss;
    if ((n + 1) & 2) cc = -cc;
    // sin is odd, cos is even
    *s = (dm_bits(x) >> 63) ? -ss : ss;
    *c = cc;
}

DM_HD double dm_sin(double x)
{
    double s, c;
    dm_sincos(x, &s, &c);
    return s;
}

DM_HD double dm_cos(double x)
{
    double s, c;
    dm_sincos(x, &s, &c);
    return c;
}

// ---------------------------------------------------------------------------------------------
// exp and log, round 3: table driven, no division (the schemes of Tang's table-driven exp / log, as used by today's libms: a
// 128-entry table and a short polynomial), because the particle-lane ray kernel is bound by VALU issue and the fdlibm forms
// (a rational function for exp, f/(2+f) for log: an 11-instruction division each) were a sixth of its instructions.
// Tables: detmath_tables.h, generated from their definitions by tools/gen_detmath_tables.py.  Accuracy < 1 ulp
// (tests/test_detmath.py, against 80-bit long double); same bits on host and device as before.
#include "detmath_tables.h"
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const uint64_t dm_exp_tab[256] = {DM_EXP_TABLE};
__device__ __constant__ static const uint64_t dm_log_tab[256] = {DM_LOG_TABLE};
#else
static const uint64_t dm_exp_tab[256] = {DM_EXP_TABLE};
static const uint64_t dm_log_tab[256] = {DM_LOG_TABLE};
#endif

// log by fdlibm's scheme: x = 2^k (1+f), sqrt(2)/2 < 1+f < sqrt(2); log(1+f) through s = f/(2+f).  Kept for arguments near 1,
// where the result is small and the table form would lose relative accuracy.  x: positive, finite, normal.
DM_HD double dm_log_near1(double x)
{
    const double ln2_hi = dm_from_bits(0x3FE62E42FEE00000ull);  // 6.93147180369123816490e-01
    const double ln2_lo = dm_from_bits(0x3DEA39EF35793C76ull);  // 1.90821492927058770002e-10
    const double Lg1 = DM_KC(0x3FE5555555555593ull, x);     // 6.666666666666735130e-01
    const double Lg2 = DM_KC(0x3FD999999997FA04ull, x);     // 3.999999999940941908e-01
    const double Lg3 = DM_KC(0x3FD2492494229359ull, x);     // 2.857142874366239149e-01
    const double Lg4 = DM_KC(0x3FCC71C51D8E78AFull, x);     // 2.222219843214978396e-01
    const double Lg5 = DM_KC(0x3FC7466496CB03DEull, x);     // 1.818357216161805012e-01
    const double Lg6 = DM_KC(0x3FC39A09D078C69Full, x);     // 1.531383769920937332e-01
    const double Lg7 = DM_KC(0x3FC2F112DF3E5244ull, x);     // 1.479819860511658591e-01
    uint64_t u = dm_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    if (hx == 0x3ff00000u && (u << 32) == 0) return 0.0;
    // reduce x into [sqrt(2)/2, sqrt(2))
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    x = dm_from_bits(u);
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * dm_fma(w, dm_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * dm_fma(w, dm_fma(w, dm_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// Natural logarithm.  x = 2^k z with z in [0.6875, 1.375); the 7 bits below the exponent of bits(z) - bits(0.6875) pick
// (invc, logc) with invc near 1/z and logc = -log(invc) to 2^-11 of its last place; r = z invc - 1 (one fma, |r| < 0.0042) and
// log(x) = k ln2 + logc + log1p(r), the last as r - r^2/2 + ... + r^7/7 (truncation below 1e-20), with the leading sum
// k ln2_hi + logc + r kept as a head and an exact tail.
DM_HD double dm_log(double x)
{
    const double ln2_hi = dm_from_bits(0x3FE62E42FEE00000ull);  // 21 trailing zero bits: k * ln2_hi is exact
    const double ln2_lo = dm_from_bits(0x3DEA39EF35793C76ull);
    uint64_t u = dm_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k0 = 0;
    if (hx < 0x00100000u || (hx >> 31))
    {
        if ((u << 1) == 0) return -1.0 / (x * x);  // log(+-0) = -inf
        if (hx >> 31) return (x - x) / 0.0;        // log(-#) = NaN
        k0 = -54;                                  // subnormal: scale up
        x *= dm_from_bits(0x4350000000000000ull);  // 2^54
        u = dm_bits(x);
        hx = (uint32_t)(u >> 32);
    }
    else if (hx >= 0x7ff00000u)
    {
        return x;
    }
    // [0.9375, 1.0625): the result is small
    if (hx - 0x3fee0000u < 0x3ff10000u - 0x3fee0000u && k0 == 0) return dm_log_near1(x);
    const uint64_t tmp = u - 0x3FE6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52) + k0;
    const double z = dm_from_bits(u - (tmp & 0xFFF0000000000000ull));
    const double invc = dm_from_bits(dm_log_tab[2 * i]), logc = dm_from_bits(dm_log_tab[2 * i + 1]);
    const double r = dm_fma(z, invc, -1.0);
    const double kd = (double)k;
    // k ln2_hi + logc + r as a head and its exact rounding errors (two Fast2Sums: |k ln2_hi| > |logc| unless k = 0, and
    // |w| >= |logc| - ... > |r|: arguments whose logc is smaller than r went to dm_log_near1)
    const double P = kd * ln2_hi;  // exact
    const double w = P + logc;
    const double ew = (P - w) + logc;
    const double hi = w + r;
    const double lo = dm_fma(kd, ln2_lo, ((w - hi) + r) + ew);
    // r^2 (A2 + r A3 + r^2 (A4 + r A5 + r^2 (A6 + r A7))), A_n = (-1)^(n+1) / n.  (DM_KC: the compiler otherwise keeps every
    // constant in a vector register pair for the whole kernel, whether written as a literal or not: measured again in round 3,
    // 75 -> 86 registers in k_rays_pl and scratch in the other ray kernels)
    const double A3 = DM_KC(0x3FD5555555555555ull, x);   //  1/3
    const double A4 = DM_KC(0xBFD0000000000000ull, x);   // -1/4
    const double A5 = DM_KC(0x3FC999999999999Aull, x);   //  1/5
    const double A6 = DM_KC(0xBFC5555555555555ull, x);   // -1/6
    const double A7 = DM_KC(0x3FC2492492492492ull, x);   //  1/7
    const double r2 = r * r;
    const double p = dm_fma(r2, dm_fma(r2, dm_fma(r, A7, A6), dm_fma(r, A5, A4)), dm_fma(r, A3, -0.5));
    return dm_fma(r2, p, lo) + hi;
}

// ---------------------------------------------------------------------------------------------
// exp(x): x = (128 m + j) ln2/128 + r, |r| <= ln2/256; exp(x) = 2^m 2^(j/128) exp(r), 2^(j/128) = hi_j (1 + tail_j) from the
// table, exp(r) - 1 = r + r^2/2 + r^3/6 + r^4/24 + r^5/120 (truncation below 6e-19).
DM_HD double dm_exp(double x)
{
    uint32_t hx = dm_hi(x);
    hx &= 0x7fffffffu;
    if (hx >= 0x4086232bu)
    {  // |x| >= 708.39 or NaN
        if (x != x) return x;
        if (x > 709.782712893383973096) return x * dm_from_bits(0x7FE0000000000000ull);  // overflow
        if (x < -745.13321910194110842) return 0.0;                                      // underflow
    }
    if (hx <= 0x3e300000u) return 1.0 + x;  // |x| <= 2^-28
    const double shift = dm_from_bits(0x4338000000000000ull);  // 1.5 * 2^52: (z + shift) - shift = z rounded to an integer
    const double z = x * dm_from_bits(DM_EXP_INV_LN2_N);
    const double kd = (z + shift) - shift;
    const int ki = (int)kd;
    double r = dm_fma(kd, dm_from_bits(DM_EXP_NEG_LN2_HI_N), x);  // exact: 35-bit constant, |kd| < 2^18
    r = dm_fma(kd, dm_from_bits(DM_EXP_NEG_LN2_LO_N), r);
    const int j = ki & 127, m = ki >> 7;
    const double tail = dm_from_bits(dm_exp_tab[2 * j]);
    const uint64_t hbits = dm_exp_tab[2 * j + 1];
    // r + r^2 (1/2 + r / 6) + r^4 (1/24 + r / 120)
    const double C3 = DM_KC(0x3FC5555555555555ull, x);  // 1/6
    const double C4 = DM_KC(0x3FA5555555555555ull, x);  // 1/24
    const double C5 = DM_KC(0x3F81111111111111ull, x);  // 1/120
    const double r2 = r * r;
    const double t = dm_fma(r2 * r2, dm_fma(r, C5, C4), dm_fma(r2, dm_fma(r, C3, 0.5), r));
    const double tmp = tail + t;
    if (m < -1020 || m > 1020)
    {  // the scale 2^m is not a normal number (or close): scale in two steps
        const int adj = m < 0 ? 100 : -8;
        const double s2 = dm_from_bits(hbits + ((uint64_t)(int64_t)(m + adj) << 52));
        const double y = dm_fma(s2, tmp, s2);
        return y * dm_from_bits((uint64_t)(0x3ff - adj) << 52);
    }
    const double scale = dm_from_bits(hbits + ((uint64_t)(int64_t)m << 52));
    return dm_fma(scale, tmp, scale);
}
