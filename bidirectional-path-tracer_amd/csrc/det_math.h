// Deterministic sin/cos/x^5/log for the device path.
//
// The reference calls libm's cos/sin on phi = 2*pi*r1 (src/core/vec3.h:153-155) and
// pow(1 - cosine, 5) (src/materials/material.h:129).  Device math libraries and glibc round
// those differently in the last bit, and one flipped bit in a scattered direction is enough
// to flip the reference's endpoint-tie visibility test (src/camera.h:425-438).  These versions
// use only IEEE + - * and fma in double-double arithmetic (~2^-100 relative error), so the
// value returned is the correctly rounded one in all but astronomically rare cases and is the
// same on every machine.  Must be compiled with -ffp-contract=off.
#ifndef BDPT_DET_MATH_H
#define BDPT_DET_MATH_H

#include <hip/hip_runtime.h>

namespace bdpt {

struct dd { double hi, lo; };

__device__ __forceinline__ dd two_sum(double a, double b) {
    dd r;
    r.hi = a + b;
    double bb = r.hi - a;
    r.lo = (a - (r.hi - bb)) + (b - bb);
    return r;
}
__device__ __forceinline__ dd fast_two_sum(double a, double b) {
    dd r;
    r.hi = a + b;
    r.lo = b - (r.hi - a);
    return r;
}
__device__ __forceinline__ dd two_prod(double a, double b) {
    dd r;
    r.hi = a * b;
    r.lo = __builtin_fma(a, b, -r.hi);
    return r;
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    dd t = two_sum(a.lo, b.lo);
    s.lo = s.lo + t.hi;
    s = fast_two_sum(s.hi, s.lo);
    s.lo = s.lo + t.lo;
    return fast_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo = p.lo + (a.hi * b.lo + a.lo * b.hi);
    return fast_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo = p.lo + a.lo * b;
    return fast_two_sum(p.hi, p.lo);
}

// (-1)^n/(2n+1)! and (-1)^n/(2n)! as (hi, lo) pairs
__device__ static const double kSinC[15][2] = {
    { 0x1.0000000000000p+0,   0x0.0p+0 },
    {-0x1.5555555555555p-3,  -0x1.5555555555555p-57 },
    { 0x1.1111111111111p-7,   0x1.1111111111111p-63 },
    {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73 },
    { 0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73 },
    {-0x1.ae64567f544e4p-26,  0x1.c062e06d1f209p-80 },
    { 0x1.6124613a86d09p-33,  0x1.f28e0cc748ebep-87 },
    {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97 },
    { 0x1.952c77030ad4ap-49,  0x1.ac981465ddc6cp-103 },
    {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112 },
    { 0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120 },
    {-0x1.761b41316381ap-75,  0x1.3423c7d91404fp-130 },
    { 0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139 },
    {-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149 },
    { 0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157 },
};
__device__ static const double kCosC[15][2] = {
    { 0x1.0000000000000p+0,   0x0.0p+0 },
    {-0x1.0000000000000p-1,   0x0.0p+0 },
    { 0x1.5555555555555p-5,   0x1.5555555555555p-59 },
    {-0x1.6c16c16c16c17p-10,  0x1.f49f49f49f49fp-65 },
    { 0x1.a01a01a01a01ap-16,  0x1.a01a01a01a01ap-76 },
    {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76 },
    { 0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83 },
    {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92 },
    { 0x1.ae7f3e733b81fp-45,  0x1.1d8656b0ee8cbp-101 },
    {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107 },
    { 0x1.e542ba4020225p-62,  0x1.ea72b4afe3c2fp-120 },
    {-0x1.0ce396db7f853p-70,  0x1.aebcdbd20331cp-124 },
    { 0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135 },
    {-0x1.88e85fc6a4e5ap-89,  0x1.71c37ebd16540p-143 },
    { 0x1.0a18a2635085dp-98,  0x1.b9e2e28e1aa54p-153 },
};

// sin and cos of x, |x| <= 2^30 (pi/2 is carried to 2^-164: the reduced argument keeps ~100 good bits for every
// such x).  Callers: random_cosine_direction (0 <= x < 2 pi), noise textures and rotation angles (any sign).
__device__ __forceinline__ void det_sincos(double x, double& s_out, double& c_out) {
    double kd = __builtin_rint(x * 0x1.45f306dc9c883p-1);     // nearest integer to x * 2/pi
    int k = (int)kd;
    const double P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54, P3 = -0x1.f1976b7ed8fbcp-110;
    double r0 = __builtin_fma(-kd, P1, x);                    // exact
    dd p2 = two_prod(kd, P2);
    dd r = two_sum(r0, -p2.hi);
    r.lo = r.lo - p2.lo;
    r.lo = r.lo - kd * P3;
    r = fast_two_sum(r.hi, r.lo);

    dd z = dd_mul(r, r);

    double qs = kSinC[14][0];
#pragma unroll
    for (int n = 13; n >= 8; n--) qs = kSinC[n][0] + z.hi * qs;
    double qc = kCosC[14][0];
#pragma unroll
    for (int n = 13; n >= 9; n--) qc = kCosC[n][0] + z.hi * qc;

    dd cn, as, ac;
    cn.hi = kSinC[7][0]; cn.lo = kSinC[7][1];
    as = dd_add(cn, dd_mul_d(z, qs));
#pragma unroll
    for (int n = 6; n >= 0; n--) {
        cn.hi = kSinC[n][0]; cn.lo = kSinC[n][1];
        as = dd_add(cn, dd_mul(as, z));
    }
    as = dd_mul(as, r);

    cn.hi = kCosC[8][0]; cn.lo = kCosC[8][1];
    ac = dd_add(cn, dd_mul_d(z, qc));
#pragma unroll
    for (int n = 7; n >= 0; n--) {
        cn.hi = kCosC[n][0]; cn.lo = kCosC[n][1];
        ac = dd_add(cn, dd_mul(ac, z));
    }

    double sr = as.hi, cr = ac.hi;
    switch (k & 3) {
        case 0: s_out = sr;  c_out = cr;  break;
        case 1: s_out = cr;  c_out = -sr; break;
        case 2: s_out = -sr; c_out = -cr; break;
        default: s_out = -cr; c_out = sr; break;
    }
}

// x^5 (stands in for std::pow(x, 5))
__device__ __forceinline__ double det_pow5(double x) {
    dd x2 = two_prod(x, x);
    dd x4 = dd_mul(x2, x2);
    dd x5 = dd_mul_d(x4, x);
    return x5.hi;
}

// 1/(2k+1), k = 0..21, as (hi, lo) pairs
__device__ static const double kOddRec[22][2] = {
    { 0x1.0000000000000p+0, 0x0.0p+0 },
    { 0x1.5555555555555p-2, 0x1.5555555555555p-56 },
    { 0x1.999999999999ap-3, -0x1.999999999999ap-57 },
    { 0x1.2492492492492p-3, 0x1.2492492492492p-57 },
    { 0x1.c71c71c71c71cp-4, 0x1.c71c71c71c71cp-58 },
    { 0x1.745d1745d1746p-4, -0x1.745d1745d1746p-59 },
    { 0x1.3b13b13b13b14p-4, -0x1.3b13b13b13b14p-58 },
    { 0x1.1111111111111p-4, 0x1.1111111111111p-60 },
    { 0x1.e1e1e1e1e1e1ep-5, 0x1.e1e1e1e1e1e1ep-61 },
    { 0x1.af286bca1af28p-5, 0x1.af286bca1af28p-59 },
    { 0x1.8618618618618p-5, 0x1.8618618618618p-59 },
    { 0x1.642c8590b2164p-5, 0x1.642c8590b2164p-60 },
    { 0x1.47ae147ae147bp-5, -0x1.eb851eb851eb8p-61 },
    { 0x1.2f684bda12f68p-5, 0x1.2f684bda12f68p-59 },
    { 0x1.1a7b9611a7b96p-5, 0x1.1a7b9611a7b96p-61 },
    { 0x1.0842108421084p-5, 0x1.0842108421084p-60 },
    { 0x1.f07c1f07c1f08p-6, -0x1.f07c1f07c1f08p-61 },
    { 0x1.d41d41d41d41dp-6, 0x1.0750750750750p-60 },
    { 0x1.bacf914c1bad0p-6, -0x1.bacf914c1bad0p-60 },
    { 0x1.a41a41a41a41ap-6, 0x1.0690690690690p-60 },
    { 0x1.8f9c18f9c18fap-6, -0x1.f3831f3831f38p-61 },
    { 0x1.7d05f417d05f4p-6, 0x1.7d05f417d05f4p-62 },
};

__device__ __forceinline__ dd dd_div(dd a, dd b) {
    double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_mul_d(b, -q1));
    double q2 = r.hi / b.hi;
    r = dd_add(r, dd_mul_d(b, -q2));
    double q3 = r.hi / b.hi;
    dd q = fast_two_sum(q1, q2);
    dd t; t.hi = q3; t.lo = 0.0;
    return dd_add(q, t);
}

// Natural logarithm of a positive finite x (stands in for the std::log of
// src/materials/volumes/constant_medium.h:40).  x = m 2^e, sqrt(1/2) <= m < sqrt(2); log m = 2 atanh(s) with
// s = (m - 1)/(m + 1), 22 terms in double-double; e ln 2 from a three-part ln 2 whose first part has 42 bits.
// The steps are kept out of line: the function is called once per medium crossing, not per ray step.
__device__ __noinline__ double det_log(double x) {
    if (x != x || x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_huge_val();
    if (x == __builtin_huge_val()) return x;
    int e = 0;
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    if (((bits >> 52) & 0x7ffull) == 0) {          // subnormal: scale into the normal range first
        x = x * 0x1p54;
        bits = (unsigned long long)__double_as_longlong(x);
        e = -54;
    }
    e += (int)((bits >> 52) & 0x7ffull) - 1022;
    double m = __longlong_as_double((long long)((bits & 0x800fffffffffffffull) | 0x3fe0000000000000ull));    // 0.5 <= m < 1
    if (m < 0x1.6a09e667f3bcdp-1) { m = m * 2.0; e = e - 1; }
    const double f = m - 1.0;                      // exact
    dd num; num.hi = f; num.lo = 0.0;
    dd den = two_sum(2.0, f);
    dd s = dd_div(num, den);
    dd z = dd_mul(s, s);
    dd acc; acc.hi = kOddRec[21][0]; acc.lo = kOddRec[21][1];
    for (int k = 20; k >= 0; k--) {
        dd c; c.hi = kOddRec[k][0]; c.lo = kOddRec[k][1];
        acc = dd_add(c, dd_mul(acc, z));
    }
    dd lm = dd_mul(acc, s);
    lm.hi = lm.hi * 2.0; lm.lo = lm.lo * 2.0;
    const double L1 = 0x1.62e42fefa3800p-1, L2 = 0x1.ef35793c76730p-45, L3 = 0x1.f97b57a079a19p-103;
    const double ed = (double)e;
    dd t1; t1.hi = ed * L1; t1.lo = 0.0;           // exact
    dd t2 = two_prod(ed, L2);
    t2.lo = t2.lo + ed * L3;
    dd r = dd_add(dd_add(t1, t2), lm);
    return r.hi;
}

}  // namespace bdpt
#endif
