// kz_crmath.h - the transcendental functions of the path, defined by their arithmetic.
//
// The reference calls libm (glibc's sinf / cosf / expf / logf / atanf / atan2f / acosf / tanf / powf / hypotf: warp.cpp:41-129, bsdf.cpp:728-734,
// common.cpp:368-400, texture.cpp:66-80, camera.cpp:191-223). Their values are "the exact result, rounded to float" in all but a fraction of a
// percent of the arguments, but not in a way another library reproduces bit for bit - and a path through small triangles turns one last bit
// of a sampled direction into another triangle (scripts/dev/bsdf_bits.py: with ocml's functions on this side and glibc's on the oracle's only
// ~55 % of the sampled directions agreed to the last bit). So each function is DEFINED here as a short sequence of IEEE double operations
// (+ - * / sqrt fma rint, all correctly rounded on gfx950 and on the host alike, no contraction) followed by ONE narrowing to float:
//   * the oracle (oracle/kz_oracle_math.h) states the same sequences independently, and kz_kat_math / tests/test_gpu_parity.py compare the
//     two bit for bit on millions of arguments: same bits by construction, not by tolerance;
//   * the double value is within ~2^-50 of the exact one, so the float IS the correctly rounded result except for about one argument in 2^25
//     (tests/test_oracle_cpu.py checks this against libm's double functions, and counts the - rare - arguments where glibc's float functions,
//     i.e. the reference on this machine, differ: they are the ones glibc does not round correctly).
// Argument ranges are the path's; the circular functions return NaN beyond |x| = 2^20. Denormal results are flushed to zero (the path runs FTZ on both sides, main.cpp:22-23).
#pragma once
#include <stdint.h>

#define KZ_CR_FN __device__ __forceinline__
// A double constant of a polynomial, held in a scalar register pair at its use: v_fma_f64 takes it as its one scalar operand. Left to itself the compiler
// puts every coefficient into a VGPR pair (v_fmac_f64 wants the addend in its destination) and hoists those out of the kernels' loops - 34 VGPRs of
// constants live across kz_wf_shade's survivor loop, which has none to spare. The empty asm only pins where the value lives, not what it is.
KZ_CR_FN double kzcrK(double c) { asm volatile("" : "+s"(c)); return c; }
#define KZ_K(c) kzcrK(c)
// The functions of the rough BSDFs, the textures and the environment map are real calls: rare next to sin / cos, and inlined at every use they cost
// the extended kernels hundreds of spilled registers.
#define KZ_CR_CALL static __device__ __noinline__

KZ_CR_FN double kzcrBits(uint64_t b) { return __builtin_bit_cast(double, b); }
KZ_CR_FN uint64_t kzcrBitsOf(double d) { return __builtin_bit_cast(uint64_t, d); }
// one narrowing, denormal results to (signed) zero
KZ_CR_FN float kzcrNarrow(double d) {
    const float f = (float)d;
    return __builtin_fabsf(f) < 1.17549435e-38f ? __builtin_copysignf(0.0f, f) : f;
}

// sin and cos of x (double), |x| <= 2^20 (the path stays below 2 pi): k = rint(x * 2/pi), r = x - k * pi/2 in two fma steps (pi/2 = 33 high bits + tail, k * high is exact),
// Taylor polynomials to r^17 / r^18 on |r| <= pi/4 (truncation < 1e-19 relative)
KZ_CR_FN void kzcrSinCosD(double x, double &s, double &c) {
    if (!(__builtin_fabs(x) <= 0x1p20)) { s = c = (double)__builtin_nanf(""); return; }      // far outside the path's range (and inf, NaN): NaN on both sides
    const double kd = __builtin_rint(x * 0x1.45f306dc9c883p-1);
    double r = __builtin_fma(-kd, 0x1.921fb54400000p+0, x);
    r = __builtin_fma(-kd, 0x1.0b4611a626331p-34, r);
    const double z = r * r;
    double ps = 0x1.952c77030ad4ap-49;
    ps = __builtin_fma(ps, z, KZ_K(-0x1.ae7f3e733b81fp-41));
    ps = __builtin_fma(ps, z, KZ_K(0x1.6124613a86d09p-33));
    ps = __builtin_fma(ps, z, KZ_K(-0x1.ae64567f544e4p-26));
    ps = __builtin_fma(ps, z, KZ_K(0x1.71de3a556c734p-19));
    ps = __builtin_fma(ps, z, KZ_K(-0x1.a01a01a01a01ap-13));
    ps = __builtin_fma(ps, z, KZ_K(0x1.1111111111111p-7));
    ps = __builtin_fma(ps, z, KZ_K(-0x1.5555555555555p-3));
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = -0x1.6827863b97d97p-53;
    pc = __builtin_fma(pc, z, KZ_K(0x1.ae7f3e733b81fp-45));
    pc = __builtin_fma(pc, z, KZ_K(-0x1.93974a8c07c9dp-37));
    pc = __builtin_fma(pc, z, KZ_K(0x1.1eed8eff8d898p-29));
    pc = __builtin_fma(pc, z, KZ_K(-0x1.27e4fb7789f5cp-22));
    pc = __builtin_fma(pc, z, KZ_K(0x1.a01a01a01a01ap-16));
    pc = __builtin_fma(pc, z, KZ_K(-0x1.6c16c16c16c17p-10));
    pc = __builtin_fma(pc, z, KZ_K(0x1.5555555555555p-5));
    pc = __builtin_fma(pc, z, KZ_K(-0x1.0000000000000p-1));
    const double cr = __builtin_fma(pc, z, 1.0);
    const int q = (int)kd;
    const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;          // quadrant 0: (s, c)  1: (c, -s)  2: (-s, -c)  3: (-c, s)
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
// (inline: as a call it still costs kz_wf_shade 21 spilled VGPRs - the caller's live state around it - where the inlined sequence with pinned coefficients costs none)
KZ_CR_FN void kzSinCos(float x, float *s, float *c) {
    double sd, cd; kzcrSinCosD((double)x, sd, cd);
    *s = (float)sd; *c = (float)cd;
}
KZ_CR_CALL float kzCos(float x) { double sd, cd; kzcrSinCosD((double)x, sd, cd); return (float)cd; }
KZ_CR_CALL float kzTan(float x) { double sd, cd; kzcrSinCosD((double)x, sd, cd); return kzcrNarrow(sd / cd); }

// exp(x), x double in [-745, 709]: k = rint(x * log2 e), r = x - k ln 2 in two fma steps, Taylor to r^13 on |r| <= 0.347 (truncation < 5e-18), times 2^k
KZ_CR_FN double kzcrExpD(double x) {
    const double kd = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(-kd, 0x1.62e42fee00000p-1, x);
    r = __builtin_fma(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = __builtin_fma(p, r, KZ_K(0x1.1eed8eff8d898p-29));
    p = __builtin_fma(p, r, KZ_K(0x1.ae64567f544e4p-26));
    p = __builtin_fma(p, r, KZ_K(0x1.27e4fb7789f5cp-22));
    p = __builtin_fma(p, r, KZ_K(0x1.71de3a556c734p-19));
    p = __builtin_fma(p, r, KZ_K(0x1.a01a01a01a01ap-16));
    p = __builtin_fma(p, r, KZ_K(0x1.a01a01a01a01ap-13));
    p = __builtin_fma(p, r, KZ_K(0x1.6c16c16c16c17p-10));
    p = __builtin_fma(p, r, KZ_K(0x1.1111111111111p-7));
    p = __builtin_fma(p, r, KZ_K(0x1.5555555555555p-5));
    p = __builtin_fma(p, r, KZ_K(0x1.5555555555555p-3));
    p = __builtin_fma(p, r, KZ_K(0x1.0000000000000p-1));
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int k = (int)kd;                                              // |k| <= 1075: two exact power-of-two factors keep each in the normal range
    const int k1 = k / 2, k2 = k - k1;
    return p * kzcrBits((uint64_t)(k1 + 1023) << 52) * kzcrBits((uint64_t)(k2 + 1023) << 52);
}
KZ_CR_CALL float kzExp(float x) {
    if (!(x > -104.0f)) return x != x ? x : 0.0f;                       // below every float (and -inf); NaN stays NaN
    if (x > 89.0f) return __builtin_inff();
    return kzcrNarrow(kzcrExpD((double)x));
}

// log(x), x a positive normal double: x = m 2^e with m in [sqrt(1/2), sqrt 2), f = (m - 1) / (m + 1), log m = 2 f (1 + f^2/3 + ... + f^20/21)
// (|f| <= 0.1716: truncation < 1e-18 relative), + e ln 2 (ln 2 = 32 high bits + tail, e * high is exact)
KZ_CR_FN double kzcrLogD(double x) {
    const uint64_t b = kzcrBitsOf(x);
    int e = (int)((b >> 52) & 0x7ffu) - 1023;
    double m = kzcrBits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }
    const double f = (m - 1.0) / (m + 1.0), z = f * f;
    double p = 0x1.8618618618618p-5;
    p = __builtin_fma(p, z, KZ_K(0x1.af286bca1af28p-5));
    p = __builtin_fma(p, z, KZ_K(0x1.e1e1e1e1e1e1ep-5));
    p = __builtin_fma(p, z, KZ_K(0x1.1111111111111p-4));
    p = __builtin_fma(p, z, KZ_K(0x1.3b13b13b13b14p-4));
    p = __builtin_fma(p, z, KZ_K(0x1.745d1745d1746p-4));
    p = __builtin_fma(p, z, KZ_K(0x1.c71c71c71c71cp-4));
    p = __builtin_fma(p, z, KZ_K(0x1.2492492492492p-3));
    p = __builtin_fma(p, z, KZ_K(0x1.999999999999ap-3));
    p = __builtin_fma(p, z, KZ_K(0x1.5555555555555p-2));
    const double f2 = f + f;
    const double lm = __builtin_fma(f2 * z, p, f2);
    const double ed = (double)e;
    return __builtin_fma(ed, 0x1.62e42fee00000p-1, __builtin_fma(ed, 0x1.a39ef35793c76p-33, lm));
}
KZ_CR_CALL float kzLog(float x) {
    if (x != x || x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    return kzcrNarrow(kzcrLogD((double)x));
}
// pow(x, y) = exp(y log x) for x > 0 (the path's uses: sRGB curves, bases in (0.003, 1e4), |y log x| < 25, relative error < 2^-47)
KZ_CR_CALL float kzPow(float x, float y) {
    if (x != x || y != y || x < 0.0f) return __builtin_nanf("");                           // (negative bases: NaN, as powf gives for the non-integer exponents of the path)
    if (x == 0.0f) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : __builtin_inff());
    if (x == __builtin_inff()) return y > 0.0f ? x : (y == 0.0f ? 1.0f : 0.0f);
    if (x == 1.0f) return 1.0f;
    const double z = (double)y * kzcrLogD((double)x);
    if (!(z > -104.0)) return 0.0f;
    if (z > 89.0) return __builtin_inff();
    return kzcrNarrow(kzcrExpD(z));
}

// atan(x), x >= 0 (or +inf): x > 1 -> pi/2 - atan(1/x); then x in [0, 1] = c + d with c = rint(4 x) / 4: atan x = atan c + atan t,
// t = (x - c) / (1 + x c), |t| <= 1/8, Taylor to t^19 (truncation < 1e-19 relative)
KZ_CR_FN double kzcrAtanPosD(double x) {
    const bool inv = x > 1.0;
    if (inv) x = 1.0 / x;
    const double jd = __builtin_rint(x * 4.0), c = jd * 0.25;
    const double t = (x - c) / __builtin_fma(x, c, 1.0), z = t * t;
    const int j = (int)jd;
    const double base = j == 0 ? 0.0 : j == 1 ? 0x1.f5b75f92c80ddp-3 : j == 2 ? 0x1.dac670561bb4fp-2 : j == 3 ? 0x1.4978fa3269ee1p-1 : 0x1.921fb54442d18p-1;
    double p = -0x1.af286bca1af28p-5;
    p = __builtin_fma(p, z, KZ_K(0x1.e1e1e1e1e1e1ep-5));
    p = __builtin_fma(p, z, KZ_K(-0x1.1111111111111p-4));
    p = __builtin_fma(p, z, KZ_K(0x1.3b13b13b13b14p-4));
    p = __builtin_fma(p, z, KZ_K(-0x1.745d1745d1746p-4));
    p = __builtin_fma(p, z, KZ_K(0x1.c71c71c71c71cp-4));
    p = __builtin_fma(p, z, KZ_K(-0x1.2492492492492p-3));
    p = __builtin_fma(p, z, KZ_K(0x1.999999999999ap-3));
    p = __builtin_fma(p, z, KZ_K(-0x1.5555555555555p-2));
    const double r = base + __builtin_fma(t * z, p, t);
    return inv ? (0x1.921fb54442d18p+0 - r) + 0x1.1a62633145c07p-54 : r;
}
KZ_CR_CALL float kzAtan(float x) {
    if (x != x) return x;
    const float r = kzcrNarrow(kzcrAtanPosD((double)__builtin_fabsf(x)));
    return __builtin_copysignf(r, x);
}
// atan2(y, x) with IEEE's conventions for zeros and infinities
KZ_CR_CALL float kzAtan2(float y, float x) {
    if (x != x || y != y) return __builtin_nanf("");
    const double ay = (double)__builtin_fabsf(y), ax = (double)__builtin_fabsf(x);
    const double inf = (double)__builtin_inff();
    double a;
    if (ay == 0.0) a = 0.0;
    else if (ay == inf) a = ax == inf ? 0x1.921fb54442d18p-1 : 0x1.921fb54442d18p+0;
    else if (ay > ax) a = (0x1.921fb54442d18p+0 - kzcrAtanPosD(ax / ay)) + 0x1.1a62633145c07p-54;
    else a = kzcrAtanPosD(ay / ax);                                     // ax >= ay > 0 (ax = inf: 0)
    if (__builtin_signbit(x)) a = (0x1.921fb54442d18p+1 - a) + 0x1.1a62633145c07p-53;
    const float r = kzcrNarrow(a);
    return __builtin_copysignf(r, y);
}
// acos(x) = 2 atan(sqrt((1 - x) / (1 + x))), |x| <= 1 (NaN outside)
KZ_CR_CALL float kzAcos(float x) {
    if (!(__builtin_fabsf(x) <= 1.0f)) return __builtin_nanf("");
    const double xd = (double)x;
    if (xd == -1.0) return (float)0x1.921fb54442d18p+1;
    return (float)(2.0 * kzcrAtanPosD(__builtin_sqrt((1.0 - xd) / (1.0 + xd))));
}
// hypot(x, y): the squares are exact in double
KZ_CR_FN float kzHypot(float x, float y) {
    const double xd = (double)x, yd = (double)y;
    return kzcrNarrow(__builtin_sqrt(xd * xd + yd * yd));
}
// x^3 (powf(x, 3.f))
KZ_CR_FN float kzCube(float x) { const double xd = (double)x; return kzcrNarrow(xd * xd * xd); }
