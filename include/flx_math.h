/*
 * flx_math.h -- the arithmetic contract of the hot path.
 *
 * Every floating-point operation the wavefront kernels perform is spelled out
 * here in plain IEEE-754 binary32 operations (+ - * / sqrt, compares, integer
 * conversions, bit casts).  Both the HIP kernels (hipcc, -ffp-contract=off,
 * correctly rounded divide/sqrt) and the CPU oracle (g++, -ffp-contract=off)
 * include this header, so that device and oracle results are BIT-IDENTICAL and
 * parity tests can demand exact equality instead of a tolerance.
 *
 * The reference calls the OpenCL built-ins sin/cos/tan/atan2/acos/pow and
 * native_sin/native_cos/native_recip (reference: src/utils.cl:75-112,
 * src/ggx.cl:19-36, src/env_map.cl:14-37, src/intersect.cl:43); their precision
 * is implementation-defined (OpenCL 1.2 s7.4: <= 4-5 ulp; native_*: unspecified),
 * so there is no single "reference bit pattern".  The functions below are
 * classic minimax/Cody-Waite constructions (Cephes-style single precision)
 * accurate to ~1-2 ulp on the ranges the path uses; the oracle-vs-reference
 * tests (tests/test_oracle_vs_ref.py) bound the difference to libm at 1e-5 rel.
 *
 * Vector helpers fix the evaluation order (x*x + y*y + z*z, left to right).
 */
#ifndef FLX_MATH_H
#define FLX_MATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define FLX_HD __host__ __device__ __forceinline__
#else
#define FLX_HD inline
#endif

namespace flx {

#define FLX_PI      3.14159265358979323846f
#define FLX_2PI     6.2831853071795864f
#define FLX_INV_PI  0.3183098861837907f
#define FLX_PIO2    1.5707963267948966192f
#define FLX_PIO4    0.7853981633974483096f
#define FLX_FLT_MAX 3.402823466e+38f

FLX_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FLX_HD float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---------------------------------------------------------------- scalars */

/* IEEE-754 minNum/maxNum (a NaN operand yields the other one) = OpenCL fmin/fmax = one
 * v_min_f32 / v_max_f32 on the device.  gfx9 orders the zeros: max(+0, -0) = +0 and min(+0, -0) = -0 whichever operand comes first;
 * the host spelling does the same (AND / OR of the bit patterns of two EQUAL operands), so that a stored zero carries the same sign
 * on both sides -- no comparison in the path observes it, but the parity tests compare stored values bit for bit
 * (round 4: lastPdfImplicit = max(0, -0) differed in the sign bit for ~1 path in 10^6). */
#if defined(__HIP_DEVICE_COMPILE__)
FLX_HD float fminf_(float a, float b) { return __builtin_fminf(a, b); }
FLX_HD float fmaxf_(float a, float b) { return __builtin_fmaxf(a, b); }
#else
FLX_HD float fminf_(float a, float b) { return a < b ? a : (b < a ? b : (a != a ? b : (b != b ? a : u2f(f2u(a) | f2u(b))))); }
FLX_HD float fmaxf_(float a, float b) { return a > b ? a : (b > a ? b : (a != a ? b : (b != b ? a : u2f(f2u(a) & f2u(b))))); }
#endif
FLX_HD float clampf(float v, float lo, float hi) { return fminf_(fmaxf_(v, lo), hi); }
FLX_HD float absf(float a) { return __builtin_fabsf(a); }

/* sin and cos of x, |x| < 8192, Cody-Waite reduction to [-pi/4, pi/4]. */
FLX_HD void sincosf_(float x, float *s, float *c)
{
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    bool sneg = x < 0.0f;
    float ax = absf(x);
    uint32_t j = (uint32_t)(ax * FOPI);
    float y = (float)j;
    if (j & 1u) { j += 1u; y += 1.0f; }
    j &= 7u;
    bool cneg = false;
    if (j > 3u) { j -= 4u; sneg = !sneg; cneg = !cneg; }
    if (j > 1u) cneg = !cneg;
    float xr = ((ax - y * DP1) - y * DP2) - y * DP3;
    float z = xr * xr;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * xr + xr;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    bool swap = (j == 1u || j == 2u);
    float sv = swap ? pc : ps;
    float cv = swap ? ps : pc;
    *s = sneg ? -sv : sv;
    *c = cneg ? -cv : cv;
}
FLX_HD float sinf_(float x) { float s, c; sincosf_(x, &s, &c); return s; }
FLX_HD float cosf_(float x) { float s, c; sincosf_(x, &s, &c); return c; }
FLX_HD float tanf_(float x) { float s, c; sincosf_(x, &s, &c); return s / c; }

FLX_HD float atanf_(float x)
{
    bool neg = x < 0.0f;
    float a = absf(x), y;
    if (a > 2.414213562373095f)       { y = FLX_PIO2; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = FLX_PIO4; a = (a - 1.0f) / (a + 1.0f); }
    else                              { y = 0.0f; }
    float z = a * a;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a;
    return neg ? -y : y;
}

FLX_HD float atan2f_(float y, float x)
{
    if (x == 0.0f) return y > 0.0f ? FLX_PIO2 : (y < 0.0f ? -FLX_PIO2 : 0.0f);
    float z = atanf_(y / x);
    if (x < 0.0f) z = (y < 0.0f) ? z - FLX_PI : z + FLX_PI;
    return z;
}

FLX_HD float asinf_(float x)
{
    bool neg = x < 0.0f;
    float a = absf(x);
    if (a > 1.0f) return 0.0f;
    if (a < 1.0e-4f) return x;
    float z, w; bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); w = sqrtf(z); }
    else     { w = a; z = a * a; }
    float r = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * w + w;
    if (big) { r = r + r; r = FLX_PIO2 - r; }
    return neg ? -r : r;
}

FLX_HD float acosf_(float x)
{
    if (x < -0.5f) return FLX_PI - 2.0f * asinf_(sqrtf(0.5f * (1.0f + x)));
    if (x >  0.5f) return 2.0f * asinf_(sqrtf(0.5f * (1.0f - x)));
    return FLX_PIO2 - asinf_(x);
}

/* natural log, x > 0 (normal numbers) */
FLX_HD float logf_(float x)
{
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xffu) - 126;              /* x = m * 2^e, m in [0.5,1) */
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else                           { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
               + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    float r = m + y;
    r += 0.693359375f * fe;
    return r;
}

/* e^x for |x| < 87 */
FLX_HD float expf_(float x)
{
    float n = floorf(1.44269504088896341f * x + 0.5f);
    x -= n * 0.693359375f;
    x -= n * -2.12194440e-4f;
    float z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
          + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    int ni = (int)n;
    /* scale by 2^ni, ni in [-126, 127] after clamping */
    if (ni < -126) return 0.0f;
    if (ni > 127) ni = 127;
    return z * u2f((uint32_t)(ni + 127) << 23);
}

/* x^y for x >= 0 (the path only raises colours to 2.2 and 1/2.2) */
FLX_HD float powf_(float x, float y)
{
    if (!(x > 1.17549435e-38f)) return 0.0f;
    float t = y * logf_(x);
    if (t < -87.0f) return 0.0f;
    if (t > 87.0f) t = 87.0f;
    return expf_(t);
}

/* ---------------------------------------------------------------- vectors */

struct f3 { float x, y, z; };
struct f2 { float x, y; };

FLX_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
FLX_HD f3 mk3(float v) { return mk3(v, v, v); }
FLX_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
FLX_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
FLX_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
FLX_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
FLX_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
FLX_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
FLX_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
FLX_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
FLX_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
FLX_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FLX_HD f3 cross(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
FLX_HD float length(f3 a) { return sqrtf(dot(a, a)); }
/* normalize: one correctly rounded reciprocal of the length, then 3 multiplies.  Edge cases as OpenCL 1.2 s7.5.1 defines them for the
 * built-in the reference calls (and as ROCm's OpenCL on the MI355X returns them, profiles/r02_ocl_builtin_gap.json): a vector of zeros is
 * returned unchanged (NOT 0 * inf = NaN -- the stale all-zero shadowDir of a path without NEE reaches bxdfEval this way,
 * src/wf_mat_diffuse.cl:34-37); an infinite element counts as +-1 and every finite one as 0; NaN stays NaN. */
FLX_HD f3 normalize(f3 a)
{
    if (a.x == 0.0f && a.y == 0.0f && a.z == 0.0f) return a;
    const float big = FLX_FLT_MAX;
    if (absf(a.x) > big || absf(a.y) > big || absf(a.z) > big) {
        a.x = absf(a.x) > big ? __builtin_copysignf(1.0f, a.x) : 0.0f * a.x;
        a.y = absf(a.y) > big ? __builtin_copysignf(1.0f, a.y) : 0.0f * a.y;
        a.z = absf(a.z) > big ? __builtin_copysignf(1.0f, a.z) : 0.0f * a.z;
    }
    float inv = 1.0f / sqrtf(dot(a, a));
    return a * inv;
}
FLX_HD bool is_zero(f3 v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }
FLX_HD f3 pow3(f3 v, float e) { return mk3(powf_(v.x, e), powf_(v.y, e), powf_(v.z, e)); }
FLX_HD f3 min3(f3 a, f3 b) { return mk3(fminf_(a.x, b.x), fminf_(a.y, b.y), fminf_(a.z, b.z)); }
FLX_HD f3 max3(f3 a, f3 b) { return mk3(fmaxf_(a.x, b.x), fmaxf_(a.y, b.y), fmaxf_(a.z, b.z)); }

/* barycentric blend (reference: src/utils.cl:25-28) */
FLX_HD f3 bary(float u, float v, f3 a, f3 b, f3 c) { return (1.0f - u - v) * a + u * b + v * c; }

/* integer hash RNG (reference: src/random.cl:7-22) */
FLX_HD uint32_t hash_u32(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
FLX_HD float rand01(uint32_t *seed)
{
    *seed = hash_u32(*seed);
    return (float)(*seed) * (1.0f / 4294967296.0f);
}

} /* namespace flx */

#endif /* FLX_MATH_H */
