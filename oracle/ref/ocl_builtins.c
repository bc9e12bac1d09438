/*
 * ocl_builtins.c -- host definitions of the OpenCL C 1.2 built-in functions that the
 * reference's kernels call, so that those kernels -- compiled UNMODIFIED from
 * /root/reference/src/*.cl for x86-64 by clang's OpenCL front end -- can execute on the CPU.
 *
 * TEST INFRASTRUCTURE (oracle/_ref build, this container only).  Nothing here is reference code
 * and nothing here is shipped: these are the language built-ins of the OpenCL C specification
 * (s6.12.1 work-item, s6.12.2 math, s6.12.4 common, s6.12.5 geometric, s6.12.7 vload/vstore,
 * s6.12.11 atomics, s6.12.14 image read), which an OpenCL CPU runtime would normally supply.
 * Math built-ins forward to the C library's float functions (sinf, cosf, ...); native_* forward
 * to the same (the reference uses native_* only as a speed hint).  No FMA contraction.
 * Work-items run in ascending global id on one thread (the parity tests: deterministic queue order) or,
 * with ref_set_threads(n > 1), spread over n host threads like a CPU OpenCL device (bench.py's cpu_baseline):
 * the current id is thread-local and the atomics are real; barrier() is a no-op (no kernel on the path needs one).  Symbols carry the Itanium-mangled names clang emits for the
 * overloadable built-ins (checked with `nm`).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));

/* ---- work-item functions (s6.12.1) */
__thread size_t ref_current_gid = 0;
size_t b_get_global_id(unsigned d) __asm__("_Z13get_global_idj");
size_t b_get_global_id(unsigned d) { return d == 0 ? ref_current_gid : 0; }
size_t b_get_local_id(unsigned d) __asm__("_Z12get_local_idj");
size_t b_get_local_id(unsigned d) { return d == 0 ? (ref_current_gid & 63) : 0; }
void b_barrier(unsigned f) __asm__("_Z7barrierj");
void b_barrier(unsigned f) { (void)f; }

/* ---- atomics (s6.12.11) */
unsigned b_atomic_inc_g(volatile unsigned *p) __asm__("_Z10atomic_incPU8CLglobalVj");
unsigned b_atomic_inc_g(volatile unsigned *p) { return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
unsigned b_atomic_inc_l(volatile unsigned *p) __asm__("_Z10atomic_incPU7CLlocalVj");
unsigned b_atomic_inc_l(volatile unsigned *p) { return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
unsigned b_atomic_add_g(volatile unsigned *p, unsigned v) __asm__("_Z10atomic_addPU8CLglobalVjj");
unsigned b_atomic_add_g(volatile unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
float b_atomic_xchg_f(volatile float *p, float v) __asm__("_Z11atomic_xchgPU8CLglobalVff");
float b_atomic_xchg_f(volatile float *p, float v)
{
    unsigned in, out; __builtin_memcpy(&in, &v, 4);
    out = __atomic_exchange_n((volatile unsigned *)p, in, __ATOMIC_RELAXED);
    float o; __builtin_memcpy(&o, &out, 4); return o;
}
unsigned b_atomic_cmpxchg(volatile unsigned *p, unsigned c, unsigned v) __asm__("_Z14atomic_cmpxchgPU8CLglobalVjjj");
unsigned b_atomic_cmpxchg(volatile unsigned *p, unsigned c, unsigned v)
{
    unsigned expected = c;
    __atomic_compare_exchange_n(p, &expected, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected;                                   /* the old value, whether or not the swap happened */
}

/* ---- math (s6.12.2) */
float b_sin(float x) __asm__("_Z3sinf");   float b_sin(float x) { return sinf(x); }
float b_cos(float x) __asm__("_Z3cosf");   float b_cos(float x) { return cosf(x); }
float b_tan(float x) __asm__("_Z3tanf");   float b_tan(float x) { return tanf(x); }
float b_acos(float x) __asm__("_Z4acosf"); float b_acos(float x) { return acosf(x); }
float b_atan2(float y, float x) __asm__("_Z5atan2ff"); float b_atan2(float y, float x) { return atan2f(y, x); }
float b_sqrt(float x) __asm__("_Z4sqrtf"); float b_sqrt(float x) { return sqrtf(x); }
float b_fabs(float x) __asm__("_Z4fabsf"); float b_fabs(float x) { return fabsf(x); }
float b_floor(float x) __asm__("_Z5floorf"); float b_floor(float x) { return floorf(x); }
float b_fmin(float a, float b) __asm__("_Z4fminff"); float b_fmin(float a, float b) { return fminf(a, b); }
float b_fmax(float a, float b) __asm__("_Z4fmaxff"); float b_fmax(float a, float b) { return fmaxf(a, b); }
float b_native_sin(float x) __asm__("_Z10native_sinf"); float b_native_sin(float x) { return sinf(x); }
float b_native_cos(float x) __asm__("_Z10native_cosf"); float b_native_cos(float x) { return cosf(x); }
/* powr(x, y): x >= 0 by definition; OpenCL 1.2 s7.5.1: powr(x < 0, y), powr(+-0, +-0), powr(+inf, +-0), powr(+1, +-inf) and any NaN
 * operand give NaN (pow() gives 1 / a signed value there).  ROCm's device returns exactly these NaNs (profiles/r02_ocl_builtin_gap.json).
 * Dead code in the reference: only schlickDielectric (src/fresnel.cl:33) calls it and nothing calls that. */
float b_native_powr(float x, float y) __asm__("_Z11native_powrff");
float b_native_powr(float x, float y)
{
    if (x != x || y != y || x < 0.0f) return NAN;
    if ((x == 0.0f || isinf(x)) && y == 0.0f) return NAN;
    if (x == 1.0f && isinf(y)) return NAN;
    return powf(x, y);
}
float3 b_native_recip3(float3 v) __asm__("_Z12native_recipDv3_f");
float3 b_native_recip3(float3 v) { float3 r; r.x = 1.0f / v.x; r.y = 1.0f / v.y; r.z = 1.0f / v.z; return r; }
float3 b_sqrt3(float3 v) __asm__("_Z4sqrtDv3_f");
float3 b_sqrt3(float3 v) { float3 r; r.x = sqrtf(v.x); r.y = sqrtf(v.y); r.z = sqrtf(v.z); return r; }
float3 b_pow3(float3 a, float3 b) __asm__("_Z3powDv3_fS_");
float3 b_pow3(float3 a, float3 b) { float3 r; r.x = powf(a.x, b.x); r.y = powf(a.y, b.y); r.z = powf(a.z, b.z); return r; }
float3 b_fmin3(float3 a, float3 b) __asm__("_Z4fminDv3_fS_");
float3 b_fmin3(float3 a, float3 b) { float3 r; r.x = fminf(a.x, b.x); r.y = fminf(a.y, b.y); r.z = fminf(a.z, b.z); return r; }
float3 b_fmax3(float3 a, float3 b) __asm__("_Z4fmaxDv3_fS_");
float3 b_fmax3(float3 a, float3 b) { float3 r; r.x = fmaxf(a.x, b.x); r.y = fmaxf(a.y, b.y); r.z = fmaxf(a.z, b.z); return r; }

/* ---- integer / common (s6.12.3, s6.12.4) */
/* max(float, float): "y if x < y, otherwise x", results undefined for NaN operands (s6.12.4).  ROCm's OpenCL compiles it to
 * v_max_f32 = fmax (a NaN operand yields the other one; measured on the MI355X, profiles/r02_ocl_builtin_gap.json), and that is
 * what the call sites with a possibly-NaN FIRST operand -- max(dot(areaLight.N, -L), 0.0f), src/wf_logic.cl:275 -- get on a real
 * device; the literal "x < y ? y : x" would pass that NaN through. */
float b_maxf(float a, float b) __asm__("_Z3maxff"); float b_maxf(float a, float b) { return fmaxf(a, b); }
unsigned b_maxu(unsigned a, unsigned b) __asm__("_Z3maxjj"); unsigned b_maxu(unsigned a, unsigned b) { return a < b ? b : a; }
unsigned b_minu(unsigned a, unsigned b) __asm__("_Z3minjj"); unsigned b_minu(unsigned a, unsigned b) { return b < a ? b : a; }
int b_mini(int a, int b) __asm__("_Z3minii"); int b_mini(int a, int b) { return b < a ? b : a; }
float b_clampf(float x, float lo, float hi) __asm__("_Z5clampfff");
float b_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
int2 b_clampi2(int2 x, int2 lo, int2 hi) __asm__("_Z5clampDv2_iS_S_");
int2 b_clampi2(int2 x, int2 lo, int2 hi)
{
    int2 r;
    r.x = x.x < lo.x ? lo.x : (x.x > hi.x ? hi.x : x.x);
    r.y = x.y < lo.y ? lo.y : (x.y > hi.y ? hi.y : x.y);
    return r;
}

/* ---- geometric (s6.12.5) */
float b_dot3(float3 a, float3 b) __asm__("_Z3dotDv3_fS_");
float b_dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float b_dot4(float4 a, float4 b) __asm__("_Z3dotDv4_fS_");
float b_dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
float3 b_cross(float3 a, float3 b) __asm__("_Z5crossDv3_fS_");
float3 b_cross(float3 a, float3 b)
{
    float3 r; r.x = a.y * b.z - a.z * b.y; r.y = a.z * b.x - a.x * b.z; r.z = a.x * b.y - a.y * b.x; return r;
}
float b_length3(float3 a) __asm__("_Z6lengthDv3_f");
float b_length3(float3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
float3 b_normalize3(float3 a) __asm__("_Z9normalizeDv3_f");
/* OpenCL 1.2 s7.5.1 edge cases: normalize(v) returns v if all elements are zero; an infinite element is replaced by copysign(1, .) and
 * every finite one by 0 * itself before proceeding; a NaN element makes the result NaN.  (Measured on the MI355X: exactly that.  The
 * device additionally rescales so that |v| < 1e-19 or > 1e19 neither underflows nor overflows; such lengths do not occur on the path.) */
float3 b_normalize3(float3 a)
{
    if (a.x == 0.0f && a.y == 0.0f && a.z == 0.0f) return a;
    if (isinf(a.x) || isinf(a.y) || isinf(a.z)) {
        a.x = isinf(a.x) ? copysignf(1.0f, a.x) : 0.0f * a.x;
        a.y = isinf(a.y) ? copysignf(1.0f, a.y) : 0.0f * a.y;
        a.z = isinf(a.z) ? copysignf(1.0f, a.z) : 0.0f * a.z;
    }
    float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); float3 r; r.x = a.x / l; r.y = a.y / l; r.z = a.z / l; return r;
}

/* ---- vload / vstore (s6.12.7) */
float4 b_vload4(size_t off, const float *p) __asm__("_Z6vload4mPU8CLglobalKf");
float4 b_vload4(size_t off, const float *p) { float4 r; memcpy(&r, p + off * 4, 16); return r; }
void b_vstore4(float4 v, size_t off, float *p) __asm__("_Z7vstore4Dv4_fmPU8CLglobalf");
void b_vstore4(float4 v, size_t off, float *p) { memcpy(p + off * 4, &v, 16); }

/* ---- images (s6.12.14, filtering per s8.2) */
typedef struct { int width, height; const float *rgba; } ref_image;   /* what an image2d_t points at here */
void *__translate_sampler_initializer(int v) { return (void *)(intptr_t)v; }
int2 b_get_image_dim(const ref_image *img) __asm__("_Z13get_image_dim14ocl_image2d_ro");
int2 b_get_image_dim(const ref_image *img) { int2 r; r.x = img->width; r.y = img->height; return r; }
static float4 texel(const ref_image *img, int i, int j)
{
    if (i < 0) i = 0; if (i > img->width - 1) i = img->width - 1;
    if (j < 0) j = 0; if (j > img->height - 1) j = img->height - 1;
    float4 r; memcpy(&r, img->rgba + ((size_t)j * img->width + i) * 4, 16); return r;
}
float4 b_read_imagef_f(const ref_image *img, void *sampler, float2 c) __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
float4 b_read_imagef_f(const ref_image *img, void *sampler, float2 c)
{
    int s = (int)(intptr_t)sampler;
    float u = c.x, v = c.y;
    if (s & 1) { u *= (float)img->width; v *= (float)img->height; }          /* CLK_NORMALIZED_COORDS_TRUE */
    if ((s & 0x30) == 0x20) {                                                  /* CLK_FILTER_LINEAR */
        float fu = u - 0.5f, fv = v - 0.5f;
        float i0f = floorf(fu), j0f = floorf(fv);
        float a = fu - i0f, b = fv - j0f;
        int i0 = (int)i0f, j0 = (int)j0f;
        float4 t00 = texel(img, i0, j0), t10 = texel(img, i0 + 1, j0), t01 = texel(img, i0, j0 + 1), t11 = texel(img, i0 + 1, j0 + 1);
        return (1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10 + (1.0f - a) * b * t01 + a * b * t11;
    }
    return texel(img, (int)floorf(u), (int)floorf(v));
}
float4 b_read_imagef_i(const ref_image *img, void *sampler, int2 c) __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4 b_read_imagef_i(const ref_image *img, void *sampler, int2 c) { (void)sampler; return texel(img, c.x, c.y); }
