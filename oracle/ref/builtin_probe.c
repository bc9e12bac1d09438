/*
 * builtin_probe.c -- array-in / array-out adapters around the OpenCL C built-ins of ocl_builtins.c, so that a test can call
 * exactly the symbols the reference's kernel objects link against (the Itanium-mangled names below) with plain pointers.
 *
 * TEST INFRASTRUCTURE (oracle/_ref build).  tests/test_gpu_ocl_builtins.py runs builder-written probe kernels that call the
 * same built-ins through a REAL OpenCL runtime (ROCm's, on the MI355X of the GPU box) and compares the results with these:
 * that pins the stand-in library against an implementation of the OpenCL C specification it stands in for.
 * No reference code here.  Compiled by the same clang as ocl_builtins.c (float3 = ext_vector_type(3) ABI).
 */
#include <stddef.h>
#include <stdint.h>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));

float o_sin(float) __asm__("_Z3sinf");
float o_cos(float) __asm__("_Z3cosf");
float o_tan(float) __asm__("_Z3tanf");
float o_acos(float) __asm__("_Z4acosf");
float o_sqrt(float) __asm__("_Z4sqrtf");
float o_fabs(float) __asm__("_Z4fabsf");
float o_floor(float) __asm__("_Z5floorf");
float o_native_sin(float) __asm__("_Z10native_sinf");
float o_native_cos(float) __asm__("_Z10native_cosf");
float o_atan2(float, float) __asm__("_Z5atan2ff");
float o_fmin(float, float) __asm__("_Z4fminff");
float o_fmax(float, float) __asm__("_Z4fmaxff");
float o_maxf(float, float) __asm__("_Z3maxff");
float o_native_powr(float, float) __asm__("_Z11native_powrff");
float o_clampf(float, float, float) __asm__("_Z5clampfff");
float3 o_native_recip3(float3) __asm__("_Z12native_recipDv3_f");
float3 o_sqrt3(float3) __asm__("_Z4sqrtDv3_f");
float3 o_pow3(float3, float3) __asm__("_Z3powDv3_fS_");
float3 o_fmin3(float3, float3) __asm__("_Z4fminDv3_fS_");
float3 o_fmax3(float3, float3) __asm__("_Z4fmaxDv3_fS_");
float o_dot3(float3, float3) __asm__("_Z3dotDv3_fS_");
float o_dot4(float4, float4) __asm__("_Z3dotDv4_fS_");
float3 o_cross(float3, float3) __asm__("_Z5crossDv3_fS_");
float o_length3(float3) __asm__("_Z6lengthDv3_f");
float3 o_normalize3(float3) __asm__("_Z9normalizeDv3_f");
unsigned o_maxu(unsigned, unsigned) __asm__("_Z3maxjj");
unsigned o_minu(unsigned, unsigned) __asm__("_Z3minjj");
int o_mini(int, int) __asm__("_Z3minii");
int2 o_clampi2(int2, int2, int2) __asm__("_Z5clampDv2_iS_S_");
float4 o_vload4(size_t, const float *) __asm__("_Z6vload4mPU8CLglobalKf");
void o_vstore4(float4, size_t, float *) __asm__("_Z7vstore4Dv4_fmPU8CLglobalf");
unsigned o_atomic_inc(volatile unsigned *) __asm__("_Z10atomic_incPU8CLglobalVj");
unsigned o_atomic_add(volatile unsigned *, unsigned) __asm__("_Z10atomic_addPU8CLglobalVjj");
float o_atomic_xchg_f(volatile float *, float) __asm__("_Z11atomic_xchgPU8CLglobalVff");
unsigned o_atomic_cmpxchg(volatile unsigned *, unsigned, unsigned) __asm__("_Z14atomic_cmpxchgPU8CLglobalVjjj");
typedef struct { int width, height; const float *rgba; } probe_image;
float4 o_read_imagef_f(const probe_image *, void *, float2) __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
int2 o_get_image_dim(const probe_image *) __asm__("_Z13get_image_dim14ocl_image2d_ro");
void *__translate_sampler_initializer(int);

/* function ids are shared with the probe kernels in tests/test_gpu_ocl_builtins.py */
int probe_f1(int fn, const float *x, float *o, int n)
{
    for (int i = 0; i < n; i++) {
        float v = x[i], r;
        switch (fn) {
        case 0: r = o_sin(v); break;        case 1: r = o_cos(v); break;        case 2: r = o_tan(v); break;
        case 3: r = o_acos(v); break;       case 4: r = o_sqrt(v); break;       case 5: r = o_fabs(v); break;
        case 6: r = o_floor(v); break;      case 7: r = o_native_sin(v); break; case 8: r = o_native_cos(v); break;
        default: return 1;
        }
        o[i] = r;
    }
    return 0;
}

int probe_f2(int fn, const float *x, const float *y, float *o, int n)
{
    for (int i = 0; i < n; i++) {
        float r;
        switch (fn) {
        case 0: r = o_atan2(x[i], y[i]); break;  case 1: r = o_fmin(x[i], y[i]); break;  case 2: r = o_fmax(x[i], y[i]); break;
        case 3: r = o_maxf(x[i], y[i]); break;   case 4: r = o_native_powr(x[i], y[i]); break;
        default: return 1;
        }
        o[i] = r;
    }
    return 0;
}

int probe_f3(int fn, const float *x, const float *y, const float *z, float *o, int n)
{
    if (fn != 0) return 1;
    for (int i = 0; i < n; i++) o[i] = o_clampf(x[i], y[i], z[i]);
    return 0;
}

/* a, b, o: n x 4 floats (xyz + pad / w) */
int probe_v3(int fn, const float *a, const float *b, float *o, int n)
{
    for (int i = 0; i < n; i++) {
        float3 A, B, R = {0.0f, 0.0f, 0.0f};
        float s = 0.0f;
        A.x = a[4 * i]; A.y = a[4 * i + 1]; A.z = a[4 * i + 2];
        B.x = b[4 * i]; B.y = b[4 * i + 1]; B.z = b[4 * i + 2];
        switch (fn) {
        case 0: R = o_normalize3(A); break;   case 1: R = o_native_recip3(A); break;  case 2: R = o_sqrt3(A); break;
        case 3: R = o_cross(A, B); break;     case 4: R = o_pow3(A, B); break;        case 5: R = o_fmin3(A, B); break;
        case 6: R = o_fmax3(A, B); break;
        case 7: s = o_dot3(A, B); R.x = s; break;
        case 8: s = o_length3(A); R.x = s; break;
        case 9: { float4 A4, B4; A4.x = A.x; A4.y = A.y; A4.z = A.z; A4.w = a[4 * i + 3]; B4.x = B.x; B4.y = B.y; B4.z = B.z; B4.w = b[4 * i + 3];
                  R.x = o_dot4(A4, B4); break; }
        default: return 1;
        }
        o[4 * i] = R.x; o[4 * i + 1] = R.y; o[4 * i + 2] = R.z; o[4 * i + 3] = 0.0f;
    }
    return 0;
}

int probe_int(int fn, const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *o, int n)
{
    for (int i = 0; i < n; i++) {
        switch (fn) {
        case 0: o[i] = o_maxu(a[i], b[i]); break;
        case 1: o[i] = o_minu(a[i], b[i]); break;
        case 2: o[i] = (uint32_t)o_mini((int)a[i], (int)b[i]); break;
        case 3: { int2 x, lo, hi, r; x.x = (int)a[i]; x.y = (int)~a[i]; lo.x = (int)b[i]; lo.y = (int)b[i] - 7; hi.x = (int)c[i]; hi.y = (int)c[i] + 9;
                  r = o_clampi2(x, lo, hi); o[i] = (uint32_t)r.x ^ ((uint32_t)r.y * 2654435761u); break; }
        default: return 1;
        }
    }
    return 0;
}

/* o[i*4..] = vload4(i, in) stored back with vstore4 at slot perm[i] */
int probe_vls(const float *in, const uint32_t *perm, float *o, int n)
{
    for (int i = 0; i < n; i++) o_vstore4(o_vload4((size_t)i, in), (size_t)perm[i], o);
    return 0;
}

/* one work-item's view of the atomics: returned old values + final memory */
int probe_atomics(uint32_t *mem /*4*/, float *fmem /*2*/, uint32_t *old /*8*/)
{
    old[0] = o_atomic_inc(&mem[0]);
    old[1] = o_atomic_inc(&mem[0]);
    old[2] = o_atomic_add(&mem[1], 5u);
    old[3] = o_atomic_add(&mem[1], 0xFFFFFFFFu);
    { float f = o_atomic_xchg_f(&fmem[0], 2.5f); __builtin_memcpy(&old[4], &f, 4); }
    { float f = o_atomic_xchg_f(&fmem[0], -0.0f); __builtin_memcpy(&old[5], &f, 4); }
    old[6] = o_atomic_cmpxchg(&mem[2], 7u, 9u);       /* mem[2] == 7: swaps */
    old[7] = o_atomic_cmpxchg(&mem[2], 7u, 11u);      /* mem[2] == 9 now: does not */
    return 0;
}

/* read_imagef(image, sampler, (u, v)) for n coordinates; sampler = the reference's
 * CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_CLAMP_TO_EDGE | CLK_FILTER_LINEAR (src/env_map.cl:10) as clang encodes it */
int probe_read_imagef(const float *rgba, int w, int h, int sampler_bits, const float *uv, float *o, int n, int *dim)
{
    probe_image img = {w, h, rgba};
    void *smp = __translate_sampler_initializer(sampler_bits);
    int2 d = o_get_image_dim(&img);
    dim[0] = d.x; dim[1] = d.y;
    for (int i = 0; i < n; i++) {
        float2 c; c.x = uv[2 * i]; c.y = uv[2 * i + 1];
        float4 r = o_read_imagef_f(&img, smp, c);
        o[4 * i] = r.x; o[4 * i + 1] = r.y; o[4 * i + 2] = r.z; o[4 * i + 3] = r.w;
    }
    return 0;
}
