// Microbenchmark 2: which VALU instructions of gfx950 issue beside a v_fma_f32 (second pipe) and which share its issue slots.
// valu_rate.hip showed: plain VALU streams (v_fma_f32, v_perm_b32, v_min_i32, v_med3_f32, v_cndmask, SDWA adds) run at ~4 SIMD-cycles per
// wave64 instruction however many waves are resident, but an alternating v_cvt_f32_ubyteN / v_fma_f32 stream runs at 2.4 -- the
// conversion rides along.  For each candidate X this prints cycles per instruction of a stream of X alone and of X alternating with
// v_fma_f32 (pair cost ~4 => X is free beside an fma; ~8 => X takes a full slot of its own).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_pairs valu_pairs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REPS 2048
#define R8(x) x x x x x x x x

// X(d) : one instruction writing register %d (d in 0..3) from %d and the shared operands %8 (s), %9 (o), %10 (q)
#define ALONE(NAME, X0, X1, X2, X3) \
__global__ __launch_bounds__(256) void alone_##NAME(float *out, const float *in) { \
    float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    float s = in[64 + (threadIdx.x & 63)], o = in[128 + (threadIdx.x & 63)]; uint32_t q = __float_as_uint(in[192 + (threadIdx.x & 63)]); \
    const long long t0 = clock64(); \
    for (int r = 0; r < REPS; r++) asm volatile(R8(X0 "\n" X1 "\n" X2 "\n" X3 "\n" X0 "\n" X1 "\n" X2 "\n" X3 "\n") \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q) : "vcc", "scc", "s20", "s21"); \
    const long long t1 = clock64(); \
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long *>(out)[1 << 18] = t1 - t0; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; } \
__global__ __launch_bounds__(256) void pair_##NAME(float *out, const float *in) { \
    float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    float s = in[64 + (threadIdx.x & 63)], o = in[128 + (threadIdx.x & 63)]; uint32_t q = __float_as_uint(in[192 + (threadIdx.x & 63)]); \
    const long long t0 = clock64(); \
    for (int r = 0; r < REPS; r++) asm volatile(R8(X0 "\n v_fma_f32 %4, %4, %8, %9\n" X1 "\n v_fma_f32 %5, %5, %8, %9\n" X2 "\n v_fma_f32 %6, %6, %8, %9\n" X3 "\n v_fma_f32 %7, %7, %8, %9\n") \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q) : "vcc", "scc", "s20", "s21"); \
    const long long t1 = clock64(); \
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long *>(out)[1 << 18] = t1 - t0; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }

#define OP4(NAME, FMT) ALONE(NAME, FMT(0), FMT(1), FMT(2), FMT(3))

#define F_FMA(d) "v_fma_f32 %" #d ", %" #d ", %8, %9"
#define F_CVT(d) "v_cvt_f32_ubyte" #d " %" #d ", %10"
#define F_CVTU(d) "v_cvt_f32_u32 %" #d ", %" #d
#define F_MOV(d) "v_mov_b32 %" #d ", %8"
#define F_MUL(d) "v_mul_f32 %" #d ", %" #d ", %8"
#define F_ADD(d) "v_add_f32 %" #d ", %" #d ", %8"
#define F_MAX(d) "v_max_f32 %" #d ", %" #d ", %8"
#define F_MAX3(d) "v_max3_f32 %" #d ", %" #d ", %8, %9"
#define F_AND(d) "v_and_b32 %" #d ", %" #d ", %10"
#define F_LSHL(d) "v_lshlrev_b32 %" #d ", 3, %" #d
#define F_ADDU(d) "v_add_u32 %" #d ", %" #d ", %10"
#define F_PERM(d) "v_perm_b32 %" #d ", %" #d ", %10, %8"
#define F_CNDS(d) "v_cndmask_b32 %" #d ", %" #d ", %8, s[20:21]"
#define F_CMP(d) "v_cmp_lt_f32 s[20:21], %" #d ", %8"
#define F_RCP(d) "v_rcp_f32 %" #d ", %" #d
#define F_MIX(d) "v_fma_mix_f32 %" #d ", %10, %8, %" #d " op_sel_hi:[1,0,0]"
#define F_BFE(d) "v_bfe_u32 %" #d ", %" #d ", 8, 8"
#define F_MINI(d) "v_min_i32 %" #d ", %" #d ", %10"
#define F_FMAC(d) "v_fmac_f32 %" #d ", %8, %9"
#define F_PKMUL(d) "v_mul_f32 %" #d ", %" #d ", %" #d
#define F_CVTH(d) "v_cvt_f32_f16 %" #d ", %10"
#define F_NOP(d) "s_nop 0"
#define F_SADD(d) "s_add_u32 s20, s20, 1"

OP4(fma, F_FMA) OP4(cvt_ubyte, F_CVT) OP4(cvt_u32, F_CVTU) OP4(mov, F_MOV) OP4(mul, F_MUL) OP4(add, F_ADD) OP4(max, F_MAX) OP4(max3, F_MAX3)
OP4(and_b32, F_AND) OP4(lshl, F_LSHL) OP4(add_u32, F_ADDU) OP4(perm, F_PERM) OP4(cndmask_s, F_CNDS) OP4(cmp, F_CMP) OP4(rcp, F_RCP) OP4(mix, F_MIX)
OP4(bfe, F_BFE) OP4(min_i32, F_MINI) OP4(fmac, F_FMAC) OP4(cvt_f16, F_CVTH) OP4(s_nop, F_NOP) OP4(s_add, F_SADD)

typedef void (*kern_t)(float *, const float *);
static double g_ticks = 0, g_ms = 0;
static double run(kern_t k, float *out, const float *in, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long ticks = 0; CHECK(hipMemcpy(&ticks, reinterpret_cast<long long *>(out) + (1 << 18), 8, hipMemcpyDeviceToHost));
    g_ticks = (double)ticks; g_ms = ms;
    return ms * 1e-3 * 2.4e9 / ((double)REPS * 64 * waves_per_simd);       // nominal SIMD-cycles per instruction (64 per rep)
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float *out, *in;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float) + (1 << 21) + 64));
    CHECK(hipMalloc(&in, 1024 * sizeof(float)));
    float h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 1.0f + i * 1e-3f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    printf("%-12s %28s %28s\n", "op", "alone: cyc/instr @2,4,8 waves", "X+fma: cyc per PAIR @2,4,8 waves");
#define ROW(NAME) { printf("%-12s", #NAME); for (int w : {2, 4, 8}) printf(" %8.2f", run(alone_##NAME, out, in, w)); printf("   |"); \
                    for (int w : {2, 4, 8}) printf(" %8.2f", 2.0 * run(pair_##NAME, out, in, w)); \
                    printf("   | wave 0 of the last run: %.0f clock64 ticks in %.3f ms of kernel = %.0f MHz if the wave lived the whole kernel\n", g_ticks, g_ms, g_ticks / (g_ms * 1e3)); }
    ROW(fma) ROW(cvt_ubyte) ROW(cvt_u32) ROW(cvt_f16) ROW(mov) ROW(mul) ROW(add) ROW(max) ROW(max3) ROW(and_b32) ROW(lshl) ROW(add_u32) ROW(bfe) ROW(min_i32)
    ROW(perm) ROW(cndmask_s) ROW(cmp) ROW(rcp) ROW(mix) ROW(fmac) ROW(s_nop) ROW(s_add)
    return 0;
}
