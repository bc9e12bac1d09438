// Microbenchmark 3: pipe membership of gfx950 VALU instructions.  valu_pairs.hip showed two issue classes: "flexible" instructions
// (v_mov / v_mul_f32 / v_add_f32 / v_and_b32 / v_add_u32 / v_fmac_f32: ~2 SIMD-cycles each in a stream of their own) and "single-pipe" ones
// (v_cvt_*, v_max / v_min, shifts, v_bfe, v_cmp, v_cndmask, v_perm, 3-operand min/max: ~4 cycles each in a stream of their own, but ~2 when
// alternating with an fma).  This one alternates X and Y for pairs of candidates and prints cycles per PAIR:
//   ~4   -> X and Y issue side by side (different pipes, or both flexible)
//   ~8   -> X and Y serialise on the same pipe
// build: hipcc --offload-arch=gfx950 -O3 -o valu_pairs3 valu_pairs2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REPS 2048
#define R8(x) x x x x x x x x

#define PAIR(NAME, X, Y) \
__global__ __launch_bounds__(256) void pair_##NAME(float *out, const float *in) { \
    float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    float s = in[64 + (threadIdx.x & 63)], o = in[128 + (threadIdx.x & 63)]; uint32_t q = __float_as_uint(in[192 + (threadIdx.x & 63)]); \
    for (int r = 0; r < REPS; r++) asm volatile(R8(X(0) "\n" Y(4) "\n" X(1) "\n" Y(5) "\n" X(2) "\n" Y(6) "\n" X(3) "\n" Y(7) "\n") \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q) : "vcc", "scc", "s20", "s21", "s22", "s23"); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }

#define F_FMA(d) "v_fma_f32 %" #d ", %" #d ", %8, %9"
#define F_CVT(d) "v_cvt_f32_ubyte1 %" #d ", %10"
#define F_MOV(d) "v_mov_b32 %" #d ", %8"
#define F_MUL(d) "v_mul_f32 %" #d ", %" #d ", %8"
#define F_ADD(d) "v_add_f32 %" #d ", %" #d ", %8"
#define F_SUB(d) "v_sub_f32 %" #d ", %" #d ", %8"
#define F_MAX(d) "v_max_f32 %" #d ", %" #d ", %8"
#define F_MAX3(d) "v_max3_f32 %" #d ", %" #d ", %8, %9"
#define F_AND(d) "v_and_b32 %" #d ", %" #d ", %10"
#define F_OR(d) "v_or_b32 %" #d ", %" #d ", %10"
#define F_XOR(d) "v_xor_b32 %" #d ", %" #d ", %10"
#define F_LSHL(d) "v_lshlrev_b32 %" #d ", 3, %" #d
#define F_ADDU(d) "v_add_u32 %" #d ", %" #d ", %10"
#define F_SUBU(d) "v_sub_u32 %" #d ", %" #d ", %10"
#define F_PERM(d) "v_perm_b32 %" #d ", %" #d ", %10, %8"
#define F_CNDS(d) "v_cndmask_b32 %" #d ", %" #d ", %8, s[20:21]"
#define F_CNDV(d) "v_cndmask_b32 %" #d ", %" #d ", %8, vcc"
#define F_CMP(d) "v_cmp_lt_f32 s[22:23], %" #d ", %8"
#define F_CMPV(d) "v_cmp_lt_f32 vcc, %" #d ", %8"
#define F_MINI(d) "v_min_i32 %" #d ", %" #d ", %10"
#define F_FMAC(d) "v_fmac_f32 %" #d ", %8, %9"
#define F_ANDOR(d) "v_and_or_b32 %" #d ", %" #d ", %10, %8"
#define F_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %10, %8"
#define F_LSHLADD(d) "v_lshl_add_u32 %" #d ", %" #d ", 3, %10"
#define F_BFI(d) "v_bfi_b32 %" #d ", %10, %" #d ", %8"
#define F_MAD24(d) "v_mad_u32_u24 %" #d ", %" #d ", %10, %8"
#define F_MULLO(d) "v_mul_u32_u24 %" #d ", %" #d ", %10"
#define F_MED3(d) "v_med3_f32 %" #d ", %" #d ", %8, %9"
#define F_FMAMK(d) "v_fmamk_f32 %" #d ", %" #d ", 0x3f000000, %9"
#define F_SNOP(d) "s_nop 0"
#define F_SAND(d) "s_and_b64 s[20:21], s[20:21], s[22:23]"
#define F_DSW(d) "ds_write_b32 %10, %" #d

PAIR(fma_fma, F_FMA, F_FMA) PAIR(mul_mul, F_MUL, F_MUL) PAIR(cvt_cvt, F_CVT, F_CVT)
PAIR(cvt_max, F_CVT, F_MAX) PAIR(cvt_cnd, F_CVT, F_CNDS) PAIR(cvt_cmp, F_CVT, F_CMP) PAIR(cmp_cnd, F_CMP, F_CNDS) PAIR(cmpv_cndv, F_CMPV, F_CNDV)
PAIR(cnd_max3, F_CNDS, F_MAX3) PAIR(cvt_lshl, F_CVT, F_LSHL) PAIR(cvt_mul, F_CVT, F_MUL) PAIR(cnd_mul, F_CNDS, F_MUL) PAIR(cnd_addu, F_CNDS, F_ADDU)
PAIR(cmp_mul, F_CMP, F_MUL) PAIR(max3_mul, F_MAX3, F_MUL) PAIR(perm_mul, F_PERM, F_MUL) PAIR(max_mul, F_MAX, F_MUL) PAIR(mini_mul, F_MINI, F_MUL)
PAIR(or_or, F_OR, F_OR) PAIR(xor_xor, F_XOR, F_XOR) PAIR(sub_sub, F_SUB, F_SUB) PAIR(subu_subu, F_SUBU, F_SUBU)
PAIR(andor_andor, F_ANDOR, F_ANDOR) PAIR(andor_mul, F_ANDOR, F_MUL) PAIR(add3_add3, F_ADD3, F_ADD3) PAIR(add3_mul, F_ADD3, F_MUL)
PAIR(lshladd_lshladd, F_LSHLADD, F_LSHLADD) PAIR(bfi_bfi, F_BFI, F_BFI) PAIR(bfi_mul, F_BFI, F_MUL) PAIR(mad24_mad24, F_MAD24, F_MAD24)
PAIR(mullo_mullo, F_MULLO, F_MULLO) PAIR(med3_mul, F_MED3, F_MUL) PAIR(fmamk_fmamk, F_FMAMK, F_FMAMK) PAIR(fma_mul, F_FMA, F_MUL)
PAIR(fma_cnd, F_FMA, F_CNDS) PAIR(fma_cvt, F_FMA, F_CVT) PAIR(mul_snop, F_MUL, F_SNOP) PAIR(cnd_sand, F_CNDS, F_SAND) PAIR(mul_sand, F_MUL, F_SAND)
PAIR(max_max, F_MAX, F_MAX) PAIR(cnd_cnd, F_CNDS, F_CNDS) PAIR(cmp_cmp, F_CMP, F_CMP) PAIR(max3_max3, F_MAX3, F_MAX3) PAIR(lshl_lshl, F_LSHL, F_LSHL)

typedef void (*kern_t)(float *, const float *);
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
    return ms * 1e-3 * 2.4e9 / ((double)REPS * 32 * waves_per_simd);       // nominal SIMD-cycles per PAIR (32 pairs per rep)
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float *out, *in;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&in, 1024 * sizeof(float)));
    float h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 1.0f + i * 1e-3f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    printf("%-18s cycles per PAIR @ 2, 4, 8 waves per SIMD (nominal 2.4 GHz)\n", "pair");
#define ROW(NAME) { printf("%-18s", #NAME); for (int w : {2, 4, 8}) printf(" %8.2f", run(pair_##NAME, out, in, w)); printf("\n"); }
    ROW(fma_fma) ROW(mul_mul) ROW(cvt_cvt) ROW(max_max) ROW(cnd_cnd) ROW(cmp_cmp) ROW(max3_max3) ROW(lshl_lshl)
    ROW(cvt_max) ROW(cvt_cnd) ROW(cvt_cmp) ROW(cmp_cnd) ROW(cmpv_cndv) ROW(cnd_max3) ROW(cvt_lshl)
    ROW(cvt_mul) ROW(cnd_mul) ROW(cnd_addu) ROW(cmp_mul) ROW(max3_mul) ROW(perm_mul) ROW(max_mul) ROW(mini_mul) ROW(med3_mul)
    ROW(or_or) ROW(xor_xor) ROW(sub_sub) ROW(subu_subu) ROW(andor_andor) ROW(andor_mul) ROW(add3_add3) ROW(add3_mul) ROW(lshladd_lshladd)
    ROW(bfi_bfi) ROW(bfi_mul) ROW(mad24_mad24) ROW(mullo_mullo) ROW(fmamk_fmamk) ROW(fma_mul) ROW(fma_cnd) ROW(fma_cvt)
    ROW(mul_snop) ROW(cnd_sand) ROW(mul_sand)
    return 0;
}
