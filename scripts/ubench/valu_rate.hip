// Microbenchmark: issue cost of the instruction sequences the 4-wide node test could be built from, on gfx950.
// Every kernel runs REPS x an unrolled block of N independent-ish instructions per wave, W waves per SIMD; the figure is
// SIMD-cycles per wave-instruction (2.0 = full rate on CDNA4's SIMD-32 with a wave64).
//   fma        v_fma_f32
//   cvt_fma    v_cvt_f32_ubyteN + v_fma_f32                (the round-2 plane evaluation: 2 instructions per plane)
//   mix        v_fma_mix_f32 with a packed-half operand    (1 instruction per plane, halves 0x6400 | byte)
//   perm       v_perm_b32
//   pk_fma     v_pk_fma_f32                                (2 planes per instruction)
//   cndmask    v_cndmask_b32 with an SGPR-pair mask
//   minmax     v_min_i32 / v_max_i32
//   med3       v_med3_f32
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REPS 4096

#define R8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in)
{
    float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float s = in[64 + (threadIdx.x & 63)], o = in[128 + (threadIdx.x & 63)];
    uint32_t q = __float_as_uint(in[192 + (threadIdx.x & 63)]);
    for (int r = 0; r < REPS; r++) {
        if (MODE == 0) {
            asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o));
        } else if (MODE == 1) {
            asm volatile(R8("v_cvt_f32_ubyte0 %0, %10\n v_fma_f32 %0, %0, %8, %9\n v_cvt_f32_ubyte1 %1, %10\n v_fma_f32 %1, %1, %8, %9\n"
                            "v_cvt_f32_ubyte2 %2, %10\n v_fma_f32 %2, %2, %8, %9\n v_cvt_f32_ubyte3 %3, %10\n v_fma_f32 %3, %3, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q));
        } else if (MODE == 2) {
            asm volatile(R8("v_fma_mix_f32 %0, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                            "v_fma_mix_f32 %2, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                            "v_fma_mix_f32 %4, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                            "v_fma_mix_f32 %6, %10, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q));
        } else if (MODE == 3) {
            asm volatile(R8("v_perm_b32 %0, %0, %10, %8\n v_perm_b32 %1, %1, %10, %8\n v_perm_b32 %2, %2, %10, %8\n v_perm_b32 %3, %3, %10, %8\n"
                            "v_perm_b32 %4, %4, %10, %8\n v_perm_b32 %5, %5, %10, %8\n v_perm_b32 %6, %6, %10, %8\n v_perm_b32 %7, %7, %10, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q));
        } else if (MODE == 4) {
            asm volatile(R8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                            "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&s), "v"(*(double *)&o));
        } else if (MODE == 5) {
            asm volatile(R8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                            "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o) : "vcc");
        } else if (MODE == 6) {
            asm volatile(R8("v_min_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_min_i32 %2, %2, %9\n v_max_i32 %3, %3, %9\n"
                            "v_min_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_min_i32 %6, %6, %9\n v_max_i32 %7, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o));
        } else if (MODE == 7) {
            asm volatile(R8("v_med3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                            "v_max3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o));
        } else if (MODE == 8) {      // compare into an SGPR pair + cndmask on it (the sort's compare-exchange building block)
            asm volatile(R8("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n"
                            "v_cmp_lt_f32 s[22:23], %2, %3\n v_cndmask_b32 %0, %0, %1, s[22:23]\n v_cndmask_b32 %4, %4, %5, s[22:23]\n v_cndmask_b32 %6, %6, %7, s[22:23]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o) : "s20", "s21", "s22", "s23");
        } else if (MODE == 9) {      // SDWA byte add (ref = base + byte of a key)
            asm volatile(R8("v_add_u32_sdwa %0, %10, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %1, %10, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                            "v_add_u32_sdwa %2, %10, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %3, %10, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                            "v_add_u32_sdwa %4, %10, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa %5, %10, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                            "v_add_u32_sdwa %6, %10, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_add_u32_sdwa %7, %10, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o), "v"(q));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static void run(const char *name, int instr_per_block, int waves_per_simd, float *out, const float *in)
{
    // 256 CUs x 4 SIMDs x waves_per_simd waves, 256-thread blocks (4 waves, one per SIMD)
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)REPS * 8 * instr_per_block;
    const double simd_cycles = ms * 1e-3 * 2.4e9;           // nominal clock; the ratio between rows is what matters
    printf("%-10s waves/SIMD %d : %.3f ms, %.2f SIMD-cycles per wave-instruction (at 2.4 GHz nominal)\n", name, waves_per_simd, ms,
           simd_cycles / (instr_per_wave * waves_per_simd));
}

int main()
{
    float *out, *in;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&in, 1024 * sizeof(float)));
    float h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 1.0f + i * 1e-3f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int w : {1, 2, 4, 8}) {
        run<0>("fma", 8, w, out, in);
        run<1>("cvt_fma", 8, w, out, in);
        run<2>("mix", 8, w, out, in);
        run<3>("perm", 8, w, out, in);
        run<4>("pk_fma", 8, w, out, in);
        run<5>("cndmask", 8, w, out, in);
        run<6>("minmax", 8, w, out, in);
        run<7>("med3", 8, w, out, in);
        run<8>("cmp+cnd", 8, w, out, in);
        run<9>("sdwa_add", 8, w, out, in);
    }
    return 0;
}
