// Microbenchmark: the memory floor of the fused logic pass's ACCESS PATTERN on gfx950, no arithmetic.
// Per path (thread = path, 256-thread blocks, like k_logic): read six 16-byte records + two 4-byte scalars, [MODE >= 1] for ~26 % of the lanes four
// more records behind the value of the first scalar (the NEE consume), [MODE >= 2] a 64-byte gather from a 30 MB table behind one of the records
// (the RAW commit's shading record); write twelve 16-byte records (MODE 3: ten, MODE 4: eight) + one byte.  Non-temporal loads / stores like
// flx_device.h.  Each mode at W waves per SIMD (amdgpu_waves_per_eu) for W = 5 (what k_logic gets at 93 VGPRs) and 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rd4(const float4 *p) { v4f v = __builtin_nontemporal_load((const v4f *)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void wr4(float4 *p, float4 v) { v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, (v4f *)p); }
struct St { float4 *rec[12]; uint32_t *blocked; float *pick; uint8_t *member; const float4 *shade; uint32_t nshade; uint32_t n; };

template <int MODE_, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void k(St st)
{
    constexpr int MODE = MODE_ % 10;
    constexpr bool TEMPORAL = MODE_ >= 10;        // MODE_ 1x: plain (temporal, write-back) stores instead of non-temporal ones
    auto wr4 = [](float4 *p, float4 v) { if (TEMPORAL) *p = v; else ::wr4(p, v); };
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= st.n) return;
    const uint32_t b = st.blocked[gid]; const float pk = st.pick[gid];
    float4 r0 = rd4(st.rec[0] + gid), r1 = rd4(st.rec[1] + gid), r2 = rd4(st.rec[2] + gid), r3 = rd4(st.rec[3] + gid), r4 = rd4(st.rec[4] + gid), r5 = rd4(st.rec[5] + gid);
    float4 acc = make_float4(r0.x + r1.x + pk, r2.y + r3.y, r4.z + r5.z, r0.w);
    if (MODE >= 1 && b == 0u) {
        float4 a = rd4(st.rec[8] + gid), c = rd4(st.rec[9] + gid), d = rd4(st.rec[10] + gid), e = rd4(st.rec[11] + gid);
        acc.x += a.x * c.y; acc.y += d.z * e.w;
    }
    if (MODE >= 2) {
        const float4 *sp = st.shade + (size_t)(__float_as_uint(r4.z) % st.nshade) * 4;
        float4 a = sp[0], c = sp[1], d = sp[2], e = sp[3];
        acc.z += a.x + c.y + d.z + e.w;
    }
    const int NW = MODE == 3 ? 10 : MODE == 4 ? 8 : 12;
    // MODE 5 / 6: 21 % of the lanes (the terminating paths; hash of the id, so the holes fall on random lanes) store only 3 of the 12 records --
    // partial 128-byte lines and partial 32-byte sectors in the other nine streams; MODE 6 adds their four float atomics on a 33 MB framebuffer
    const bool term = (MODE == 5 || MODE == 6 || MODE == 7) && ((gid * 2654435761u) >> 8) % 100u < 21u;
    #pragma unroll
    for (int r = 0; r < NW; r++) if (!term || r == 4 || r == 5 || r == 6) wr4(st.rec[r] + gid, make_float4(acc.x + r, acc.y, acc.z, r == 4 ? r4.z : acc.w));
    if (MODE == 7) {       // the union of two DIVERGENT partial stores covers every line: does the L2 merge them before they reach HBM?
        const bool t2 = ((gid * 2654435761u) >> 8) % 100u < 21u;
        if (t2) {
            #pragma unroll
            for (int r = 0; r < 12; r++) if (!(r == 4 || r == 5 || r == 6)) wr4(st.rec[r] + gid, make_float4(acc.y + r, acc.x, acc.z, acc.w));
        }
    }
    if (MODE == 6 && term) {
        float *px = (float *)st.rec[7] + (size_t)((gid * 40503u) % 2073600u) * 4;
        unsafeAtomicAdd(px, acc.x); unsafeAtomicAdd(px + 1, acc.y); unsafeAtomicAdd(px + 2, acc.z); unsafeAtomicAdd(px + 3, 1.0f);
    }
    st.member[gid] = (uint8_t)(b + 1u);
}

template <int MODE, int W>
static float run(St st, int reps)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 g((st.n + 255) / 256), b(256);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, W>), g, b, 0, 0, st);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<MODE, W>), g, b, 0, 0, st);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    St st; st.n = 1u << 23; st.nshade = 30u * 1024 * 1024 / 64;
    for (int r = 0; r < 12; r++) { CHECK(hipMalloc(&st.rec[r], (size_t)st.n * 16)); CHECK(hipMemset(st.rec[r], 0, (size_t)st.n * 16)); }
    CHECK(hipMalloc(&st.blocked, st.n * 4)); CHECK(hipMalloc(&st.pick, st.n * 4)); CHECK(hipMalloc(&st.member, st.n));
    float4 *sh; CHECK(hipMalloc(&sh, (size_t)st.nshade * 64)); CHECK(hipMemset(sh, 0, (size_t)st.nshade * 64)); st.shade = sh;
    std::vector<uint32_t> hb(st.n); std::vector<float4> h4(st.n);
    uint32_t s = 12345u;
    for (uint32_t i = 0; i < st.n; i++) { s = s * 1664525u + 1013904223u; hb[i] = ((s >> 8) % 100u) < 26u ? 0u : 1u; h4[i] = make_float4(0.f, 0.f, 0.f, 0.f); uint32_t t = s ^ (s >> 13); h4[i].z = *(float *)&t; }
    CHECK(hipMemcpy(st.blocked, hb.data(), st.n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(st.pick, 0, st.n * 4));
    const int reps = 20;
    const double rd0 = 6 * 16 + 8, wr12 = 12 * 16 + 1;
#define RUN(M, W, bytes) do { CHECK(hipMemcpy(st.rec[4], h4.data(), (size_t)st.n * 16, hipMemcpyHostToDevice)); float ms = run<M, W>(st, reps); \
    printf("mode %d  %d waves/SIMD  %.3f ms  %.2f TB/s of %.0f B/path (state bytes; the 64-B gather not counted)\n", M, W, ms, (bytes) * st.n / ms / 1e9, (double)(bytes)); } while (0)
    RUN(0, 5, rd0 + wr12); RUN(0, 8, rd0 + wr12);
    RUN(1, 5, rd0 + wr12 + 0.26 * 64); RUN(1, 8, rd0 + wr12 + 0.26 * 64);
    RUN(2, 5, rd0 + wr12 + 0.26 * 64); RUN(2, 8, rd0 + wr12 + 0.26 * 64);
    RUN(3, 5, rd0 + 10 * 16 + 1 + 0.26 * 64); RUN(4, 5, rd0 + 8 * 16 + 1 + 0.26 * 64);
    RUN(2, 5, rd0 + wr12 + 0.26 * 64);
    RUN(5, 5, rd0 + (0.79 * 12 + 0.21 * 3) * 16 + 1 + 0.26 * 64); RUN(6, 5, rd0 + (0.79 * 12 + 0.21 * 3) * 16 + 1 + 0.26 * 64);
    RUN(2, 5, rd0 + wr12 + 0.26 * 64); RUN(5, 5, rd0 + (0.79 * 12 + 0.21 * 3) * 16 + 1 + 0.26 * 64);
    RUN(7, 5, rd0 + wr12 + 0.26 * 64); RUN(2, 5, rd0 + wr12 + 0.26 * 64); RUN(7, 5, rd0 + wr12 + 0.26 * 64);
    RUN(12, 5, rd0 + wr12 + 0.26 * 64); RUN(15, 5, rd0 + (0.79 * 12 + 0.21 * 3) * 16 + 1 + 0.26 * 64); RUN(17, 5, rd0 + wr12 + 0.26 * 64);
    RUN(12, 5, rd0 + wr12 + 0.26 * 64); RUN(15, 5, rd0 + (0.79 * 12 + 0.21 * 3) * 16 + 1 + 0.26 * 64); RUN(17, 5, rd0 + wr12 + 0.26 * 64);
    return 0;
}
