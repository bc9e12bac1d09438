// Microbenchmark: random per-lane record gathers on gfx950 -- record size x working-set sweep (dependent chase).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int Q>   // Q x 16 bytes per record
__global__ __launch_bounds__(64) void k(const float4 *rec, const uint32_t *idx, float *out, int iters, uint32_t nrec)
{
    uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t i = idx[t] % nrec;
    float acc = 0.f;
    for (int kk = 0; kk < iters; kk++) {
        const float4 *p = rec + (size_t)i * Q;
        float4 v[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) v[q] = p[q];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < Q; q++) s += v[q].y;
        acc += s;
        i = (__float_as_uint(v[Q - 1].x) ^ (uint32_t)kk) % nrec;
    }
    out[t] = acc;
}

int main()
{
    const uint32_t maxBytes = 256u << 20;
    const uint32_t nthreads = 1u << 20;
    const int iters = 24;
    std::vector<float4> h(maxBytes / 16);
    uint32_t s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; uint32_t nx = s >> 4; float f; memcpy(&f, &nx, 4); v = make_float4(f, 0.5f, 0.25f, 0.125f); }
    std::vector<uint32_t> hi(nthreads);
    for (auto &v : hi) { s = s * 1664525u + 1013904223u; v = (s >> 4); }
    float4 *rec; uint32_t *idx; float *out;
    CHECK(hipMalloc(&rec, maxBytes)); CHECK(hipMalloc(&idx, nthreads * 4)); CHECK(hipMalloc(&out, nthreads * 4));
    CHECK(hipMemcpy(rec, h.data(), maxBytes, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), nthreads * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t ws[] = {1u << 20, 2u << 20, 8u << 20, 32u << 20, 128u << 20, 256u << 20};
    for (int Q : {1, 2, 3, 4, 8}) for (uint32_t w : ws) {
        uint32_t nrec = w / (16 * Q);
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            switch (Q) {
            case 1: hipLaunchKernelGGL(k<1>, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec); break;
            default: hipLaunchKernelGGL(k<8>, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec); break;
            }
            (void)hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("rec %3d B  working set %4u MiB: %.3f ms  %6.1f G rec/s  %6.0f GB/s\n", 16 * Q, w >> 20, ms, (double)nthreads * iters / ms / 1e6,
               (double)nthreads * iters * 16 * Q / ms / 1e6);
    }
    return 0;
}
