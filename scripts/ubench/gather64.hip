// Microbenchmark: cost of fetching one random 64-byte record per lane on gfx950.
//  A: each lane issues 4 x dwordx4 to its own record (what the trace kernel does)
//  B: quad-cooperative: 4 passes, lane l fetches piece (l&3) of record of lane 16p+(l>>2); LDS transpose
//  C: like B without the LDS transpose (pure load cost, result reduced arithmetically)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void kA(const float4 *rec, const uint32_t *idx, float *out, int iters, uint32_t nrec)
{
    uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t i = idx[t];
    float acc = 0.f;
    for (int k = 0; k < iters; k++) {
        const float4 *p = rec + (size_t)i * 4;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        i = (__float_as_uint(d.x) ^ (uint32_t)k) % nrec;     // dependent chase, like a traversal
    }
    out[t] = acc;
}

__global__ __launch_bounds__(64) void kB(const float4 *rec, const uint32_t *idx, float *out, int iters, uint32_t nrec)
{
    __shared__ float4 s[64 * 4];
    uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t lane = threadIdx.x;
    uint32_t i = idx[t];
    float acc = 0.f;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            uint32_t src = 16 * p + (lane >> 2);
            uint32_t ri = __shfl(i, src, 64);
            float4 v = rec[(size_t)ri * 4 + (lane & 3)];
            s[src * 4 + (lane & 3)] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);   // wave-private LDS region, no barrier needed (single wave)
        float4 a = s[lane * 4 + 0], b = s[lane * 4 + 1], c = s[lane * 4 + 2], d = s[lane * 4 + 3];
        acc += a.x + b.y + c.z + d.w;
        i = (__float_as_uint(d.x) ^ (uint32_t)k) % nrec;
    }
    out[t] = acc;
}

int main()
{
    const uint32_t nrec = 1u << 18;            // 16 MiB of 64-B records (L2/MALL resident like the BVH)
    const uint32_t nthreads = 1u << 20;
    const int iters = 24;
    std::vector<float4> h((size_t)nrec * 4);
    uint32_t s = 12345;
    for (size_t r = 0; r < nrec; r++) for (int q = 0; q < 4; q++) {
        s = s * 1664525u + 1013904223u; uint32_t nx = s >> 8;
        float f; memcpy(&f, &nx, 4);
        h[r * 4 + q] = make_float4(q == 3 ? f : 1.0f, 0.5f, 0.25f, 0.125f);
    }
    std::vector<uint32_t> hi(nthreads);
    for (auto &v : hi) { s = s * 1664525u + 1013904223u; v = (s >> 8) % nrec; }
    float4 *rec; uint32_t *idx; float *out;
    CHECK(hipMalloc(&rec, h.size() * 16)); CHECK(hipMalloc(&idx, nthreads * 4)); CHECK(hipMalloc(&out, nthreads * 4));
    CHECK(hipMemcpy(rec, h.data(), h.size() * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx, hi.data(), nthreads * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(kA, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec);
            else hipLaunchKernelGGL(kB, dim3(nthreads / 64), dim3(64), 0, 0, rec, idx, out, iters, nrec);
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s: %.3f ms  %.2f G records/s  %.0f GB/s\n", variant == 0 ? "A per-lane 4xdwordx4" : "B quad-coop + LDS", ms,
                                 (double)nthreads * iters / ms / 1e6, (double)nthreads * iters * 64 / ms / 1e6);
        }
    }
    std::vector<float> ho(16); CHECK(hipMemcpy(ho.data(), out, 64, hipMemcpyDeviceToHost)); printf("chk %f\n", ho[3]);
    return 0;
}
