// Microbenchmark: throughput law of the vector-memory path (TA/TCP) on gfx950 for random record gathers.
// Independent (non-chained) addresses so that latency is hidden and the steady-state request rate shows.
//   MODE 0: every active lane reads its own 64-B record with 4 x dwordx4 (what the traversal kernels do)
//   MODE 1: every active lane reads 16 B of 4 different records (same number of lane-loads, 4x the lines)
//   MODE 2: quad-cooperative: the 4 lanes of a quad read the 4 chunks of ONE record per instruction, 4 instructions
//           cover the quad's 4 records (same bytes and lane-loads as MODE 0, lines per instruction / 4)
//   MODE 3: one dwordx4 per lane per record (16-B records)
// ACTIVE = number of active lanes per wave (the others are masked off by a branch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(64) void k(const float4 *rec, float *out, int iters, uint32_t nrec, int active)
{
    const uint32_t lane = threadIdx.x;
    uint32_t t = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if ((int)lane < active) {
        uint32_t s = hash(t + 1u);
        for (int kk = 0; kk < iters; kk++) {
            s = hash(s + (uint32_t)kk);
            float4 v0, v1, v2, v3;
            if (MODE == 0) {
                const float4 *p = rec + (size_t)(s % nrec) * 4;
                v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3];
            } else if (MODE == 1) {
                v0 = rec[(size_t)(s % nrec) * 4]; v1 = rec[(size_t)(hash(s) % nrec) * 4 + 1];
                v2 = rec[(size_t)(hash(s + 7u) % nrec) * 4 + 2]; v3 = rec[(size_t)(hash(s + 13u) % nrec) * 4 + 3];
            } else if (MODE == 2) {
                // records of the quad's 4 lanes: r_j = record chosen by lane (quadbase + j); all 4 lanes compute all 4
                const uint32_t qb = (t & ~3u), c = lane & 3u;
                uint32_t r0 = hash(hash(qb + 1u) + (uint32_t)kk * 4u + 0u) % nrec, r1 = hash(hash(qb + 2u) + (uint32_t)kk * 4u + 1u) % nrec;
                uint32_t r2 = hash(hash(qb + 3u) + (uint32_t)kk * 4u + 2u) % nrec, r3 = hash(hash(qb + 4u) + (uint32_t)kk * 4u + 3u) % nrec;
                v0 = rec[(size_t)r0 * 4 + c]; v1 = rec[(size_t)r1 * 4 + c]; v2 = rec[(size_t)r2 * 4 + c]; v3 = rec[(size_t)r3 * 4 + c];
            } else {
                v0 = rec[(size_t)(s % (nrec * 4))]; v1 = v2 = v3 = v0;
            }
            acc += v0.y + v1.z + v2.w + v3.x;
        }
    }
    out[t] = acc;
}

int main()
{
    const size_t maxBytes = 256u << 20;
    const uint32_t nthreads = 1u << 20;
    const int iters = 32;
    std::vector<float4> h(maxBytes / 16, make_float4(1.f, 0.5f, 0.25f, 0.125f));
    float4 *rec; float *out;
    CHECK(hipMalloc(&rec, maxBytes)); CHECK(hipMalloc(&out, nthreads * 4));
    CHECK(hipMemcpy(rec, h.data(), maxBytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t ws[] = {2u << 20, 32u << 20, 256u << 20};
    for (uint32_t w : ws) for (int mode = 0; mode < 4; mode++) for (int active : {64, 32, 16, 8}) {
        uint32_t nrec = w / 64;
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(nthreads / 64), dim3(64), 0, 0, rec, out, iters, nrec, active); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(nthreads / 64), dim3(64), 0, 0, rec, out, iters, nrec, active); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(nthreads / 64), dim3(64), 0, 0, rec, out, iters, nrec, active); break;
            default: hipLaunchKernelGGL(k<3>, dim3(nthreads / 64), dim3(64), 0, 0, rec, out, iters, nrec, active); break;
            }
            (void)hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double waves = nthreads / 64.0, waveLoads = waves * iters * (mode == 3 ? 1 : 4), laneLoads = waveLoads * active;
        const double cuCycles = ms * 1e-3 * 2.4e9 * 256;
        printf("ws %3u MiB mode %d active %2d: %.3f ms  %6.1f G lane-loads/s  %5.2f CU-cycles/lane-load  %6.1f CU-cycles/wave-load  %6.0f GB/s\n", w >> 20, mode, active, ms,
               laneLoads / ms / 1e6, cuCycles / laneLoads, cuCycles / waveLoads, laneLoads * 16 / ms / 1e6);
    }
    return 0;
}
