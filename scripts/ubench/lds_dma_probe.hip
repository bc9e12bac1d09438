// Probe: where does global_load_lds_dwordx4 (gfx950 LDS-DMA, 16 bytes per lane) put lane i's data, and what happens to the slots of lanes
// that exec masks off?  Prints the LDS slab after (a) a full-wave load, (b) a load with only the odd lanes active, (c) a load issued inside a
// divergent branch followed by more plain loads (the k_logic<LOGIC_MLP> pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(1))) const void *gptr;
typedef __attribute__((address_space(3))) void *lptr;

__global__ __launch_bounds__(256) void probe(const float4 *src, float4 *out, int mode)
{
    __shared__ float4 slab[4][256];
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    for (int r = 0; r < 4; r++) slab[r][tid] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    float4 *const base = &slab[0][tid & ~63];
    if (mode == 0) {
        __builtin_amdgcn_global_load_lds((gptr)(src + gid), (lptr)base, 16, 0, 0);
    } else if (mode == 1) {
        if (tid & 1) __builtin_amdgcn_global_load_lds((gptr)(src + gid), (lptr)base, 16, 0, 2);
    } else {
        if ((tid % 3) == 0) {
            __builtin_amdgcn_global_load_lds((gptr)(src + gid), (lptr)base, 16, 0, 2);
            __builtin_amdgcn_global_load_lds((gptr)(src + gid + 1024), (lptr)(base + 256), 16, 0, 2);
            __builtin_amdgcn_global_load_lds((gptr)(src + gid + 2048), (lptr)(base + 512), 16, 0, 2);
        }
    }
    float4 v = slab[0][tid], w = slab[1][tid], x = slab[2][tid];
    out[gid] = v; out[gid + 1024] = w; out[gid + 2048] = x;
}

int main()
{
    const int N = 4096;
    std::vector<float4> h(N);
    for (int i = 0; i < N; i++) h[i] = make_float4((float)i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *src, *out;
    CHECK(hipMalloc(&src, N * 16)); CHECK(hipMalloc(&out, N * 16));
    CHECK(hipMemcpy(src, h.data(), N * 16, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; mode++) {
        CHECK(hipMemset(out, 0, N * 16));
        hipLaunchKernelGGL(probe, dim3(4), dim3(256), 0, 0, src, out, mode);
        CHECK(hipDeviceSynchronize());
        std::vector<float4> o(N);
        CHECK(hipMemcpy(o.data(), out, N * 16, hipMemcpyDeviceToHost));
        int bad = 0, untouched_ok = 0;
        for (int g = 0; g < 1024; g++) {
            const bool active = mode == 0 || (mode == 1 && (g & 1)) || (mode == 2 && ((g & 255) % 3) == 0);
            for (int r = 0; r < (mode == 2 ? 3 : 1); r++) {
                const float4 v = o[g + 1024 * r]; const float4 e = h[g + 1024 * r];
                if (active) { if (v.x != e.x || v.y != e.y || v.z != e.z || v.w != e.w) { if (bad < 6) printf("mode %d gid %d rec %d: got %g %g %g %g want %g %g %g %g\n", mode, g, r, v.x, v.y, v.z, v.w, e.x, e.y, e.z, e.w); bad++; } }
                else if (v.x == -1.f && v.w == -1.f) untouched_ok++;
            }
        }
        printf("mode %d: %d wrong active slots, %d inactive slots untouched\n", mode, bad, untouched_ok);
    }
    return 0;
}
