#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
int main() {
    for (int n : {2048, 4096, 1 << 20}) for (int bits : {4, 15, 30}) {
        std::vector<uint32_t> k(n), v(n);
        uint32_t s = 1;
        int qlen = n - n / 5;
        for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; k[i] = i < qlen ? (s >> 2) : 0xFFFFFFFFu; v[i] = i < qlen ? 1000 + i : 0; }
        uint32_t *dk, *dko, *dv, *dvo; hipMalloc(&dk, n * 4); hipMalloc(&dko, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dvo, n * 4);
        hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
        size_t bytes = 0; hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dko, dv, dvo, n, 30 - bits, 32);
        void *tmp; hipMalloc(&tmp, bytes + 256);
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, dk, dko, dv, dvo, n, 30 - bits, 32, st);
        hipStreamSynchronize(st);
        std::vector<uint32_t> ov(n), ok(n); hipMemcpy(ov.data(), dvo, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ok.data(), dko, n * 4, hipMemcpyDeviceToHost);
        std::vector<uint32_t> a(ov.begin(), ov.begin() + qlen), b(v.begin(), v.begin() + qlen);
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        bool sorted = true; for (int i = 1; i < n; i++) if ((ok[i - 1] >> (30 - bits)) > (ok[i] >> (30 - bits))) sorted = false;
        printf("n %d bits %d: err %d temp %zu perm_ok %d sorted %d\n", n, bits, (int)e, bytes, (int)(a == b), (int)sorted);
    }
    return 0;
}
