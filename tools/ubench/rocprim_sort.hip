// Measurement only (not linked into libps_amd.so): what rocPRIM's device radix sort (onesweep) takes for the multi-hot
// shape of the embedding backward -- 3.19 M (row key, bag) pairs, 22 key bits -- as a yardstick for kernels_sort.hip.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/rocprim_sort.hip -o tools/ubench/rocprim_sort && tools/ubench/rocprim_sort [n] [bits]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 3190000;
    const unsigned bits = argc > 2 ? (unsigned)atoi(argv[2]) : 22;
    std::vector<uint32_t> hk(n), hv(n);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hk[i] = (uint32_t)(x >> 20) & ((1u << bits) - 1); hv[i] = (uint32_t)i; }
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, 4 * n); hipMalloc(&k1, 4 * n); hipMalloc(&v0, 4 * n); hipMalloc(&v1, 4 * n);
    hipMemcpy(k0, hk.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), 4 * n, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    void *tmp; hipMalloc(&tmp, tmp_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    const int R = 50;
    hipEventRecord(e0, 0);
    for (int r = 0; r < R; ++r) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> ok(n);
    hipMemcpy(ok.data(), k1, 4 * n, hipMemcpyDeviceToHost);
    printf("rocprim radix_sort_pairs n=%zu bits=%u tmp=%zu B: %.1f us per sort, sorted=%d\n", n, bits, tmp_bytes, 1e3 * ms / R, (int)std::is_sorted(ok.begin(), ok.end()));
    return 0;
}
