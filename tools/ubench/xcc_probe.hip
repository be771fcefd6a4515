// Which XCD does workgroup b of a launch run on?  (MI355X: 8 XCDs; the GEMMs' tile order assumes b % 8, "observed".)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xcc_probe.hip -o tools/ubench/xcc_probe && tools/ubench/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);      // keep the workgroups resident for a while
}
int main() {
    for (int grid : {8, 64, 256, 512, 768, 2048, 8192}) {
        for (int threads : {256, 512}) {
            unsigned *d; hipMalloc(&d, 4 * grid);
            hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d, 200);
            std::vector<unsigned> h(grid); hipMemcpy(h.data(), d, 4 * grid, hipMemcpyDeviceToHost);
            int bad = 0; unsigned lo = ~0u, hi = 0;
            for (int b = 0; b < grid; ++b) { unsigned id = h[b] & 0xF; if ((int)id != b % 8) ++bad; lo = id < lo ? id : lo; hi = id > hi ? id : hi; }
            printf("grid %5d x %3d threads: XCC ids %u..%u, %d of %d workgroups NOT on XCD (blockIdx %% 8); first 16:", grid, threads, lo, hi, bad, grid);
            for (int b = 0; b < 16 && b < grid; ++b) printf(" %u", h[b] & 0xF);
            printf("\n");
            hipFree(d);
        }
    }
    return 0;
}
