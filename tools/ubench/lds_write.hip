// LDS write throughput of the GEMM staging patterns on gfx950 (ds_write_b128 / b64 / b32 of a wave): lane l writes chunk
// l % CPR of row l / CPR (row stride = pad bytes), i.e. CPR lanes fill one tile row contiguously.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_write.hip -o /tmp/lds_write && /tmp/lds_write
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
template <int W>
__global__ void k(unsigned long long *cyc, int stride_b, int cpr, int iters, int mode) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    unsigned addr;
    if (mode == 0) addr = (lane / cpr) * stride_b + (lane % cpr) * W;
    else addr = lane * W;                                   // linear
    addr += (threadIdx.x >> 6) * 8192;
    vf4 v = {1.f, 2.f, 3.f, (float)lane};
    vf2 v2 = {1.f, (float)lane};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (W == 16) asm volatile("ds_write_b128 %0, %1" ::"v"(addr + (q & 3) * 1152), "v"(v));
            else if (W == 8) asm volatile("ds_write_b64 %0, %1" ::"v"(addr + (q & 3) * 1152), "v"(v2));
            else asm volatile("ds_write_b32 %0, %1" ::"v"(addr + (q & 3) * 1152), "v"(v.x));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) cyc[blockIdx.x] = 0;
}
template <int W>
void run(const char *name, int threads, int stride_b, int cpr, int mode) {
    const int grid = 256, iters = 2000;
    unsigned long long *d; hipMalloc(&d, 8 * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(grid), dim3(threads), 65536, 0, d, stride_b, cpr, iters, mode);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<W>, dim3(grid), dim3(threads), 65536, 0, d, stride_b, cpr, iters, mode);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;                  // at 2.4 GHz
    const double per = cyc / ((double)iters * 8 * (threads / 64));
    printf("%-6s %3d threads stride %4d B, %2d lanes per row, mode %d: %8.1f us, %.2f cycles (2.4 GHz) per wave-level write, %6.1f B/clk/CU\n", name, threads, stride_b, cpr, mode,
           ms * 1e3, per, 64.0 * W / per);
    hipFree(d);
}
int main() {
    for (int threads : {256, 512}) {
        run<16>("b128", threads, 0, 0, 1);
        for (int st : {128, 144, 160, 176, 272}) run<16>("b128", threads, st, 8, 0);
        run<16>("b128", threads, 80, 4, 0);     // 16-float slabs (stride 20 floats)
        run<16>("b128", threads, 144, 1, 0);    // one lane per row (the READ pattern, as a write)
        run<16>("b128", threads, 144, 2, 0);
        run<16>("b128", threads, 144, 4, 0);
        run<8>("b64", threads, 0, 0, 1);
        run<8>("b64", threads, 144, 16, 0);
        run<4>("b32", threads, 0, 0, 1);
        run<4>("b32", threads, 144, 32, 0);
    }
    return 0;
}
