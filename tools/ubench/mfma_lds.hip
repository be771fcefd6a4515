// What does a wave-level LDS read cost the matrix pipe on gfx950?  Every wave loops over { N independent LDS reads of the
// MFMA fragment pattern (never waited for inside the loop), 16 dependent f32 MFMAs }.  Cycles per iteration against N, the
// read width and the MFMA shape; 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int W, int N, int SHAPE, int GLOBAL, int WAIT = 0>   // WAIT 1: the reads are waited for BEHIND the iteration's MFMAs; 2: in front of them
// W bytes per lane per read; SHAPE 0: 32x32x2 (one accumulator), 1: 16x16x4 (four accumulators)
__global__ void k(unsigned long long *cyc, float *sink, const float *gsrc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (lane & 31) * 144 + (lane >> 5) * 16 + (threadIdx.x >> 6) * 4608;
    const float *gp = gsrc + (size_t)(blockIdx.x * 256 + (lane & 31)) * 128 + (lane >> 5) * 4;
    f32x16 acc; f32x4 a4[4];
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) a4[b][r] = 0.f;
    float x = (float)lane, y = 1.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            if (GLOBAL) { float4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(gp + (q & 7) * 8)); }
            else if (W == 16) { float4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr + (q & 7) * 32)); }
            else if (W == 8) { float2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr + (q & 7) * 32)); }
            else { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr + (q & 7) * 32)); }
        }
        if (WAIT == 2) asm volatile("s_waitcnt lgkmcnt(0)");
        if (SHAPE == 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 32; ++m) a4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[m & 3], 0, 0, 0);
        }
        if (WAIT == 1) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r]; for (int b = 0; b < 4; ++b) s += a4[b][0];
    if (s == 12345.678f) sink[0] = s;
}
template <int W, int N, int SHAPE, int GLOBAL, int WAIT = 0>
void run(const char *name, int threads) {
    const int grid = 256, iters = 500;
    unsigned long long *d; float *s, *g; hipMalloc(&d, 8 * grid); hipMalloc(&s, 4); hipMalloc(&g, (size_t)grid * 256 * 128 * 4 + 4096);
    hipMemset(g, 0, (size_t)grid * 256 * 128 * 4 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, N, SHAPE, GLOBAL, WAIT>), dim3(grid), dim3(threads), 65536, 0, d, s, g, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<W, N, SHAPE, GLOBAL, WAIT>), dim3(grid), dim3(threads), 65536, 0, d, s, g, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    printf("%-22s %3d threads, %2d reads of %2d B per 1024 MFMA cycles: %8.1f cycles per iteration (per wave on its SIMD: %6.1f); wall %.1f us -> %.2f ns per tick; %.1f TFLOP/s\n", name, threads, N, W, avg / iters,
           avg / iters / (threads / 256), ms * 1e3, ms * 1e6 / avg, 2.0 * 32 * 32 * 2 * 16 * iters * (threads / 64) * grid / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(s); hipFree(g);
}
int main() {
    for (int threads : {256, 512}) {
        run<16, 0, 0, 0>("32x32x2", threads);
        run<16, 4, 0, 0>("32x32x2 + b128", threads);
        run<16, 8, 0, 0>("32x32x2 + b128", threads);
        run<16, 16, 0, 0>("32x32x2 + b128", threads);
        run<8, 8, 0, 0>("32x32x2 + b64", threads);
        run<8, 16, 0, 0>("32x32x2 + b64", threads);
        run<4, 8, 0, 0>("32x32x2 + b32", threads);
        run<4, 32, 0, 0>("32x32x2 + b32", threads);
        run<16, 0, 1, 0>("16x16x4", threads);
        run<16, 8, 1, 0>("16x16x4 + b128", threads);
        run<16, 16, 1, 0>("16x16x4 + b128", threads);
        run<16, 8, 0, 0, 1>("32x32x2 + b128 waitB", threads);
        run<16, 2, 0, 0, 1>("32x32x2 + b128 waitB", threads);
        run<16, 8, 0, 0, 2>("32x32x2 + b128 waitF", threads);
        run<16, 8, 1, 0, 1>("16x16x4 + b128 waitB", threads);
        run<16, 8, 1, 0, 2>("16x16x4 + b128 waitF", threads);
        run<16, 8, 0, 1>("32x32x2 + global x4", threads);
        run<16, 16, 0, 1>("32x32x2 + global x4", threads);
    }
    return 0;
}
