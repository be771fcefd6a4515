// The wave's instruction stream of the GEMM slab loop, one group per iteration: { NW ds_write_b128, 2 ds_read_b128,
// s_waitcnt lgkmcnt(allow), 4 dependent 32x32x2 MFMAs (operands: constants or the fragments read one group earlier) }.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_lds2.hip -o /tmp/mfma_lds2 && /tmp/mfma_lds2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int NW, int NR, int DEP, int BAR>
__global__ void k(float *sink, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned raddr = (lane & 31) * 144 + (lane >> 5) * 16 + (w & 1) * 4608;
    const unsigned waddr = (lane >> 3) * 144 + (lane & 7) * 16 + 20000 + w * 1152;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    vf4 f0a = {1.f, 1.f, 1.f, 1.f}, f0b = f0a, f1a = f0a, f1b = f0a, wv = {1.f, 2.f, 3.f, (float)lane};
#define GROUP(RA, RB, UA, UB, OFF)                                                                                      \
    {                                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < NW; ++q) asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + q * 9216), "v"(wv));   \
        if (NR) {                                                                                                       \
            asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(RA) : "v"(raddr));                                   \
            asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(RB) : "v"(raddr + 9216));                            \
            if (NW == 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(UA), "+v"(UB));                                     \
            else if (NW == 1) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(UA), "+v"(UB));                                \
            else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(UA), "+v"(UB));                                             \
        }                                                                                                               \
        const vf4 xa = DEP ? UA : wv, xb = DEP ? UB : wv;                                                               \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[0], xb[0], acc, 0, 0, 0);                                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[1], xb[1], acc, 0, 0, 0);                                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2], xb[2], acc, 0, 0, 0);                                         \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[3], xb[3], acc, 0, 0, 0);                                         \
    }
    for (int it = 0; it < iters; it += 4) {
        GROUP(f1a, f1b, f0a, f0b, 0)
        GROUP(f0a, f0b, f1a, f1b, 32)
        GROUP(f1a, f1b, f0a, f0b, 64)
        GROUP(f0a, f0b, f1a, f1b, 96)
        if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0a), "+v"(f0b)); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) sink[0] = s;
}
template <int NW, int NR, int DEP, int BAR>
void run(const char *name, int threads) {
    const int grid = 256, iters = 4000;
    float *s; hipMalloc(&s, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NW, NR, DEP, BAR>), dim3(grid), dim3(threads), 65536, 0, s, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NW, NR, DEP, BAR>), dim3(grid), dim3(threads), 65536, 0, s, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %3d threads: %8.1f us  %6.1f TFLOP/s\n", name, threads, ms * 1e3, 2.0 * 32 * 32 * 2 * 4 * iters * (threads / 64) * grid / (ms * 1e-3) / 1e12);
    hipFree(s);
}
int main() {
    for (int threads : {256, 512}) {
        run<0, 0, 0, 0>("MFMA only", threads);
        run<0, 1, 1, 0>("2 reads, MFMAs use them", threads);
        run<1, 0, 0, 0>("1 write", threads);
        run<1, 1, 0, 0>("1 write, 2 reads (waited), MFMAs independent", threads);
        run<1, 1, 1, 0>("1 write, 2 reads, MFMAs use them", threads);
        run<1, 1, 1, 1>("1 write, 2 reads, MFMAs use them, barrier/4", threads);
        run<4, 1, 1, 0>("4 writes, 2 reads, MFMAs use them", threads);
    }
    return 0;
}
