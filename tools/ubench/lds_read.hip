// LDS read throughput of the MFMA fragment access patterns on gfx950: what does a ds_read_b128 / b64 / b32 of a wave cost
// when 32 lanes read 32 consecutive tile ROWS (row stride = pad) and the two half-waves read neighbouring k chunks?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_read.hip -o tools/ubench/lds_read && tools/ubench/lds_read
// One workgroup of 256 / 512 threads per CU; every wave issues `iters` x 8 reads back to back (waits only for the last).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int W>   // bytes per lane: 16, 8, 4
__global__ void k(unsigned long long *cyc, float *sink, int stride_b, int half_b, int iters, int mode) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr;
    if (mode == 0) addr = (lane & 31) * stride_b + (lane >> 5) * half_b;       // fragment pattern
    else if (mode == 1) addr = lane * W;                                       // linear (conflict-free by construction)
    else addr = ((lane & 31) * stride_b + (((lane >> 5) ^ ((lane >> 2) & 1)) * half_b));   // (unused variant)
    addr += (threadIdx.x >> 6) * 64;       // waves start at different places
    float4 acc = make_float4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (W == 16) { float4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr + q * 32)); acc.x += v.x; }
            else if (W == 8) { float2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr + q * 32)); acc.x += v.x; }
            else { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr + q * 32)); acc.x += v; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc.x == 12345.678f) sink[0] = acc.x;
}
template <int W>
void run(const char *name, int threads, int stride_b, int half_b, int mode) {
    const int grid = 256, iters = 2000;
    unsigned long long *d; float *s; hipMalloc(&d, 8 * grid); hipMalloc(&s, 4);
    hipLaunchKernelGGL(k<W>, dim3(grid), dim3(threads), 65536, 0, d, s, stride_b, half_b, iters, mode);
    hipLaunchKernelGGL(k<W>, dim3(grid), dim3(threads), 65536, 0, d, s, stride_b, half_b, iters, mode);
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    // s_memtime counts at 100 MHz on this part; report per-instruction time relative to the linear pattern instead of cycles
    const double per = avg / ((double)iters * 8 * (threads / 64));
    printf("%-8s %3d threads stride %4d B half %3d B mode %d: %10.0f ticks, %.4f ticks per wave-level read, %6.1f B/tick/CU\n", name, threads, stride_b, half_b, mode, avg, per,
           64.0 * W / per);
    hipFree(d); hipFree(s);
}
int main() {
    for (int threads : {256, 512}) {
        run<16>("b128", threads, 0, 0, 1);
        for (int st : {128, 144, 160, 176, 192, 208, 272}) run<16>("b128", threads, st, 16, 0);
        run<8>("b64", threads, 0, 0, 1);
        for (int st : {128, 136, 144, 152, 160}) run<8>("b64", threads, st, 8, 0);
        run<4>("b32", threads, 0, 0, 1);
        for (int st : {128, 132, 144}) run<4>("b32", threads, st, 4, 0);
    }
    return 0;
}
