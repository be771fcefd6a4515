// Measurement only: the cadence of DEPENDENT v_add_f32 on gfx950 (one wave, a chain on one register), with the addends
// in registers and with the addends read from LDS 16 ds_read_b128 at a time (the embedding update's fold loop).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/addchain.hip -o tools/ubench/addchain && tools/ubench/addchain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain_reg(float *out, int iters, long long *cyc) {
    float acc = out[threadIdx.x];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = out[64 + threadIdx.x + i];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = v[i] + acc;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain_lds(float *out, int iters, long long *cyc) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 68];
    for (int i = threadIdx.x; i < 64 * 68; i += 64) buf[i] = out[i % 128];
    __syncthreads();
    float acc = out[threadIdx.x];
    const float *row = buf + (threadIdx.x & 15) * 68;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4 *>(row + 4 * k);
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc = v[k].x + acc; acc = v[k].y + acc; acc = v[k].z + acc; acc = v[k].w + acc; }
        asm volatile("" : "+v"(acc) :: "memory");
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// two register sets: the next 32 entries' reads are issued before the current 32 adds
__global__ void k_chain_lds2(float *out, int iters, long long *cyc) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 68];
    for (int i = threadIdx.x; i < 64 * 68; i += 64) buf[i] = out[i % 128];
    __syncthreads();
    float acc = out[threadIdx.x];
    const float *row = buf + (threadIdx.x & 15) * 68;
    float4 va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) va[k] = *reinterpret_cast<const float4 *>(row + 4 * k);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) vb[k] = *reinterpret_cast<const float4 *>(row + 32 + 4 * k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc = va[k].x + acc; acc = va[k].y + acc; acc = va[k].z + acc; acc = va[k].w + acc; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) va[k] = *reinterpret_cast<const float4 *>(row + 4 * k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc = vb[k].x + acc; acc = vb[k].y + acc; acc = vb[k].z + acc; acc = vb[k].w + acc; }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// one read re-issued right after the four adds that consumed its register: the reads trail the adds by one 64-entry lap
__global__ void k_chain_lds3(float *out, int iters, long long *cyc) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 68];
    for (int i = threadIdx.x; i < 64 * 68; i += 64) buf[i] = out[i % 128];
    __syncthreads();
    float acc = out[threadIdx.x];
    const float *row = buf + (threadIdx.x & 15) * 68;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4 *>(row + 4 * k);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc = v[k].x + acc; acc = v[k].y + acc; acc = v[k].z + acc; acc = v[k].w + acc;
            __builtin_amdgcn_sched_barrier(0);
            v[k] = *reinterpret_cast<const float4 *>(row + 4 * k);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *d; long long *c, h;
    hipMalloc(&d, 4096 * 4); hipMalloc(&c, 8);
    hipMemset(d, 0, 4096 * 4);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_chain_reg, dim3(1), dim3(64), 0, 0, d, iters, c);
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) printf("registers: %.2f clock64 ticks per dependent v_add_f32 (64 per iteration)\n", (double)h / (64.0 * iters));
        hipLaunchKernelGGL(k_chain_lds, dim3(1), dim3(64), 0, 0, d, iters, c);
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) printf("LDS, 16 x ds_read_b128 then 64 adds: %.2f ticks per add\n", (double)h / (64.0 * iters));
        hipLaunchKernelGGL(k_chain_lds2, dim3(1), dim3(64), 0, 0, d, iters, c);
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) printf("LDS, two sets of 8 reads, the next set issued before the current 32 adds: %.2f ticks per add\n", (double)h / (64.0 * iters));
        hipLaunchKernelGGL(k_chain_lds3, dim3(1), dim3(64), 0, 0, d, iters, c);
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        if (rep) printf("LDS, each read re-issued after the 4 adds that used its register: %.2f ticks per add\n", (double)h / (64.0 * iters));
    }
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("(clock64 = shader clock; device clock rate attribute %d kHz)\n", khz);
    return 0;
}
