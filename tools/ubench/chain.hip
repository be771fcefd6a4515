// micro-benchmark: cost of a strict dependent f32 add chain fed from LDS by one wave (the long-key fold of k_emb_reduce_update)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDE 260
template <int MODE>
__global__ void k(float *out, long long *cyc, int n) {
    __shared__ __attribute__((aligned(16))) float lds[16 * LDE];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * LDE; i += 64) lds[i] = 1.0f + 1e-7f * i;
    __syncthreads();
    float acc = 0.f;
    const long long t0 = clock64(); const long long w0 = wall_clock64();
    if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 3) {
        const float *row = lds + (lane & 15) * LDE;
        if (lane < 16) {
            for (int rep = 0; rep < n / 256; ++rep) {
                if (MODE == 0) {            // pure chain, operands in registers
                    float v = row[rep & 63];
#pragma unroll
                    for (int k = 0; k < 256; ++k) acc = v + acc;
                } else if (MODE == 1) {     // 8 x ds_read_b128 then 32 adds
                    for (int j = 0; j < 256; j += 32) {
                        float4 v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(row + j + 4 * k);
#pragma unroll
                        for (int k = 0; k < 8; ++k) { acc = v[k].x + acc; acc = v[k].y + acc; acc = v[k].z + acc; acc = v[k].w + acc; }
                    }
                } else if (MODE == 2) {     // 32 x ds_read_b32 then 32 adds
                    for (int j = 0; j < 256; j += 32) {
                        float v[32];
#pragma unroll
                        for (int k = 0; k < 32; ++k) v[k] = row[j + k];
#pragma unroll
                        for (int k = 0; k < 32; ++k) acc = v[k] + acc;
                    }
                } else {                    // 16 x ds_read_b64 then 32 adds
                    for (int j = 0; j < 256; j += 32) {
                        float2 v[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float2 *>(row + j + 2 * k);
#pragma unroll
                        for (int k = 0; k < 16; ++k) { acc = v[k].x + acc; acc = v[k].y + acc; }
                    }
                }
            }
        }
    } else if (MODE == 4) {
        // all 64 lanes read: lane = 4*d + q reads entries 16*i + 4*q .. +3 of component d; quad lane 0 folds after DPP broadcasts
        const int d = lane >> 2, q = lane & 3;
        const float *row = lds + d * LDE + 4 * q;
        for (int rep = 0; rep < n / 256; ++rep) {
            for (int j = 0; j < 256; j += 64) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(row + j + 16 * k);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float e[16];
                    e[0] = v[k].x; e[1] = v[k].y; e[2] = v[k].z; e[3] = v[k].w;
#pragma unroll
                    for (int qq = 1; qq < 4; ++qq) {
                        e[4 * qq + 0] = __shfl(v[k].x, qq, 4); e[4 * qq + 1] = __shfl(v[k].y, qq, 4);
                        e[4 * qq + 2] = __shfl(v[k].z, qq, 4); e[4 * qq + 3] = __shfl(v[k].w, qq, 4);
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc = e[t] + acc;
                }
            }
        }
    } else if (MODE == 5) {
        // same with explicit DPP quad_perm broadcasts (no LDS permute)
        const int d = lane >> 2, q = lane & 3;
        const float *row = lds + d * LDE + 4 * q;
        for (int rep = 0; rep < n / 256; ++rep) {
            for (int j = 0; j < 256; j += 64) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(row + j + 16 * k);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float c[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    float e[16];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int ci = __float_as_int(c[t]);
                        e[0 + t] = __int_as_float(__builtin_amdgcn_mov_dpp(ci, 0x00, 0xF, 0xF, true));    // quad_perm [0,0,0,0]
                        e[4 + t] = __int_as_float(__builtin_amdgcn_mov_dpp(ci, 0x55, 0xF, 0xF, true));    // [1,1,1,1]
                        e[8 + t] = __int_as_float(__builtin_amdgcn_mov_dpp(ci, 0xAA, 0xF, 0xF, true));    // [2,2,2,2]
                        e[12 + t] = __int_as_float(__builtin_amdgcn_mov_dpp(ci, 0xFF, 0xF, 0xF, true));   // [3,3,3,3]
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc = e[t] + acc;
                }
            }
        }
    }
    else if (MODE == 6) {
        // software pipelined: the next 8 x ds_read_b128 are in flight while the current 32 adds run
        const float *row = lds + (lane & 15) * LDE;
        if (lane < 16) {
            for (int rep = 0; rep < n / 256; ++rep) {
                float4 va[8], vb[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) va[k] = *reinterpret_cast<const float4 *>(row + 4 * k);
                for (int j = 0; j < 256; j += 64) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) vb[k] = *reinterpret_cast<const float4 *>(row + j + 32 + 4 * k);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc = va[k].x + acc; acc = va[k].y + acc; acc = va[k].z + acc; acc = va[k].w + acc; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) va[k] = *reinterpret_cast<const float4 *>(row + ((j + 64) & 255) + 4 * k);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc = vb[k].x + acc; acc = vb[k].y + acc; acc = vb[k].z + acc; acc = vb[k].w + acc; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else if (MODE == 7) {
        // mode 1 with all 64 lanes active (4 copies of the 16 rows): does the read cost depend on the active lanes?
        const float *row = lds + (lane & 15) * LDE;
        for (int rep = 0; rep < n / 256; ++rep) {
            for (int j = 0; j < 256; j += 32) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(row + j + 4 * k);
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc = v[k].x + acc; acc = v[k].y + acc; acc = v[k].z + acc; acc = v[k].w + acc; }
            }
        }
    }
    const long long t1 = clock64(); const long long w1 = wall_clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    out[lane] = acc;
}
int main() {
    float *o; long long *c, h[2];
    hipMalloc(&o, 256); hipMalloc(&c, 16);
    const int n = 65536;
    for (int mode = 0; mode < 8; ++mode) {
        for (int r = 0; r < 2; ++r) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, o, c, n);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, o, c, n);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        float ho[64]; hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
        printf("mode %d: %.2f ticks/add, %.2f ns/add (acc lane0 %.3f)\n", mode, (double)h[0] / n, 10.0 * (double)h[1] / n, ho[0]);
    }
    return 0;
}
