// Row-panel-local FC forward chain as ONE launch, no cross-workgroup hand-over: a workgroup owns 16 batch rows, keeps
// x / h1 / h2 of those rows in LDS ([k/4][row][4] images, 1 KiB per 16-k step, conflict-free b128 reads) and streams the
// weights of BOTH layers from L2 in MFMA-fragment order (1 KiB contiguous per wave load, no LDS staging: the weights are
// the A operand of v_mfma_f32_16x16x4_f32, the 16 rows of activations the B operand, so a lane's four accumulator
// registers are four consecutive output features of one row = one ds_write_b128 into the next layer's image).
// Weight re-use per CU is 16 rows: (W0 + W1) x 256 workgroups = 353 MB from L2 per forward.  Question: what fraction of
// the f32 MFMA rate does that reach against k_gemm_nt's two launches (18.8 + 3 + 12.4 us in the step)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/rowpanel_mlp.hip -o /tmp/rowpanel_mlp && /tmp/rowpanel_mlp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int B = 4096, ROWS = 16;
constexpr int K0 = 416, K0P = 432, N0 = 512;   // layer 0: 416 embedding columns + ones column, padded to 27 steps of 16
constexpr int K1P = 528, N1 = 256;             // layer 1: 512 + ones column, padded to 33 steps of 16
constexpr int KS0 = K0P / 16, KS1 = K1P / 16;

// One layer for this wave: T tiles of 16 output features, KS steps of 16 k, D steps of weights in flight.
// wp: this wave's first tile in the packed weights [tile][ks][lane] of vf4; in: LDS image [k/4][row] of vf4.
template <bool NT>
__device__ __forceinline__ vf4 ldw(const vf4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
// MODE (timing only, wrong results): 1 weight loads wrap inside 8 KiB (L1 hits), 2 no weight loads in the loop, 3 no x staging
template <int MODE>
__device__ __forceinline__ size_t widx(size_t i) { return MODE == 1 ? (i & 511) : i; }
template <int KS, int T, int D, bool NT, int MODE>
__device__ __forceinline__ void layer_mm(const vf4 *__restrict__ wp, const vf4 *in, int lane, vf4 (&acc)[T], vf4 (&wr)[D][T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = vf4{0.f, 0.f, 0.f, 0.f};
    vf4 b = in[lane];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        vf4 bn = b;
        if (ks + 1 < KS) bn = in[(ks + 1) * 64 + lane];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const vf4 a = wr[ks % D][t];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[t], 0, 0, 0);
            if (ks + D < KS && MODE != 2) wr[ks % D][t] = ldw<NT>(wp + widx<MODE>(((size_t)t * KS + ks + D) * 64 + lane));
        }
        b = bn;
    }
}
template <int KS, int T, int D, bool NT, int MODE>
__device__ __forceinline__ void layer_prefetch(const vf4 *__restrict__ wp, int lane, vf4 (&wr)[D][T]) {
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int t = 0; t < T; ++t) wr[d][t] = ldw<NT>(wp + widx<MODE>(((size_t)t * KS + d) * 64 + lane));
}

template <int WAVES, int D, bool NT, bool WRITE_H1, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_rowpanel_fwd(const float *__restrict__ x, const vf4 *__restrict__ w0p,
                                                             const vf4 *__restrict__ w1p, float *__restrict__ h1g,
                                                             float *__restrict__ h2g) {
    constexpr int T0 = N0 / 16 / WAVES, T1 = N1 / 16 / WAVES;
    __shared__ vf4 xs[K0P / 4 * 16];
    __shared__ vf4 h1s[K1P / 4 * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * ROWS;
    vf4 wr0[D][T0];
    layer_prefetch<KS0, T0, D, NT, MODE>(w0p + (size_t)w * T0 * KS0 * 64, lane, wr0);
    // x panel -> LDS image: group g (4 floats of k) of row r at xs[g * 16 + r]
    if (MODE != 3) for (int i = tid; i < K0P / 4 * 16; i += WAVES * 64) {
        const int r = i / (K0P / 4), g = i % (K0P / 4);   // consecutive threads walk a row: coalesced global reads
        vf4 v = {0.f, 0.f, 0.f, 0.f};
        if (g < K0 / 4) v = *(const vf4 *)(x + (size_t)(row0 + r) * K0 + g * 4);
        else if (g == K0 / 4) v[0] = 1.f;
        xs[g * 16 + r] = v;
    }
    __syncthreads();
    vf4 acc0[T0];
    layer_mm<KS0, T0, D, NT, MODE>(w0p + (size_t)w * T0 * KS0 * 64, xs, lane, acc0, wr0);
    vf4 wr1[D][T1];
    layer_prefetch<KS1, T1, D, NT, MODE>(w1p + (size_t)w * T1 * KS1 * 64, lane, wr1);
    // epilogue 0: relu, into layer 1's image; lane (j = lane % 16 the row, q = lane / 16) holds features n0 + 4q .. 4q + 3
#pragma unroll
    for (int t = 0; t < T0; ++t) {
        vf4 v = acc0[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        const int g = (w * T0 + t) * 4 + (lane >> 4);
        h1s[g * 16 + (lane & 15)] = v;
        if (WRITE_H1) *(vf4 *)(h1g + (size_t)(row0 + (lane & 15)) * N0 + g * 4) = v;
    }
    if (tid < 64) {   // ones column + padding: groups 128 .. 131
        vf4 v = {0.f, 0.f, 0.f, 0.f};
        if ((tid >> 4) == 0) v[0] = 1.f;
        h1s[(N0 / 4 + (tid >> 4)) * 16 + (tid & 15)] = v;
    }
    __syncthreads();
    vf4 acc1[T1];
    layer_mm<KS1, T1, D, NT, MODE>(w1p + (size_t)w * T1 * KS1 * 64, h1s, lane, acc1, wr1);
#pragma unroll
    for (int t = 0; t < T1; ++t) {
        vf4 v = acc1[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        const int g = (w * T1 + t) * 4 + (lane >> 4);
        *(vf4 *)(h2g + (size_t)(row0 + (lane & 15)) * N1 + g * 4) = v;
    }
}

static void pack(const std::vector<float> &W, int N, int K, int KP, std::vector<float> &P) {
    const int KS = KP / 16;
    P.assign((size_t)N * KP, 0.f);
    for (int tile = 0; tile < N / 16; ++tile)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int m = 0; m < 4; ++m) {
                    const int n = tile * 16 + (lane & 15), k = ks * 16 + 4 * (lane >> 4) + m;
                    P[(((size_t)tile * KS + ks) * 64 + lane) * 4 + m] = k < K ? W[(size_t)n * K + k] : 0.f;
                }
}

template <int WAVES, int D, bool WRITE_H1, bool NT = false, int MODE = 0>
static void run(const char *name, const float *x, const vf4 *w0p, const vf4 *w1p, float *h1, float *h2, const std::vector<float> &ref, int reps) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHK(hipMemsetAsync(h2, 0, (size_t)B * N1 * 4, s));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_rowpanel_fwd<WAVES, D, NT, WRITE_H1, MODE>), dim3(B / ROWS), dim3(WAVES * 64), 0, s, x, w0p, w1p, h1, h2);
    CHK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rowpanel_fwd<WAVES, D, NT, WRITE_H1, MODE>), dim3(B / ROWS), dim3(WAVES * 64), 0, s, x, w0p, w1p, h1, h2);
    CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> out((size_t)B * N1);
    CHK(hipMemcpy(out.data(), h2, out.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < ref.size(); ++i) {   // ref holds rows 0..15 and 4080..4095
        const size_t row = i / N1 < 16 ? i / N1 : B - 32 + i / N1;
        const double d = fabs((double)out[row * N1 + i % N1] - ref[i]);
        if (d > maxerr) maxerr = d;
        if (fabs(ref[i]) > maxref) maxref = fabs(ref[i]);
    }
    const double us = ms * 1e3 / reps, flop = 2.0 * B * ((double)N0 * (K0 + 1) + (double)N1 * (N0 + 1));
    printf("%-34s %7.2f us per launch  %6.1f TFLOP/s  %.3f of 157.3   max|err| %.2e (max|ref| %.2f)\n", name, us, flop / us / 1e6,
           flop / us / 1e6 / 157.3, maxerr, maxref);
    CHK(hipStreamDestroy(s));
}

int main() {
    std::vector<float> X((size_t)B * K0), W0((size_t)N0 * (K0 + 1)), W1((size_t)N1 * (N0 + 1)), P0, P1;
    srand(1);
    for (auto &v : X) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto &v : W0) v = (rand() % 2001 - 1000) / 20000.f;
    for (auto &v : W1) v = (rand() % 2001 - 1000) / 20000.f;
    pack(W0, N0, K0 + 1, K0P, P0);
    pack(W1, N1, N0 + 1, K1P, P1);
    std::vector<float> ref((size_t)32 * N1);
    for (int rr = 0; rr < 32; ++rr) {
        const int row = rr < 16 ? rr : B - 32 + rr;
        std::vector<double> h1(N0 + 1);
        for (int n = 0; n < N0; ++n) {
            double s = W0[(size_t)n * (K0 + 1) + K0];
            for (int k = 0; k < K0; ++k) s += (double)W0[(size_t)n * (K0 + 1) + k] * X[(size_t)row * K0 + k];
            h1[n] = s > 0 ? s : 0;
        }
        h1[N0] = 1.0;
        for (int n = 0; n < N1; ++n) {
            double s = 0;
            for (int k = 0; k <= N0; ++k) s += (double)W1[(size_t)n * (N0 + 1) + k] * h1[k];
            ref[(size_t)rr * N1 + n] = (float)(s > 0 ? s : 0);
        }
    }
    float *x, *h1, *h2; vf4 *w0p, *w1p;
    CHK(hipMalloc(&x, X.size() * 4)); CHK(hipMalloc(&w0p, P0.size() * 4)); CHK(hipMalloc(&w1p, P1.size() * 4));
    CHK(hipMalloc(&h1, (size_t)B * N0 * 4)); CHK(hipMalloc(&h2, (size_t)B * N1 * 4));
    CHK(hipMemcpy(x, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(w0p, P0.data(), P0.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(w1p, P1.data(), P1.size() * 4, hipMemcpyHostToDevice));
    printf("row-panel forward chain, %d workgroups of %d rows: fwd0 %dx%dx%d + fwd1 %dx%dx%d, MFMA floor %.1f us\n", B / ROWS, ROWS, B, N0,
           K0 + 1, B, N1, N0 + 1, 2.0 * B * ((double)N0 * (K0 + 1) + (double)N1 * (N0 + 1)) / 157.3e6);
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 2, true>("8 waves, 2 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 3, true>("8 waves, 3 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, true>("8 waves, 4 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 6, true>("8 waves, 6 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, true, true>("8 waves, 4 steps ahead, nt loads", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, false>("8 waves, 4 steps ahead, h1 kept", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, true, false, 1>("8 waves, 4 ahead, loads hit L1 (wrong)", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, true, false, 2>("8 waves, 4 ahead, no loads (wrong)", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, true, false, 3>("8 waves, 4 ahead, no x staging (wrong)", x, w0p, w1p, h1, h2, ref, 200);
        run<8, 4, false, false, 2>("8 waves, no loads, h1 kept (wrong)", x, w0p, w1p, h1, h2, ref, 200);
        run<4, 2, true>("4 waves, 2 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
        run<4, 3, true>("4 waves, 3 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
        run<16, 4, true>("16 waves, 4 steps ahead, h1 stored", x, w0p, w1p, h1, h2, ref, 200);
    }
    return 0;
}
