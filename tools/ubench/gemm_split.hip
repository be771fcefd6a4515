// Feasibility probe (round 4): the FC GEMM's f32 operands split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits) while
// they are staged into LDS, NP of the nine partial products on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the f32
// MFMA rate), f32 accumulators.  NP = 9: every product a*b is reproduced exactly (48 bits) before it is accumulated;
// NP = 6: the three smallest partial products (<= 2^-23 |a b|) are dropped.
//   C[m][n] = sum_k A[m][k] * Bt[n][k]       (the layout of k_gemm_nt: the forward and the data-gradient GEMMs)
// Times the kernel on fc_fwd0's shape (4096 x 512 x 429) and compares it and a plain f32 FMA kernel with a float64 sum.
//   hipcc --offload-arch=gfx950 -O3 -o gemm_split gemm_split.hip && ./gemm_split [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#include <utility>
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// a == hi + mid + lo exactly (truncating split; the pieces are bf16 values held in the high halves of f32 words)
__device__ __forceinline__ void split3(float a, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const uint32_t ua = __float_as_uint(a);
    hi = ua & 0xFFFF0000u;
    const float r = a - __uint_as_float(hi);
    const uint32_t ur = __float_as_uint(r);
    mid = ur & 0xFFFF0000u;
    lo = __float_as_uint(r - __uint_as_float(mid));       // <= 8 significant bits: a bf16 value already
}
__device__ __forceinline__ uint32_t pack_hi(uint32_t even, uint32_t odd) { return (even >> 16) | (odd & 0xFFFF0000u); }

constexpr int TM = 64, TN = 64, BK = 32;          // workgroup tile, K slab
constexpr int ROWB = BK * 2 + 16;                  // bytes of one row of a plane in LDS (32 bf16 + pad: conflict-free b128 reads)
constexpr int PLANE = 64 * ROWB;                   // one 64-row plane
// LDS: [buffer 2][operand 2][plane 3][64 rows][ROWB]
constexpr int BUF = 2 * 3 * PLANE;

template <int NP>
__global__ __launch_bounds__(256) void k_gemm_split(const float *__restrict__ A, int lda, const float *__restrict__ Bt, int ldb,
                                                    float *__restrict__ C, int ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = (N + TN - 1) / TN;
    const int m0 = (blockIdx.x / ntn) * TM, n0 = (blockIdx.x % ntn) * TN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // staging: thread t loads rows r = t / 4 (64 rows) and 8 consecutive k = (t % 4) * 8 of both operands
    const int sr = tid >> 2, sk = (tid & 3) * 8;
    const float *ga = A + (size_t)min(m0 + sr, M - 1) * lda + sk;
    const float *gb = Bt + (size_t)min(n0 + sr, N - 1) * ldb + sk;
    float ra[8], rb[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = k0 + sk + i < K;
            ra[i] = ok ? ga[k0 + i] : 0.f;
            rb[i] = ok ? gb[k0 + i] : 0.f;
        }
    };
    auto gload_fast = [&](int k0) {         // (rows are 16-byte aligned and the slab lies inside K)
        const float4 a0 = *reinterpret_cast<const float4 *>(ga + k0), a1 = *reinterpret_cast<const float4 *>(ga + k0 + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(gb + k0), b1 = *reinterpret_cast<const float4 *>(gb + k0 + 4);
        ra[0] = a0.x; ra[1] = a0.y; ra[2] = a0.z; ra[3] = a0.w; ra[4] = a1.x; ra[5] = a1.y; ra[6] = a1.z; ra[7] = a1.w;
        rb[0] = b0.x; rb[1] = b0.y; rb[2] = b0.z; rb[3] = b0.w; rb[4] = b1.x; rb[5] = b1.y; rb[6] = b1.z; rb[7] = b1.w;
    };
    auto sstore = [&](int buf) {
        unsigned char *base = lds + buf * BUF + sr * ROWB + sk * 2;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const float *r = op ? rb : ra;
            uint32_t h[8], m[8], l[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) split3(r[i], h[i], m[i], l[i]);
            u32x4 vh, vm, vl;
#pragma unroll
            for (int i = 0; i < 4; ++i) { vh[i] = pack_hi(h[2 * i], h[2 * i + 1]); vm[i] = pack_hi(m[2 * i], m[2 * i + 1]); vl[i] = pack_hi(l[2 * i], l[2 * i + 1]); }
            unsigned char *p = base + op * 3 * PLANE;
            *reinterpret_cast<u32x4 *>(p) = vh;
            *reinterpret_cast<u32x4 *>(p + PLANE) = vm;
            *reinterpret_cast<u32x4 *>(p + 2 * PLANE) = vl;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nslab = (K + BK - 1) / BK;
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0);
    if (aligned && BK <= K) gload_fast(0); else gload(0);
    sstore(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        if (s + 1 < nslab) { if (aligned && (s + 2) * BK <= K) gload_fast((s + 1) * BK); else gload((s + 1) * BK); }
        const unsigned char *pa = lds + cur * BUF + (wm + (lane & 31)) * ROWB + (lane >> 5) * 16;
        const unsigned char *pb = lds + cur * BUF + 3 * PLANE + (wn + (lane & 31)) * ROWB + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8 *>(pa + p * PLANE + ks * 32);
                b[p] = *reinterpret_cast<const bf16x8 *>(pb + p * PLANE + ks * 32);
            }
            // smallest partial products first
            if (NP >= 9) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
            }
            if (NP >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            }
            if (NP >= 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        if (s + 1 < nslab) sstore(cur ^ 1);
        __syncthreads();
    }
    // C fragment of 32x32 f32: acc[4 j + i] = C[8 j + 4 (lane / 32) + i][lane % 32]
    // (operand A of the MFMA = rows of C: the MFMA computes D[i][j] = sum_k A[i][k] B[k][j] with A given per row i, B per column j)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + wm + 8 * j + 4 * (lane >> 5) + i, c = n0 + wn + (lane & 31);
            if (r < M && c < N) C[(size_t)r * ldc + c] = acc[4 * j + i];
        }
}


// ---- v2: the general tile (WM x WN waves of TM x TN 32x32 tiles), loads two slabs ahead in two named register sets, unconditional
// clamped loads (no branches around memory operations: see kernels_gemm.hip), split + LDS store of slab s+1 behind the MFMAs of slab s
template <int WM, int WN, int TM_, int TN_, int BKT, int NP, int NSET, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64) void k_gemm_split2(const float *__restrict__ A, int lda, int a_rows, const float *__restrict__ Bt, int ldb, int b_rows,
                                                             float *__restrict__ C, int ldc, int M, int N, int K) {
    constexpr int NTH = WM * WN * 64, BM = WM * TM_ * 32, BN = WN * TN_ * 32;
    constexpr int RB = BKT * 2 + 16;                 // bytes per plane row
    constexpr int APL = BM * RB, BPL = BN * RB;      // one plane
    constexpr int BUFB = 3 * (APL + BPL);
    constexpr int CPR = BKT / 8;                     // 8-float chunks per row
    constexpr int A_CH = (BM * CPR + NTH - 1) / NTH, B_CH = (BN * CPR + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w / WN, wn = w % WN;
    const int ntn = (N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const float *pa[A_CH], *pb[B_CH];
    int ka[A_CH], kb[B_CH], oa[A_CH], ob[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int e = tid + i * NTH, r = min(e / CPR, BM - 1), c = e % CPR;
        pa[i] = A + (size_t)min(m0 + r, a_rows - 1) * lda; ka[i] = c * 8; oa[i] = r * RB + c * 16;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int e = tid + i * NTH, r = min(e / CPR, BN - 1), c = e % CPR;
        pb[i] = Bt + (size_t)min(n0 + r, b_rows - 1) * ldb; kb[i] = c * 8; ob[i] = 3 * APL + r * RB + c * 16;
    }
    float4 ra0[A_CH][2], rb0[B_CH][2], ra1[A_CH][2], rb1[B_CH][2];
    auto gload = [&](int kt, float4 (&ra)[A_CH][2], float4 (&rb)[B_CH][2]) {
        const int k0 = kt * BKT;
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int c = k0 + ka[i] + 4 * h; ra[i][h] = *reinterpret_cast<const float4 *>(pa[i] + (c < K ? c : K - 4)); }
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) { const int c = k0 + kb[i] + 4 * h; rb[i][h] = *reinterpret_cast<const float4 *>(pb[i] + (c < K ? c : K - 4)); }
    };
    auto store8 = [&](unsigned char *p, int plane_bytes, const float4 &v0, const float4 &v1, int c0) {
        const uint32_t m0_ = c0 < K ? 0xFFFFFFFFu : 0u, m1_ = c0 + 4 < K ? 0xFFFFFFFFu : 0u;
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint32_t h[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ABL & 1) { h[i] = __float_as_uint(x[i]); m[i] = h[i] + 1; l[i] = h[i] + 2; }
            else split3(__uint_as_float(__float_as_uint(x[i]) & (i < 4 ? m0_ : m1_)), h[i], m[i], l[i]);
        }
        u32x4 vh, vm, vl;
#pragma unroll
        for (int i = 0; i < 4; ++i) { vh[i] = pack_hi(h[2 * i], h[2 * i + 1]); vm[i] = pack_hi(m[2 * i], m[2 * i + 1]); vl[i] = pack_hi(l[2 * i], l[2 * i + 1]); }
        if (ABL & 2) { asm volatile("" ::"v"(vh[0]), "v"(vh[3]), "v"(vm[0]), "v"(vm[3]), "v"(vl[0]), "v"(vl[3])); return; }
        *reinterpret_cast<u32x4 *>(p) = vh;
        *reinterpret_cast<u32x4 *>(p + plane_bytes) = vm;
        *reinterpret_cast<u32x4 *>(p + 2 * plane_bytes) = vl;
    };
    auto sstore = [&](int buf, int kt, const float4 (&ra)[A_CH][2], const float4 (&rb)[B_CH][2]) {
        const int k0 = kt * BKT;
        unsigned char *base = lds + buf * BUFB;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) if ((BM * CPR) % NTH == 0 || tid + i * NTH < BM * CPR) store8(base + oa[i], APL, ra[i][0], ra[i][1], k0 + ka[i]);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) if ((BN * CPR) % NTH == 0 || tid + i * NTH < BN * CPR) store8(base + ob[i], BPL, rb[i][0], rb[i][1], k0 + kb[i]);
    };
    f32x16 acc[TM_][TN_];
#pragma unroll
    for (int i = 0; i < TM_; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fa_off = (wm * TM_ * 32 + (lane & 31)) * RB + (lane >> 5) * 16;
    const int fb_off = 3 * APL + (wn * TN_ * 32 + (lane & 31)) * RB + (lane >> 5) * 16;
    auto compute = [&](int buf) {
        const unsigned char *base = lds + buf * BUFB;
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            bf16x8 a[TM_][3], b[TN_][3];
#pragma unroll
            for (int i = 0; i < TM_; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (ABL & 4) { u32x4 t = {(uint32_t)lane, (uint32_t)p, (uint32_t)ks, (uint32_t)buf}; a[i][p] = __builtin_bit_cast(bf16x8, t); }
                    else a[i][p] = *reinterpret_cast<const bf16x8 *>(base + fa_off + i * 32 * RB + p * APL + ks * 32);
                }
#pragma unroll
            for (int j = 0; j < TN_; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (ABL & 4) { u32x4 t = {(uint32_t)lane, (uint32_t)p, (uint32_t)ks, (uint32_t)buf + 7}; b[j][p] = __builtin_bit_cast(bf16x8, t); }
                    else b[j][p] = *reinterpret_cast<const bf16x8 *>(base + fb_off + j * 32 * RB + p * BPL + ks * 32);
                }
            // partial products from the smallest to the largest; consecutive MFMAs go to different accumulators
            constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 9 - NP; q < 9; ++q)
#pragma unroll
                for (int i = 0; i < TM_; ++i)
#pragma unroll
                    for (int j = 0; j < TN_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = (K + BKT - 1) / BKT;
    if constexpr (NSET == 2) {
    gload(0, ra0, rb0);
    gload(1, ra1, rb1);
    sstore(0, 0, ra0, rb0);
    gload(2, ra0, rb0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        sstore(1, kt + 1, ra1, rb1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 16)) gload(kt + 3, ra1, rb1);
        if (!(ABL & 8)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        sstore(0, kt + 2, ra0, rb0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 16)) gload(kt + 4, ra0, rb0);
        if (!(ABL & 8)) __syncthreads();
    }
    if (kt < nk) compute(0);
    } else {
        // NSET register sets: the rows of slab s + NSET are requested while slab s is multiplied (a load takes ~2 us under load,
        // a slab a fraction of that).  Slabs past the end load clamped addresses and store zeros: no conditionals in the loop.
        float4 RA[NSET][A_CH][2], RB[NSET][B_CH][2];
        static_for<NSET>([&](auto sc) { constexpr int q = decltype(sc)::value; gload(q, RA[q], RB[q]); });
        sstore(0, 0, RA[0], RB[0]);
        gload(NSET, RA[0], RB[0]);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += NSET) {
            static_for<NSET>([&](auto sc) {
                constexpr int q = decltype(sc)::value, nx = (q + 1) % NSET;
                __builtin_amdgcn_sched_barrier(0);
                compute(q & 1);
                sstore((q + 1) & 1, kt + q + 1, RA[nx], RB[nx]);
                __builtin_amdgcn_sched_barrier(0);
                gload(kt + q + 1 + NSET, RA[nx], RB[nx]);
                __syncthreads();
            });
        }
    }
#pragma unroll
    for (int ti = 0; ti < TM_; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN_; ++tj)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = m0 + (wm * TM_ + ti) * 32 + 8 * j + 4 * (lane >> 5) + i, c = n0 + (wn * TN_ + tj) * 32 + (lane & 31);
                    if (r < M && c < N) C[(size_t)r * ldc + c] = acc[ti][tj][4 * j + i];
                }
}

template <int WM, int WN, int TM_, int TN_, int BKT, int NP, int NSET, int ABL = 0>
static float run2(const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K, int reps) {
    constexpr int BM = WM * TM_ * 32, BN = WN * TN_ * 32, RB = BKT * 2 + 16, BUFB = 3 * (BM + BN) * RB;
    const int nwg = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    auto kern = k_gemm_split2<WM, WN, TM_, TN_, BKT, NP, NSET, ABL>;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUFB));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), 2 * BUFB, 0, A, lda, M, Bt, ldb, N, C, ldc, M, N, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), 2 * BUFB, 0, A, lda, M, Bt, ldb, N, C, ldc, M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

__global__ void k_gemm_f32(const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K) {
    const int c = blockIdx.x * 16 + threadIdx.x, r = blockIdx.y * 16 + threadIdx.y;
    if (r >= M || c >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)r * lda + k], Bt[(size_t)c * ldb + k], s);
    C[(size_t)r * ldc + c] = s;
}

template <int NP>
static float run(const float *A, int lda, const float *Bt, int ldb, float *C, int ldc, int M, int N, int K, int reps) {
    const int nwg = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    CHECK(hipFuncSetAttribute((const void *)k_gemm_split<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_gemm_split<NP>, dim3(nwg), dim3(256), 2 * BUF, 0, A, lda, Bt, ldb, C, ldc, M, N, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm_split<NP>, dim3(nwg), dim3(256), 2 * BUF, 0, A, lda, Bt, ldb, C, ldc, M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

int main(int argc, char **argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 429;
    const int lda = (K + 3) / 4 * 4 + 4, ldb = lda, ldc = N;
    std::vector<float> hA((size_t)M * lda), hB((size_t)N * ldb);
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto &x : hA) x = rnd() * (rnd() > 0.f ? 1.f : 0.03f);       // mixed magnitudes
    for (auto &x : hB) x = rnd() * 0.1f;
    for (int r = 0; r < M; ++r) for (int k = K; k < lda; ++k) hA[(size_t)r * lda + k] = 0.f;
    for (int r = 0; r < N; ++r) for (int k = K; k < ldb; ++k) hB[(size_t)r * ldb + k] = 0.f;
    float *A, *B, *C, *C2;
    CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&B, hB.size() * 4)); CHECK(hipMalloc(&C, (size_t)M * ldc * 4)); CHECK(hipMalloc(&C2, (size_t)M * ldc * 4));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gemm_f32, dim3((N + 15) / 16, (M + 15) / 16), dim3(16, 16), 0, 0, A, lda, B, ldb, C2, ldc, M, N, K);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref32((size_t)M * ldc), out((size_t)M * ldc);
    CHECK(hipMemcpy(ref32.data(), C2, ref32.size() * 4, hipMemcpyDeviceToHost));
    // float64 truth on a sample of rows
    std::vector<int> rows;
    for (int r = 0; r < M; r += 97) rows.push_back(r);
    rows.push_back(M - 1);
    std::vector<double> truth(rows.size() * (size_t)N), scale(rows.size() * (size_t)N);
    for (size_t i = 0; i < rows.size(); ++i)
        for (int c = 0; c < N; ++c) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)rows[i] * lda + k] * (double)hB[(size_t)c * ldb + k]; s += p; sa += fabs(p); }
            truth[i * N + c] = s; scale[i * N + c] = sa;
        }
    auto report = [&](const char *name, const std::vector<float> &o) {
        double worst = 0, sum = 0;
        for (size_t i = 0; i < rows.size(); ++i)
            for (int c = 0; c < N; ++c) {
                const double e = fabs((double)o[(size_t)rows[i] * ldc + c] - truth[i * N + c]) / scale[i * N + c];      // relative to sum |a b|
                worst = e > worst ? e : worst; sum += e;
            }
        printf("%-28s error / sum|a b|: max %.3e  mean %.3e   (2^-24 = %.3e)\n", name, worst, sum / (rows.size() * (size_t)N), ldexp(1.0, -24));
    };
    report("plain f32 FMA chain", ref32);
    const double flops = 2.0 * M * N * K;
    float us;
    us = run<9>(A, lda, B, ldb, C, ldc, M, N, K, 200);
    CHECK(hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost));
    printf("bf16 x9: %.2f us  (%.1f TF/s algorithmic, %.0f TF/s issued)\n", us, flops / us * 1e-6, 9 * flops / us * 1e-6);
    report("bf16 x9", out);
    us = run<6>(A, lda, B, ldb, C, ldc, M, N, K, 200);
    CHECK(hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost));
    printf("bf16 x6: %.2f us  (%.1f TF/s algorithmic, %.0f TF/s issued)\n", us, flops / us * 1e-6, 6 * flops / us * 1e-6);
    report("bf16 x6", out);
    us = run<3>(A, lda, B, ldb, C, ldc, M, N, K, 200);
    CHECK(hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost));
    printf("bf16 x3: %.2f us\n", us);
    report("bf16 x3", out);
    us = run<1>(A, lda, B, ldb, C, ldc, M, N, K, 200);
    CHECK(hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost));
    printf("bf16 x1: %.2f us (the staging + one MFMA per step: what the loop costs without the extra products)\n", us);
    report("bf16 x1", out);

#define RUN2(WM, WN, TM_, TN_, BKT, NP, NSET, name) do { \
    CHECK(hipMemset(C, 0, (size_t)M * ldc * 4)); \
    us = run2<WM, WN, TM_, TN_, BKT, NP, NSET>(A, lda, B, ldb, C, ldc, M, N, K, 200); \
    CHECK(hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost)); \
    printf("v2 %-34s %.2f us (%.1f TF/s algorithmic)\n", name, us, flops / us * 1e-6); report(name, out); } while (0)
    RUN2(2, 2, 1, 1, 32, 9, 2, "64x64/32 x9 2 sets");
    RUN2(2, 2, 1, 1, 32, 1, 2, "64x64/32 x1 2 sets");
#define RUNA(ABL, name) do { us = run2<2, 2, 1, 1, 32, 1, 2, ABL>(A, lda, B, ldb, C, ldc, M, N, K, 200); printf("x1 ablation %-40s %.2f us\n", name, us); } while (0)
    RUNA(1, "no split VALU");
    RUNA(2, "no LDS writes");
    RUNA(4, "no LDS reads");
    RUNA(6, "no LDS reads or writes");
    RUNA(7, "no LDS, no split");
    RUNA(8, "no barriers");
    RUNA(16, "no global loads in the loop");
    RUNA(23, "no LDS, no split, no global loads");
    RUNA(31, "nothing but the loop + 2 MFMAs");
#define RUNB(ABL, name) do { us = run2<2, 2, 1, 1, 32, 9, 2, ABL>(A, lda, B, ldb, C, ldc, M, N, K, 200); printf("x9 ablation %-40s %.2f us\n", name, us); } while (0)
    RUNB(1, "no split VALU");
    RUNB(2, "no LDS writes");
    RUNB(4, "no LDS reads");
    RUNB(7, "no LDS, no split");
    RUNB(23, "no LDS, no split, no global loads");
    return 0;
}
