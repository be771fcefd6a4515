// kernels_gemm.hip -- FcLayer forward/backward contractions on the gfx950
// matrix cores with exact-f32 MFMA (v_mfma_f32_32x32x2_f32: an fmaf chain in
// k, one rounding per product; no TF32 exists on CDNA4).
//
//   FcLayer.forward   layer/FcLayer.java:76-77   Z = W*A + b      -> gemm_nt (bias folded in as
//                                                                   a ones column of A, see ps_model)
//   FcLayer.backward  layer/FcLayer.java:108     delta = W^T*d    -> gemm_nt (+ relu' mask epilogue)
//   FcLayer.backward  layer/FcLayer.java:103-105 dW = d*A^T / B,
//                                                db = rowMeans(d) -> gemm_tn_splitk (split over the batch;
//                                                                   the reducer is the dense updater)
//
// Tiling is for 64-wide waves: a workgroup is 4 waves in a WM x WN grid, each
// wave owns TM x TN accumulator tiles of 32x32 (16 acc VGPRs each).  K is
// walked in BK=16 slabs staged through double-buffered LDS with a register
// prefetch of the next slab, one barrier per slab.  At the f32 MFMA rate
// (64 cycles per instruction per SIMD) LDS bandwidth is far from binding, so
// the fragment reads stay simple: ds_read_b128 along k with a k-permutation
// (lanes 0-31 take k 0-3 / 8-11, lanes 32-63 take k 4-7 / 12-15 of a slab).
#include "ps_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int LDS_LD = 20;  // 16 + 4 pad floats: 16-B aligned rows, conflict-free ds_read_b128

struct NtArgs {
    const float *A; int lda; int a_rows;
    const float *Bt; int ldb; int b_rows;
    float *C; int ldc;
    int M, N, K;
    int epi;
    const float *mask; int ldmask; int mask_cols;
    const int *skip;
};

__device__ __forceinline__ float4 ld4_rows(const float *p, int ld, int row, int nrows, int col) {
    if (row < nrows) return *reinterpret_cast<const float4 *>(p + (size_t)row * ld + col);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float sigmoid_clip_dev(float x) {
    // activations/Sigmoid.java:11 -- float constants, double exp, cast to float
    return (float)(0.001f + (double)(.999f - 0.001f) / (1.0 + exp(-(double)x)));
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm_nt(NtArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = BM * 4 / 256 > 0 ? BM * 4 / 256 : 1;  // float4 per thread per slab
    constexpr int B_F4 = BN * 4 / 256 > 0 ? BN * 4 / 256 : 1;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_LD];
    if (a.skip && *a.skip) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nk = a.K / BK;

    float4 ra[A_F4], rb[B_F4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * 256;
            if (BM * 4 >= 256 || e < BM * 4)
                ra[i] = ld4_rows(a.A, a.lda, m0 + (e >> 2), a.a_rows, k0 + (e & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * 256;
            if (BN * 4 >= 256 || e < BN * 4)
                rb[i] = ld4_rows(a.Bt, a.ldb, n0 + (e >> 2), a.b_rows, k0 + (e & 3) * 4);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * 256;
            if (BM * 4 >= 256 || e < BM * 4)
                *reinterpret_cast<float4 *>(&As[buf][(e >> 2) * LDS_LD + (e & 3) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * 256;
            if (BN * 4 >= 256 || e < BN * 4)
                *reinterpret_cast<float4 *>(&Bs[buf][(e >> 2) * LDS_LD + (e & 3) * 4]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    swrite(0);
    __syncthreads();
    const int arow = (wm * TM * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int brow = (wn * TN * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const float4 *>(&As[buf][arow + i * 32 * LDS_LD + q * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const float4 *>(&Bs[buf][brow + j * 32 * LDS_LD + q * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

    // epilogue: acc[r] -> row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < a.M && col < a.N) {
                    float v = acc[i][j][r];
                    if (a.epi == EPI_RELU) v = v > 0.f ? v : 0.f;
                    else if (a.epi == EPI_SIGMOID) v = sigmoid_clip_dev(v);
                    else if (a.epi == EPI_MASK_POS) {
                        if (col < a.mask_cols) v *= a.mask[(size_t)row * a.ldmask + col] > 0.f ? 1.f : 0.f;
                    }
                    a.C[(size_t)row * a.ldc + col] = v;
                }
            }
        }
}

struct TnArgs {
    const float *A; int lda; int a_cols;
    const float *D; int ldd; int d_cols;
    float *Cpart; int ldc; long long part_stride;
    int Kout, N, M, mchunk;
    const int *skip;
};

// Cpart[z][kout][n] = sum_{m in split z} A[m][kout] * D[m][n]
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm_tn(TnArgs a) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = BM * 4 / 256 > 0 ? BM * 4 / 256 : 1;  // 16 rows * B?/4 float4 / 256 threads
    constexpr int B_F4 = BN * 4 / 256 > 0 ? BN * 4 / 256 : 1;
    __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
    __shared__ __attribute__((aligned(16))) float Ds[2][BK * BN];
    if (a.skip && *a.skip) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    const int k0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int m_begin = blockIdx.z * a.mchunk;
    const int m_end = m_begin + a.mchunk < a.M ? m_begin + a.mchunk : a.M;
    const int nk = m_end > m_begin ? (m_end - m_begin + BK - 1) / BK : 0;

    float4 ra[A_F4], rb[B_F4];
    auto gload = [&](int kt) {
        const int mb = m_begin + kt * BK;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * 256;
            const int r = e / (BM / 4), c = (e % (BM / 4)) * 4;
            const int gm = mb + r, gc = k0 + c;
            ra[i] = (e < BM * 4 && gm < m_end && gc < a.a_cols) ? *reinterpret_cast<const float4 *>(a.A + (size_t)gm * a.lda + gc)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * 256;
            const int r = e / (BN / 4), c = (e % (BN / 4)) * 4;
            const int gm = mb + r, gc = n0 + c;
            rb[i] = (e < BN * 4 && gm < m_end && gc < a.d_cols) ? *reinterpret_cast<const float4 *>(a.D + (size_t)gm * a.ldd + gc)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * 256;
            if (e < BM * 4) *reinterpret_cast<float4 *>(&As[buf][(e / (BM / 4)) * BM + (e % (BM / 4)) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * 256;
            if (e < BN * 4) *reinterpret_cast<float4 *>(&Ds[buf][(e / (BN / 4)) * BN + (e % (BN / 4)) * 4]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        gload(0);
        swrite(0);
    }
    __syncthreads();
    const int acol = wm * TM * 32 + (lane & 31);
    const int bcol = wn * TN * 32 + (lane & 31);
    const int kh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = As[buf][(2 * s + kh) * BM + acol + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = Ds[buf][(2 * s + kh) * BN + bcol + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }
    float *Cz = a.Cpart + (size_t)blockIdx.z * a.part_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const int rbase = k0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < a.Kout && col < a.N) Cz[(size_t)row * a.ldc + col] = acc[i][j][r];
            }
        }
}

}  // namespace

int gemm_nt(const float *A, int lda, int a_rows, const float *Bt, int ldb, int b_rows, float *C,
            int ldc, int M, int N, int K, int epi, const float *mask, int ldmask, int mask_cols,
            const int *skip_flag, hipStream_t st) {
    if (K % BK != 0 || (lda & 3) || (ldb & 3))
        return ps_set_err(PS_E_BAD_ARG, "gemm_nt: K=%d lda=%d ldb=%d must be multiples of 16/4/4", K, lda, ldb);
    if (M <= 0 || N <= 0) return PS_OK;
    NtArgs a{A, lda, a_rows, Bt, ldb, b_rows, C, ldc, M, N, K, epi, mask, ldmask, mask_cols, skip_flag};
    auto tiles = [&](int bm, int bn) { return (long long)cdiv(M, bm) * cdiv(N, bn); };
    if (N <= 32) {
        hipLaunchKernelGGL((k_gemm_nt<4, 1, 1, 1>), dim3(cdiv(M, 128), cdiv(N, 32)), dim3(256), 0, st, a);
    } else if (tiles(128, 128) >= 256) {
        hipLaunchKernelGGL((k_gemm_nt<2, 2, 2, 2>), dim3(cdiv(M, 128), cdiv(N, 128)), dim3(256), 0, st, a);
    } else if (tiles(64, 128) >= 256) {
        hipLaunchKernelGGL((k_gemm_nt<2, 2, 1, 2>), dim3(cdiv(M, 64), cdiv(N, 128)), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_gemm_nt<2, 2, 1, 1>), dim3(cdiv(M, 64), cdiv(N, 64)), dim3(256), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int gemm_tn_choose_split(int Kout, int N, int M) {
    const long long tiles = N <= 32 ? (long long)cdiv(Kout, 128) * cdiv(N, 32) : (long long)cdiv(Kout, 64) * cdiv(N, 64);
    int s = (int)((384 + tiles - 1) / tiles);          // aim at ~1.5 workgroups per CU
    const int max_s = M / 128 > 0 ? M / 128 : 1;       // keep >= 128 batch rows per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

int gemm_tn_splitk(const float *A, int lda, int a_cols, const float *D, int ldd, int d_cols,
                   float *Cpart, int ldc, int64_t part_stride, int Kout, int N, int M, int nsplit,
                   const int *skip_flag, hipStream_t st) {
    if ((lda & 3) || (ldd & 3) || (a_cols & 3) || (d_cols & 3))
        return ps_set_err(PS_E_BAD_ARG, "gemm_tn: leading dims / cols must be multiples of 4");
    if (Kout <= 0 || N <= 0 || nsplit <= 0) return PS_OK;
    const int mchunk = (int)round_up(cdiv(M > 0 ? M : 1, nsplit), BK);
    TnArgs a{A, lda, a_cols, D, ldd, d_cols, Cpart, ldc, (long long)part_stride, Kout, N, M, mchunk, skip_flag};
    if (N <= 32)
        hipLaunchKernelGGL((k_gemm_tn<4, 1, 1, 1>), dim3(cdiv(Kout, 128), cdiv(N, 32), nsplit), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((k_gemm_tn<2, 2, 1, 1>), dim3(cdiv(Kout, 64), cdiv(N, 64), nsplit), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}
