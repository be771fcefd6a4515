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
// walked in BKT-wide slabs staged through double-buffered LDS with a register
// prefetch of the next slab, one barrier per slab.  At the f32 MFMA rate
// (64 cycles per instruction per SIMD) LDS bandwidth is far from binding, so
// the fragment reads stay simple: ds_read_b128 along k with a k-permutation
// (per 8 k's: lanes 0-31 take k 0-3, lanes 32-63 take k 4-7).
#include "ps_common.h"
#include "kernels_emb.h"     // (FwdPanelArgs: the product build's stubs of the lab-only row-panel forward)
#include <string.h>
#include <strings.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct NtArgs {
    const float *A; int lda; int a_rows;
    const float *Bt; int ldb; int b_rows;
    float *C; int ldc;
    int M, N, K;
    int epi;
    const float *mask; int ldmask; int mask_cols;
    const int *skip;
    int xcd_swizzle;
    int ablate;      // measurement only: 1 = no global loads in the loop, 2 = no LDS writes, 4 = no barriers
    unsigned long long *ts;
    unsigned int *flag; unsigned int flag_val;      // "this launch has started" for a device-side waiter (launch_spin_until)
    const unsigned int *wait_flag; unsigned int wait_val;   // workgroup 0 ends only once *wait_flag has reached wait_val
    int prio;                                               // raise the waves' priority (a main-chain launch of the fused step)
    WaitBound bound;                                        // ... or gives up after bound.ticks and reports it in *bound.err
};

// XCD-aware work-group order.  MI355X dispatches block b to XCD b % 8 (observed, used for speed
// only), each XCD with its own 4 MiB L2.  The remap gives every XCD a CONTIGUOUS chunk of logical
// tile ids, so the tiles that share an operand panel (all N tiles of one M tile; all tiles of one
// batch split) run on one XCD and the panel is fetched from HBM once instead of once per XCD.
// Bijective for any block count (guide section 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_chunked_id(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float sigmoid_clip_dev(float x) {
    // activations/Sigmoid.java:11 -- float constants, double exp, cast to float
    return (float)(0.001f + (double)(.999f - 0.001f) / (1.0 + exp(-(double)x)));
}

// C[M][N] = epi(A[M][K] * Bt[N][K]^T); LDS rows are BKT+4 floats (16-B aligned,
// conflict-free ds_read_b128 for both BKT=16 (stride 20) and BKT=32 (stride 36)).
// KS: split of every K slab over KS groups of waves INSIDE the workgroup (KS = 2: 8 waves on a 64 x 64 tile, each group
// multiplies half of every slab's k range into its own accumulators; the groups' partial tiles are added through LDS
// before the epilogue).  The FC shapes give 1-2 workgroups of 4 waves per CU -- one or two waves per SIMD, nothing to
// hide a barrier or an LDS round trip behind; the in-workgroup split doubles the waves per SIMD at the same tile size,
// global traffic, LDS footprint and number of output tiles (no extra partial slabs, unlike a split over workgroups).
// One output tile of C = epi(A * Bt^T): everything of k_gemm_nt behind the choice of the tile (m0, n0).  As / Bs: the
// workgroup's double-buffered operand tiles.  Shared by k_gemm_nt and k_fc_fwd_pair.
// PS_GEMM_ABLATE (a MEASUREMENT build, tools/gemm_ablate_build.sh; results are garbage): what one part of a slab costs.
//   1 B fragments not read from LDS    2 no LDS reads at all    4 no MFMAs    8 no barriers in the loop
//   16 no global loads in the loop     32 no LDS writes in the loop
#ifndef PS_GEMM_ABLATE
#define PS_GEMM_ABLATE 0
#endif
// PS_GEMM_LAB (a MEASUREMENT build: tools/gemm_lab_build.sh -> ps_amd/lib/libps_amd_lab.so): every tile shape, slab loop and
// kernel that rounds 2-3 built, measured and did NOT make the default -- k_gemm_nt16 (16x16x4 MFMAs), k_fc_fwd_pair (two
// forward GEMMs in one launch), k_gemm_nt_lds (operands DMA'd to LDS), PIPE = 0 / 1 / 2 / 4, the in-workgroup K split of the
// NT kernel, 8-wave and 128-wide tiles, the pipelined dW GEMM.  The product library compiles only what gemm_nt_cfg = 0 /
// gemm_tn_cfg = 0 can reach; asking it for anything else is PS_E_UNSUPPORTED.  DESIGN.md 4.2 has the numbers.
#ifndef PS_GEMM_LAB
#define PS_GEMM_LAB 0
#endif
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// PIPE = 1: the slab loop software-pipelined INSIDE the wave (round 3; tools/gemm_ablate_build.sh priced the parts of a
// slab: MFMAs alone 96 % of the f32 MFMA rate, with the fragment reads from LDS 70 % -- hipcc issued each group's
// ds_read_b128 right in front of the MFMAs that use them, behind the previous group's dependent MFMA chain, so every group
// exposed an LDS round trip, and the LDS write of the next slab, the barrier and the first fragment read stood in a row at
// every slab boundary).  Now:
//   * the fragments of group g + 1 are read while group g's MFMAs run (two named fragment sets);
//   * THREE LDS buffers: slab kt + 2 is written during slab kt (its rows arrived in registers two slabs ago) -- the write's
//     latency hides behind MFMAs, and the first fragments of slab kt + 1 (written and made visible one barrier earlier) are
//     read BEFORE the barrier that ends slab kt: no LDS latency at the slab boundary at all;
//   * the masks / LDS writes / next global loads are dealt out one chunk per MFMA of the first group, so the VALU work sits
//     in the shadow of a running MFMA instead of between two groups.
// Same products in the same order per accumulator as PIPE = 0: bit-identical results.
template <int WM, int WN, int TM, int TN, int BKT, int KS, int PIPE = 0>
__device__ __forceinline__ void gemm_nt_tile(const NtArgs &a, const int m0, const int n0, float *const As, float *const Bs) {
    constexpr int LDB = (PIPE == 3 && BKT == 16 && KS == 1) ? 16 : BKT + 4;              // (row length: see SWZ below)
    constexpr int ASZ = WM * TM * 32 * LDB, BSZ = WN * TN * 32 * LDB;                    // floats per LDS buffer
    constexpr int NTH = WM * WN * 64 * KS;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // SWZ (16-wide slabs, the default loop): rows of 16 floats WITHOUT padding, the 16-byte chunk c of row r at position
    // c ^ ((r >> 2) & 3).  The padded rows (20 floats) are conflict-free for the fragment reads but not for the staging writes
    // (4 lanes fill one row: rows r and r + 3 of a 16-lane group overlap in 12 banks -- profiles/r03_gemm_pmc.txt: LDS bank
    // conflict cycles 0.7 of the busy cycles); with the XOR both are: 16 lanes of a write cover 4 whole rows = 64 banks, 16 lanes
    // of a read take the same chunk of 16 consecutive rows = 4 (r & 3) x 4 ((r >> 2) & 3) distinct bank groups.
    constexpr bool SWZ = PIPE == 3 && BKT == 16 && KS == 1;
    constexpr int LD = SWZ ? 16 : BKT + 4;
    auto wpos = [](int row, int chunk) -> int { return SWZ ? (chunk ^ ((row >> 2) & 3)) : chunk; };
    constexpr int RF4 = BKT / 4;                               // float4 per tile row
    constexpr int A_F4 = (BM * RF4 + NTH - 1) / NTH, B_F4 = (BN * RF4 + NTH - 1) / NTH;
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) % (WM * WN), kg = (tid >> 6) / (WM * WN);
    const int wm = w / WN, wn = w % WN;
    const int nk = (a.K + BKT - 1) / BKT;

    // TWO register sets: slab t+2 is already in flight while slab t is multiplied and slab t+1
    // waits in registers for its LDS slot -- one slab of prefetch does not cover the ~1.5 us a
    // load takes under load when a slab is only ~0.4 us of MFMA work (measured: 50 % of peak).
    // The sets are named (not indexed by t) so they stay in registers; the loop is unrolled by 2.
    float4 ra0[A_F4], rb0[B_F4], ra1[A_F4], rb1[B_F4];
    // Loads are UNCONDITIONAL and branch-free: rows beyond the operand are clamped to its last row (their products
    // land in output rows/columns the epilogue never stores), columns beyond K are clamped to the last float4 and
    // zeroed with a bit mask.  A select (ok ? v : 0) around the load was turned into exec-masked branches by
    // hipcc, and with branches between the loads its wait-count pass falls back to s_waitcnt vmcnt(0) before the LDS
    // write: the slab-(t+2) loads just issued were waited for as well, i.e. the second register set bought nothing
    // (measured: 42..53 % of the f32 MFMA peak on the FC shapes whatever the tile shape).
    const float *pa[A_F4], *pb[B_F4];
    int ca[A_F4], cb[B_F4], ar[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int e = tid + i * NTH;
        int r = m0 + (e / RF4 < BM ? e / RF4 : BM - 1);
        r = r < a.a_rows ? r : a.a_rows - 1;
        ca[i] = (e % RF4) * 4;
        pa[i] = a.A + (size_t)r * a.lda;
        ar[i] = r;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int e = tid + i * NTH;
        int r = n0 + (e / RF4 < BN ? e / RF4 : BN - 1);
        r = r < a.b_rows ? r : a.b_rows - 1;
        cb[i] = (e % RF4) * 4;
        pb[i] = a.Bt + (size_t)r * a.ldb;
    }
    // PS_GEMM_ABLATE & 512 (measurement): the A chunk of (row r, slab kt) comes from a pseudo-random 64-byte row of a big table
    // (a.mask, a.ldmask rows of 16 floats; half of the draws from a 4096-row hot set) -- what fusing the embedding gather into
    // the first FC GEMM's A-operand load would do to its loop
    auto emu_src = [&](int r, int kt2, int chunk) -> const float * {
        unsigned h = (unsigned)r * 2654435761u + (unsigned)kt2 * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        unsigned idx = h % (unsigned)a.ldmask;
        if (h & 0x10000u) idx &= 4095u;
        return a.mask + (size_t)idx * 16 + (chunk & 12);
    };
    // (the mask is applied when the registers go to LDS, a slab or two later: touching the loaded value in gload
    // would put the wait for it right behind the load)
    auto gload = [&](int kt, float4 (&ra)[A_F4], float4 (&rb)[B_F4]) {
        const int k0 = kt * BKT;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int c = k0 + ca[i];
            if (PS_GEMM_ABLATE & 512) ra[i] = *reinterpret_cast<const float4 *>(emu_src(ar[i], kt, ca[i]));
            else ra[i] = *reinterpret_cast<const float4 *>(pa[i] + (c < a.K ? c : a.K - 4));
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) { const int c = k0 + cb[i]; rb[i] = *reinterpret_cast<const float4 *>(pb[i] + (c < a.K ? c : a.K - 4)); }
    };
    auto masked = [&](float4 v, int c) -> float4 {
        if (PS_GEMM_ABLATE & 256) return make_float4(1.f, 2.f, 3.f, 4.f);      // (the LDS writes carry constants: no wait for the loads, no VALU)
        if (PS_GEMM_ABLATE & 128) return v;                                     // (no masking VALU)
        const int m = c < a.K ? -1 : 0;
        v.x = __int_as_float(__float_as_int(v.x) & m); v.y = __int_as_float(__float_as_int(v.y) & m);
        v.z = __int_as_float(__float_as_int(v.z) & m); v.w = __int_as_float(__float_as_int(v.w) & m);
        return v;
    };
    auto swrite = [&](int buf, int kt, const float4 (&ra)[A_F4], const float4 (&rb)[B_F4]) {
        const int k0 = kt * BKT;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * NTH;
            if ((BM * RF4) % NTH == 0 || e < BM * RF4)
                *reinterpret_cast<float4 *>(As + buf * ASZ + (e / RF4) * LD + wpos(e / RF4, e % RF4) * 4) = masked(ra[i], k0 + ca[i]);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * NTH;
            if ((BN * RF4) % NTH == 0 || e < BN * RF4)
                *reinterpret_cast<float4 *>(Bs + buf * BSZ + (e / RF4) * LD + wpos(e / RF4, e % RF4) * 4) = masked(rb[i], k0 + cb[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = (wm * TM * 32 + (lane & 31)) * LD + (SWZ ? 0 : (lane >> 5) * 4);
    const int brow = (wn * TN * 32 + (lane & 31)) * LD + (SWZ ? 0 : (lane >> 5) * 4);
    const int fxa = ((wm * TM * 32 + (lane & 31)) >> 2) & 3, fxb = ((wn * TN * 32 + (lane & 31)) >> 2) & 3, fh = lane >> 5;    // (SWZ)
    auto compute = [&](int buf) {
#pragma unroll
        for (int qq = 0; qq < BKT / 8 / KS; ++qq) {
            const int q = qq + kg * (BKT / 8 / KS);          // this wave group's part of the slab
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (PS_GEMM_ABLATE & 2) fa[i] = make_float4((float)q, (float)lane, 1.f, (float)buf);
                else fa[i] = *reinterpret_cast<const float4 *>(As + buf * ASZ + arow + i * 32 * LD + q * 8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (PS_GEMM_ABLATE & 3) fb[j] = fa[0];
                else fb[j] = *reinterpret_cast<const float4 *>(Bs + buf * BSZ + brow + j * 32 * LD + q * 8);
            }
            if (PS_GEMM_ABLATE & 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fa[i].x), "v"(fa[i].w), "v"(fb[j].x), "v"(fb[j].w));
                continue;
            }
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
    };
    // The loop body has NO conditional around a load or an LDS write: a slab index past the end loads from clamped
    // (valid) addresses and writes zeros to the LDS buffer nobody reads again.  With conditionals hipcc's wait-count
    // pass merges the two paths conservatively and waits for EVERY outstanding load before the next one is issued.
    if constexpr (PIPE == 0) {
        gload(0, ra0, rb0);
        gload(1, ra1, rb1);
        swrite(0, 0, ra0, rb0);
        __syncthreads();
        // sched_barrier: the loads of slab t+2 are issued BEFORE the MFMAs of slab t and the LDS write of slab t+1
        // comes AFTER them (left alone, hipcc's scheduler sinks the loads behind the LDS write: ~6 MFMAs of lookahead)
        int kt = 0;
        for (; kt + 2 <= nk; kt += 2) {
            // even slab kt: in LDS buffer 0; set 1 holds slab kt+1; set 0 is free for slab kt+2
            if (!(PS_GEMM_ABLATE & 16)) gload(kt + 2, ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(PS_GEMM_ABLATE & 32)) swrite(1, kt + 1, ra1, rb1);
            if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
            // odd slab kt+1: in LDS buffer 1; set 0 holds slab kt+2; set 1 is free for slab kt+3
            if (!(PS_GEMM_ABLATE & 16)) gload(kt + 3, ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            if (!(PS_GEMM_ABLATE & 32)) swrite(0, kt + 2, ra0, rb0);
            if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
        }
        if (kt < nk) compute(0);                                // odd slab count: the last slab sits in buffer 0
    } else {
        constexpr int NG = BKT / 8 / KS;                    // fragment groups per slab and wave
        constexpr int MF = TM * TN * 4;                     // MFMAs per group
        constexpr int NCH = A_F4 + B_F4;                    // operand chunks (float4) per thread and slab
        float4 FA[PIPE == 2 ? 4 : 2][TM], FB[PIPE == 2 ? 4 : 2][TN];          // fragment sets (indexed by compile-time constants only: registers)
        // chunk c of a slab: c < A_F4 -> A chunk c, else B chunk c - A_F4
        auto gload1 = [&](int kt2, float4 (&ra)[A_F4], float4 (&rb)[B_F4], int c) {
            const int k0 = kt2 * BKT;
            if (c < A_F4) {
                const int cc = k0 + ca[c];
                if (PS_GEMM_ABLATE & 512) ra[c] = *reinterpret_cast<const float4 *>(emu_src(ar[c], kt2, ca[c]));
                else ra[c] = *reinterpret_cast<const float4 *>(pa[c] + (cc < a.K ? cc : a.K - 4));
            }
            else { const int i = c - A_F4; const int cc = k0 + cb[i]; rb[i] = *reinterpret_cast<const float4 *>(pb[i] + (cc < a.K ? cc : a.K - 4)); }
        };
        auto swrite1 = [&](int oa, int ob, int kt2, const float4 (&ra)[A_F4], const float4 (&rb)[B_F4], int c) {
            const int k0 = kt2 * BKT;
            if (c < A_F4) {
                const int e = tid + c * NTH;
                if ((BM * RF4) % NTH == 0 || e < BM * RF4)
                    *reinterpret_cast<float4 *>(As + oa + (e / RF4) * LD + wpos(e / RF4, e % RF4) * 4) = masked(ra[c], k0 + ca[c]);
            } else {
                const int i = c - A_F4, e = tid + i * NTH;
                if ((BN * RF4) % NTH == 0 || e < BN * RF4)
                    *reinterpret_cast<float4 *>(Bs + ob + (e / RF4) * LD + wpos(e / RF4, e % RF4) * 4) = masked(rb[i], k0 + cb[i]);
            }
        };
        auto fread = [&](float4 (&fa)[TM], float4 (&fb)[TN], int oa, int ob, int g) {
            const int q = g + kg * NG;
            if (PS_GEMM_ABLATE & 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = make_float4((float)q, (float)lane, 1.f, (float)oa);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = fa[0];
                return;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4 *>(As + oa + arow + i * 32 * LD + (SWZ ? (((2 * q + fh) ^ fxa) << 2) : q * 8));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = (PS_GEMM_ABLATE & 1) ? fa[0] : *reinterpret_cast<const float4 *>(Bs + ob + brow + j * 32 * LD + (SWZ ? (((2 * q + fh) ^ fxb) << 2) : q * 8));
        };
        // MFMA m of a group: k component m / (TM * TN) of tile m % (TM * TN) -- consecutive MFMAs go to different
        // accumulators where there are several; per accumulator the k order is that of PIPE = 0
        auto fmfma = [&](const float4 (&fa)[TM], const float4 (&fb)[TN], auto mc) {
            constexpr int m = decltype(mc)::value, c = m / (TM * TN), t = m % (TM * TN), i = t / TN, j = t % TN;
            float x = c == 0 ? fa[i].x : c == 1 ? fa[i].y : c == 2 ? fa[i].z : fa[i].w;
            float y = c == 0 ? fb[j].x : c == 1 ? fb[j].y : c == 2 ? fb[j].z : fb[j].w;
            if (PS_GEMM_ABLATE & 64) {      // the reads happen, the MFMAs do not depend on them (consumed once per group, below)
                if (c == 3) asm volatile("" ::"v"(fa[i].x), "v"(fa[i].w), "v"(fb[j].x), "v"(fb[j].w));
                x = (float)lane; y = 1.f;
            }
            if (PS_GEMM_ABLATE & 4) { asm volatile("" ::"v"(x), "v"(y)); return; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
        };
        // One slab (in the LDS buffer at o_cur), first fragment set P0 already read.  FULL: the slab's part in the pipeline too
        // (slab kt + 2 from register set r to the buffer at o_wr, slab kt + 4 into r, first fragments of slab kt + 1 from o_nxt).
        // program order is the schedule: neither the IR passes (memory clobber) nor the machine scheduler (sched_barrier) may
        // move anything across
#define PS_ORDER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        // LOOK = PIPE: how many groups ahead the fragments are read (1: two fragment sets; 2: four -- the LDS queue behind the
        // writes of eight waves can be deeper than one group's 256 MFMA cycles)
        // PIPE = 3: THREE register sets of operand chunks in flight (the rows of slab kt + 5 are requested while slab kt is
        // multiplied: three slab times of load latency covered instead of two) and everything static: the loop is unrolled
        // by 3, register set and LDS buffer of a slab are both (slab % 3)
        // PIPE = 4: the pipeline inside a slab only, on TWO LDS buffers (37 KB per 64 x 64 workgroup instead of 55): slab kt + 1 is
        // written during slab kt, the first fragments of a slab are read behind the barrier (one exposed LDS round trip per slab)
        constexpr int LOOK = PIPE == 2 ? 2 : 1, NS = 2 * LOOK, GS = PIPE == 3 ? 3 : 2;
        constexpr bool CROSS = PIPE != 4;                   // fragments of the next slab are read across the barrier
        constexpr int WAHEAD = CROSS ? 2 : 1;               // the slab written while slab kt is multiplied
        static_assert(NG >= LOOK && (2 * NG) % NS == 0, "fragment sets must be back at set 0 after two slabs");
        auto slab = [&](auto p0c, auto fullc, float4 (&ra)[A_F4], float4 (&rb)[B_F4], int kt2, int oca, int ocb, int ona, int onb, int owa, int owb) {
            constexpr int P0 = decltype(p0c)::value;
            constexpr bool FULL = decltype(fullc)::value;
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value, P = (P0 + g) % NS, PN = (P0 + g + LOOK) % NS;
                PS_ORDER();
                if constexpr (!CROSS && g == 0) fread(FA[P], FB[P], oca, ocb, 0);
                if constexpr (g + LOOK < NG) fread(FA[PN], FB[PN], oca, ocb, g + LOOK);
                else if constexpr (FULL && CROSS) fread(FA[PN], FB[PN], ona, onb, g + LOOK - NG);
                PS_ORDER();
                static_for<MF>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    fmfma(FA[P], FB[P], mc);
                    if constexpr (FULL) {
                        // this MFMA's share of the chunks: dealt out over ALL MFMAs of the slab (LDS writes run at ~75 B/clk per CU,
                        // tools/ubench/lds_write.hip, a third of the read rate: the four waves' writes of a whole slab in one burst
                        // are 220 cycles of LDS time in front of the fragment reads queued behind them -- with the reads alone, or
                        // the writes alone, the loop runs at 94 % of the MFMA rate, with both in bursts at 74 %)
                        constexpr int mm = g * MF + m, T = NG * MF, c0 = (mm * NCH + T - 1) / T, c1 = ((mm + 1) * NCH + T - 1) / T;    // (behind the FIRST MFMA of a stretch)
                        if constexpr (c1 > c0) {
                            PS_ORDER();
                            static_for<c1 - c0>([&](auto cc) {
                                constexpr int c = c0 + decltype(cc)::value;
                                if (!(PS_GEMM_ABLATE & 32)) swrite1(owa, owb, kt2 + WAHEAD, ra, rb, c);
                                if (!(PS_GEMM_ABLATE & 16)) gload1(kt2 + WAHEAD + GS, ra, rb, c);
                            });
                            PS_ORDER();
                        }
                    }
                });
            });
            PS_ORDER();     // (the slab's last MFMAs are issued BEFORE the barrier: its wait for the prefetched fragments hides behind them)
        };
#undef PS_ORDER
        constexpr auto I0 = std::integral_constant<int, 0>{};
        constexpr auto I1 = std::integral_constant<int, NG % NS>{};
        constexpr auto YES = std::integral_constant<bool, true>{};
        constexpr auto NO = std::integral_constant<bool, false>{};
        if constexpr (PIPE == 4) {
            gload(0, ra0, rb0);
            gload(1, ra1, rb1);
            swrite(0, 0, ra0, rb0);
            gload(2, ra0, rb0);
            __syncthreads();
            int kt = 0;
            for (; kt + 2 <= nk; kt += 2) {
                slab(I0, YES, ra1, rb1, kt, 0, 0, 0, 0, ASZ, BSZ);
                __syncthreads();
                slab(I1, YES, ra0, rb0, kt + 1, ASZ, BSZ, 0, 0, 0, 0);
                __syncthreads();
            }
            if (kt < nk) slab(I0, NO, ra1, rb1, kt, 0, 0, 0, 0, ASZ, BSZ);
        } else if constexpr (PIPE == 3) {
            static_assert(NG % 2 == 0, "PIPE = 3: an even number of fragment groups per slab (the fragment sets start every slab at set 0)");
            float4 ra2[A_F4], rb2[B_F4];
            gload(0, ra0, rb0);
            gload(1, ra1, rb1);
            gload(2, ra2, rb2);
            swrite(0, 0, ra0, rb0);
            gload(3, ra0, rb0);
            swrite(1, 1, ra1, rb1);
            gload(4, ra1, rb1);
            __syncthreads();
            fread(FA[0], FB[0], 0, 0, 0);
            constexpr int A0 = 0, A1 = ASZ, A2 = 2 * ASZ, B0 = 0, B1 = BSZ, B2 = 2 * BSZ;
            int kt = 0;
            for (; kt + 3 <= nk; kt += 3) {
                slab(I0, YES, ra2, rb2, kt, A0, B0, A1, B1, A2, B2);
                if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
                slab(I0, YES, ra0, rb0, kt + 1, A1, B1, A2, B2, A0, B0);
                if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
                slab(I0, YES, ra1, rb1, kt + 2, A2, B2, A0, B0, A1, B1);
                if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
            }
            if (nk - kt == 2) {
                slab(I0, YES, ra2, rb2, kt, A0, B0, A1, B1, A2, B2);
                __syncthreads();
                slab(I0, NO, ra0, rb0, kt + 1, A1, B1, A2, B2, A0, B0);
            } else if (nk - kt == 1) {
                slab(I0, NO, ra2, rb2, kt, A0, B0, A1, B1, A2, B2);
            }
        } else {
            gload(0, ra0, rb0);
            gload(1, ra1, rb1);
            swrite(0, 0, ra0, rb0);
            gload(2, ra0, rb0);
            swrite(1, 1, ra1, rb1);
            gload(3, ra1, rb1);
            __syncthreads();
            fread(FA[0], FB[0], 0, 0, 0);
            if constexpr (LOOK == 2) fread(FA[1], FB[1], 0, 0, 1);
            int oca = 0, ocb = 0, ona = ASZ, onb = BSZ, owa = 2 * ASZ, owb = 2 * BSZ;
            auto rotate = [&]() { const int ta = oca, tb = ocb; oca = ona; ocb = onb; ona = owa; onb = owb; owa = ta; owb = tb; };
            int kt = 0;
            for (; kt + 2 <= nk; kt += 2) {
                slab(I0, YES, ra0, rb0, kt, oca, ocb, ona, onb, owa, owb);
                if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
                rotate();
                slab(I1, YES, ra1, rb1, kt + 1, oca, ocb, ona, onb, owa, owb);
                if (!(PS_GEMM_ABLATE & 8)) __syncthreads();
                rotate();
            }
            if (kt < nk) slab(I0, NO, ra0, rb0, kt, oca, ocb, ona, onb, owa, owb);    // odd slab count (the sets are back at parity 0)
        }
    }
    if (KS > 1) {
        // the wave groups' partial tiles: group 1 parks its accumulators in LDS (the operand buffers are free now: 16
        // floats per lane, lane-major per register so that both sides touch consecutive words), group 0 adds them
        __syncthreads();
        float *scr = As + w * 16 * 64;                      // (2 * BM * LD floats >= WM * WN * 1024)
        static_assert(KS == 1 || 2 * BM * LD >= WM * WN * 16 * 64, "reduction scratch does not fit the A buffers");
        if (kg == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[r * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (kg != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += scr[r * 64 + lane];
    }

    // epilogue: acc[r] -> row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
            // relu' mask: all 16 loads issued up front with clamped (always valid) addresses --
            // a load inside the per-element `if` would be waited for one by one
            float mk[16];
            if (a.epi == EPI_MASK_POS) {
                const int mc = col < a.mask_cols ? col : a.mask_cols - 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    mk[r] = a.mask[(size_t)(row < a.M ? row : a.M - 1) * a.ldmask + mc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r];
                if (a.epi == EPI_RELU) v = v > 0.f ? v : 0.f;
                else if (a.epi == EPI_SIGMOID) v = sigmoid_clip_dev(v);
                else if (a.epi == EPI_MASK_POS) v *= (col >= a.mask_cols || mk[r] > 0.f) ? 1.f : 0.f;
                if (row < a.M && col < a.N) a.C[(size_t)row * a.ldc + col] = v;
            }
        }
}

template <int WM, int WN, int TM, int TN, int BKT, int KS = 1, int PIPE = 0>
__global__ __launch_bounds__(WM * WN * 64 * KS) void k_gemm_nt(NtArgs a) {
    static_assert(KS == 1 || (KS == 2 && TM == 1 && TN == 1 && BKT % 16 == 0), "in-workgroup K split: 2 groups, one 32 x 32 tile per wave");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;       // (4 waves -- the default tiles -- or 8)
    constexpr int LD = (PIPE == 3 && BKT == 16 && KS == 1) ? 16 : BKT + 4;      // (16-wide slabs: swizzled rows without padding, gemm_nt_tile)
    __shared__ __attribute__((aligned(16))) float As[((PIPE && PIPE != 4) ? 3 : 2) * BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[((PIPE && PIPE != 4) ? 3 : 2) * BN * LD];
    // Main-chain kernel: its waves go ahead of the side chains' waves (field sort, dW GEMMs) wherever they share a CU.
    // HIP stream priorities changed nothing on this runtime; the wave priority does: with it fc_fwd1 (one workgroup per
    // CU, 26 of them beside a sort workgroup) takes 13.8 us instead of 16.3 and the last delta GEMM 25.2 instead of 29.5.
    // Every GEMM of the step (NT and TN) and the head carry it -- armed per launch by the model, single-hot steps only:
    // with the NT GEMMs alone 0.1495 ms/step, with all of them 0.1469 (the dW GEMMs no longer fall behind), and the
    // sharded step settles at 0.189-0.193.
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    EndWait end_wait(a.wait_flag, a.wait_val, a.bound);       // (declared first: runs after the stamp's end)
    StampScope stamp(a.ts);
    if (a.flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.skip && *a.skip) return;
    const int tn = (a.N + BN - 1) / BN;
    const int wg = a.xcd_swizzle ? xcd_chunked_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    gemm_nt_tile<WM, WN, TM, TN, BKT, KS, PIPE>(a, (wg / tn) * BM, (wg % tn) * BN, As, Bs);     // consecutive ids: the N tiles of one M tile
}

#if PS_GEMM_LAB
#include "lab/kernels_gemm_lab.inc"      // (the rejected variants of rounds 2-3: lab build only)
#endif      // PS_GEMM_LAB

struct TnArgs {
    const float *A; int lda; int a_cols;
    const float *D; int ldd; int d_cols;
    float *Cpart; int ldc; long long part_stride;
    int Kout, N, M, mchunk;
    const int *skip;
    int xcd_swizzle;
    unsigned long long *ts;
    int prio;        // raise the waves' priority (per launch like NtArgs.prio)
    const unsigned int *wait_flag; unsigned int wait_val; WaitBound bound;    // start wait (ps_common.h start_wait)
};

// Cpart[z][kout][n] = sum_{m in split z} A[m][kout] * D[m][n]
// KS: the batch rows of every slab split over KS groups of waves inside the workgroup (see k_gemm_nt)
// PIPE = 3: the slab loop software-pipelined like gemm_nt_tile<PIPE = 3> (fragments of the next four batch-row pairs read
// while the current four are multiplied, three LDS buffers and three register sets, one operand chunk's LDS write + next
// global load per stretch of MFMAs); same products in the same order per accumulator: bit-identical results.
template <int WM, int WN, int TM, int TN, int BKT, int KS = 1, int PIPE = 0>
__global__ __launch_bounds__(256 * KS) void k_gemm_tn(TnArgs a) {
    static_assert(WM * WN == 4, "four waves per K group");
    static_assert(KS == 1 || (KS == 2 && TM == 1 && TN == 1 && BKT % 4 == 0), "in-workgroup K split: 2 groups, one 32 x 32 tile per wave");
    constexpr int NTH = 256 * KS;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = (BKT * BM / 4 + NTH - 1) / NTH, B_F4 = (BKT * BN / 4 + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) float As[PIPE == 3 ? 3 : 2][BKT * BM];
    __shared__ __attribute__((aligned(16))) float Ds[PIPE == 3 ? 3 : 2][BKT * BN];
    // (see k_gemm_nt: every GEMM of the fused step ahead of the sort, the small kernels and the updates; prio 2 / 3: one or
    // two levels below the delta GEMMs of the main chain -- ps_tune_set("tn_prio"))
    if (a.prio == 1) __builtin_amdgcn_s_setprio(3);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(1);
    StampScope stamp(a.ts);
    start_wait(a.wait_flag, a.wait_val, a.bound);
    if (a.skip && *a.skip) return;
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, kg = tid >> 8;
    const int wm = w / WN, wn = w % WN;
    const int tk = (a.Kout + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
    const int wg = a.xcd_swizzle ? xcd_chunked_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int z = wg / (tk * tn), t = wg % (tk * tn);        // consecutive ids: all tiles of one batch split
    const int k0 = (t / tn) * BM, n0 = (t % tn) * BN;
    const int m_begin = z * a.mchunk;
    const int m_end = m_begin + a.mchunk < a.M ? m_begin + a.mchunk : a.M;
    const int nk = m_end > m_begin ? (m_end - m_begin + BKT - 1) / BKT : 0;

    float4 ra0[A_F4], rb0[B_F4], ra1[A_F4], rb1[B_F4];     // two sets: see k_gemm_nt
    // unconditional, branch-free loads (see k_gemm_nt): batch rows beyond this split are clamped and zeroed with a
    // bit mask (they are part of the SUM here, unlike the clamped rows of the NT kernel); columns beyond the operand
    // are clamped only (their outputs are never stored)
    int ra_r[A_F4], ra_c[A_F4], rb_r[B_F4], rb_c[B_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int e = tid + i * NTH < BKT * BM / 4 ? tid + i * NTH : BKT * BM / 4 - 1;
        ra_r[i] = e / (BM / 4);
        const int gc = k0 + (e % (BM / 4)) * 4;
        ra_c[i] = gc < a.a_cols ? gc : a.a_cols - 4;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
        const int e = tid + i * NTH < BKT * BN / 4 ? tid + i * NTH : BKT * BN / 4 - 1;
        rb_r[i] = e / (BN / 4);
        const int gc = n0 + (e % (BN / 4)) * 4;
        rb_c[i] = gc < a.d_cols ? gc : a.d_cols - 4;
    }
    auto ld4 = [&](const float *base, int ld, int gm, int c) -> float4 {
        const int mm = gm < a.M ? gm : a.M - 1;
        return *reinterpret_cast<const float4 *>(base + (size_t)mm * ld + c);
    };
    auto gload = [&](int kt, float4 (&ra)[A_F4], float4 (&rb)[B_F4]) {
        const int mb = m_begin + kt * BKT;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) ra[i] = ld4(a.A, a.lda, mb + ra_r[i], ra_c[i]);
#pragma unroll
        for (int i = 0; i < B_F4; ++i) rb[i] = ld4(a.D, a.ldd, mb + rb_r[i], rb_c[i]);
    };
    auto masked = [&](float4 v, int gm) -> float4 {           // applied on the way to LDS (see k_gemm_nt)
        const int m = gm < m_end ? -1 : 0;
        v.x = __int_as_float(__float_as_int(v.x) & m); v.y = __int_as_float(__float_as_int(v.y) & m);
        v.z = __int_as_float(__float_as_int(v.z) & m); v.w = __int_as_float(__float_as_int(v.w) & m);
        return v;
    };
    auto swrite = [&](int buf, int kt, const float4 (&ra)[A_F4], const float4 (&rb)[B_F4]) {
        const int mb = m_begin + kt * BKT;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * NTH;
            if (e < BKT * BM / 4) *reinterpret_cast<float4 *>(&As[buf][(e / (BM / 4)) * BM + (e % (BM / 4)) * 4]) = masked(ra[i], mb + ra_r[i]);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * NTH;
            if (e < BKT * BN / 4) *reinterpret_cast<float4 *>(&Ds[buf][(e / (BN / 4)) * BN + (e % (BN / 4)) * 4]) = masked(rb[i], mb + rb_r[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int acol = wm * TM * 32 + (lane & 31);
    const int bcol = wn * TN * 32 + (lane & 31);
    const int kh = lane >> 5;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ss = 0; ss < BKT / 2 / KS; ++ss) {
            const int s = ss + kg * (BKT / 2 / KS);           // this wave group's batch rows of the slab
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
    };
    if constexpr (PIPE == 0) {
        gload(0, ra0, rb0);                                     // (no conditionals around loads / LDS writes: see k_gemm_nt)
        gload(1, ra1, rb1);
        swrite(0, 0, ra0, rb0);
        __syncthreads();
        int kt = 0;
        for (; kt + 2 <= nk; kt += 2) {
            gload(kt + 2, ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            swrite(1, kt + 1, ra1, rb1);
            __syncthreads();
            gload(kt + 3, ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            swrite(0, kt + 2, ra0, rb0);
            __syncthreads();
        }
        if (kt < nk) compute(0);
    } else {
        constexpr int SPW = BKT / 2 / KS;                   // batch-row pairs (MFMA k steps) per wave and slab
        constexpr int GM = SPW % 8 == 0 ? 4 : SPW % 4 == 0 ? 2 : 1;      // ... per fragment group (an even number of groups per slab)
        constexpr int NG = SPW / GM, MF = GM * TM * TN, NCH = A_F4 + B_F4;
        static_assert(SPW % GM == 0 && NG % 2 == 0, "an even number of fragment groups per slab");
        float4 ra2[A_F4], rb2[B_F4];
        float FA[2][GM][TM], FB[2][GM][TN];
        auto gload1 = [&](int kt2, float4 (&ra)[A_F4], float4 (&rb)[B_F4], int c) {
            const int mb = m_begin + kt2 * BKT;
            if (c < A_F4) ra[c] = ld4(a.A, a.lda, mb + ra_r[c], ra_c[c]);
            else rb[c - A_F4] = ld4(a.D, a.ldd, mb + rb_r[c - A_F4], rb_c[c - A_F4]);
        };
        auto swrite1 = [&](int buf, int kt2, const float4 (&ra)[A_F4], const float4 (&rb)[B_F4], int c) {
            const int mb = m_begin + kt2 * BKT;
            if (c < A_F4) {
                const int e = tid + c * NTH;
                if (e < BKT * BM / 4) *reinterpret_cast<float4 *>(&As[buf][(e / (BM / 4)) * BM + (e % (BM / 4)) * 4]) = masked(ra[c], mb + ra_r[c]);
            } else {
                const int i = c - A_F4, e = tid + i * NTH;
                if (e < BKT * BN / 4) *reinterpret_cast<float4 *>(&Ds[buf][(e / (BN / 4)) * BN + (e % (BN / 4)) * 4]) = masked(rb[i], mb + rb_r[i]);
            }
        };
        auto fread = [&](float (&fa)[GM][TM], float (&fb)[GM][TN], int buf, int g) {
#pragma unroll
            for (int t = 0; t < GM; ++t) {
                const int sidx = g * GM + t + kg * SPW;
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[t][i] = As[buf][(2 * sidx + kh) * BM + acol + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[t][j] = Ds[buf][(2 * sidx + kh) * BN + bcol + j * 32];
            }
        };
        auto fmfma = [&](const float (&fa)[GM][TM], const float (&fb)[GM][TN], auto mc) {
            constexpr int m = decltype(mc)::value, t = m / (TM * TN), q = m % (TM * TN), i = q / TN, j = q % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][i], fb[t][j], acc[i][j], 0, 0, 0);
        };
#define PS_ORDER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        // one slab (LDS buffer cur); FULL: slab kt + 2 from register set r to buffer wr, slab kt + 5 into r, the first fragments of
        // slab kt + 1 from buffer nxt
        auto slab = [&](auto fullc, float4 (&ra)[A_F4], float4 (&rb)[B_F4], int kt2, auto curc, auto nxtc, auto wrc) {
            constexpr bool FULL = decltype(fullc)::value;
            constexpr int cur = decltype(curc)::value, nxt = decltype(nxtc)::value, wr = decltype(wrc)::value;
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value, P = g & 1;
                PS_ORDER();
                if constexpr (PIPE == 4 && g == 0) fread(FA[0], FB[0], cur, 0);      // (two LDS buffers: a slab's first fragments behind the barrier)
                if constexpr (g + 1 < NG) fread(FA[P ^ 1], FB[P ^ 1], cur, g + 1);
                else if constexpr (FULL && PIPE == 3) fread(FA[P ^ 1], FB[P ^ 1], nxt, 0);
                PS_ORDER();
                static_for<MF>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    fmfma(FA[P], FB[P], mc);
                    if constexpr (FULL) {
                        constexpr int mm = g * MF + m, T = NG * MF, c0 = (mm * NCH + T - 1) / T, c1 = ((mm + 1) * NCH + T - 1) / T;
                        if constexpr (c1 > c0) {
                            PS_ORDER();
                            static_for<c1 - c0>([&](auto cc) {
                                constexpr int c = c0 + decltype(cc)::value;
                                swrite1(wr, kt2 + (PIPE == 3 ? 2 : 1), ra, rb, c);
                                gload1(kt2 + (PIPE == 3 ? 5 : 3), ra, rb, c);
                            });
                            PS_ORDER();
                        }
                    }
                });
            });
            PS_ORDER();
        };
#undef PS_ORDER
        constexpr auto YES = std::integral_constant<bool, true>{};
        constexpr auto NO = std::integral_constant<bool, false>{};
        constexpr auto B0 = std::integral_constant<int, 0>{};
        constexpr auto B1 = std::integral_constant<int, 1>{};
        constexpr auto B2 = std::integral_constant<int, 2>{};
        if constexpr (PIPE == 4) {
            gload(0, ra0, rb0);
            gload(1, ra1, rb1);
            swrite(0, 0, ra0, rb0);
            gload(2, ra0, rb0);
            __syncthreads();
            int kt = 0;
            for (; kt + 2 <= nk; kt += 2) {
                slab(YES, ra1, rb1, kt, B0, B1, B1);
                __syncthreads();
                slab(YES, ra0, rb0, kt + 1, B1, B0, B0);
                __syncthreads();
            }
            if (kt < nk) slab(NO, ra1, rb1, kt, B0, B1, B1);
        } else {
        gload(0, ra0, rb0);
        gload(1, ra1, rb1);
        gload(2, ra2, rb2);
        swrite(0, 0, ra0, rb0);
        gload(3, ra0, rb0);
        swrite(1, 1, ra1, rb1);
        gload(4, ra1, rb1);
        __syncthreads();
        fread(FA[0], FB[0], 0, 0);
        int kt = 0;
        for (; kt + 3 <= nk; kt += 3) {
            slab(YES, ra2, rb2, kt, B0, B1, B2);
            __syncthreads();
            slab(YES, ra0, rb0, kt + 1, B1, B2, B0);
            __syncthreads();
            slab(YES, ra1, rb1, kt + 2, B2, B0, B1);
            __syncthreads();
        }
        if (nk - kt == 2) {
            slab(YES, ra2, rb2, kt, B0, B1, B2);
            __syncthreads();
            slab(NO, ra0, rb0, kt + 1, B1, B2, B0);
        } else if (nk - kt == 1) {
            slab(NO, ra2, rb2, kt, B0, B1, B2);
        }
        }
    }
    if (KS > 1) {           // the two wave groups' partial tiles, added through LDS (see k_gemm_nt)
        __syncthreads();
        float *scr = &As[0][0] + w * 16 * 64;
        static_assert(KS == 1 || 2 * BKT * BM >= 4 * 16 * 64, "reduction scratch does not fit the A buffers");
        if (kg == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[r * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (kg != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += scr[r * 64 + lane];
    }
    float *Cz = a.Cpart + (size_t)z * a.part_stride;
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

// tuning knobs (ps_tune_set): 0 = automatic
int g_gemm_nt_cfg = 0;   // 1 128x128/16, 2 64x128/16, 3 64x64/16, 4 128x32/16, 5 64x64/32, 6 64x128/32, 7 128x128/32, 8 128x32/32
int g_gemm_ablate = 0;  // measurement-only ablation bits for k_gemm_nt (results are garbage when set)
int g_gemm_xcd = 1;      // XCD-aware work-group order on/off (for A/B runs)
int g_gemm_tn_cfg = 0;   // 1 64x64/16, 2 64x64/32, 3 128x128/16, 4 128x32/16, 5 128x32/32

#define NT_LAUNCH(WM, WN, TM, TN, BKT)                                                                   \
    PS_LAUNCH_EV((k_gemm_nt<WM, WN, TM, TN, BKT>), dim3(cdiv(M, WM * TM * 32) * cdiv(N, WN * TN * 32)), dim3(WM * WN * 64), 0, st, stop_ev, a)
#define NT_LAUNCH_P(WM, WN, TM, TN, BKT, KS)                                                             \
    PS_LAUNCH_EV((k_gemm_nt<WM, WN, TM, TN, BKT, KS, 1>), dim3(cdiv(M, WM * TM * 32) * cdiv(N, WN * TN * 32)), dim3(WM * WN * 64 * KS), 0, st, stop_ev, a)
#define NT_LAUNCH_P2(WM, WN, TM, TN, BKT, KS)                                                            \
    PS_LAUNCH_EV((k_gemm_nt<WM, WN, TM, TN, BKT, KS, 2>), dim3(cdiv(M, WM * TM * 32) * cdiv(N, WN * TN * 32)), dim3(WM * WN * 64 * KS), 0, st, stop_ev, a)
#define NT_LAUNCH_KS(WM, WN, TM, TN, BKT, KS)                                                            \
    PS_LAUNCH_EV((k_gemm_nt<WM, WN, TM, TN, BKT, KS>), dim3(cdiv(M, WM * TM * 32) * cdiv(N, WN * TN * 32)), dim3(WM * WN * 64 * KS), 0, st, stop_ev, a)

int gemm_nt(const float *A, int lda, int a_rows, const float *Bt, int ldb, int b_rows, float *C,
            int ldc, int M, int N, int K, int epi, const float *mask, int ldmask, int mask_cols,
            const int *skip_flag, hipStream_t st, LaunchOpts *lo, unsigned int *werr) {
    if (lo) lo->launched = false;
    if ((K & 3) || (lda & 3) || (ldb & 3))
        return ps_set_err(PS_E_BAD_ARG, "gemm_nt: K=%d lda=%d ldb=%d must be multiples of 4", K, lda, ldb);
    if (M <= 0 || N <= 0) return PS_OK;
    const LaunchOpts none;
    const LaunchOpts &o = lo ? *lo : none;
    NtArgs a{A, lda, a_rows, Bt, ldb, b_rows, C, ldc, M, N, K, epi, mask, ldmask, mask_cols, skip_flag, g_gemm_xcd, g_gemm_ablate, stamp_next("gemm_nt"),
             o.flag, o.flag_val, o.wait, o.wait_val, o.prio, wait_bound(werr, 100)};
    const hipEvent_t stop_ev = o.stop_event;
    int cfg = g_gemm_nt_cfg;
    if (cfg == 30 && (K & 15)) cfg = 0;       // the LDS-DMA kernel multiplies whole or half slabs
    if (cfg == 0) {
        // 64x64 tiles put >= 2 workgroups on every CU for the FC shapes of the CTR models
        // (measured best on MI355X for M=4096, N in 256..512, K in 256..528); narrow N: 128x32.
        auto tiles = [&](int bm, int bn) { return (long long)cdiv(M, bm) * cdiv(N, bn); };
        if (N <= 32) cfg = 8;
        else if (tiles(64, 128) >= 2048) cfg = 6;
#if PS_GEMM_LAB
        // 8-wave workgroups on 128 x 64 tiles where that still gives ~one workgroup per CU: the A panel is shared by
        // twice the waves (global traffic per flop -25%); measured alone (tools/gemm_sweep2.py, M = 4096):
        // fwd0 22.7 -> 21.1 us, delta1 14.8 -> 14.0, delta0 (224 workgroups) 23.9 -> 23.1; N = 256 (128 workgroups) 15.4 -> 23.3.
        // In the step the difference disappears (0.1613 vs 0.1614 ms, six runs each): off by default.
        // gemm_pipe: the software-pipelined slab loop (5: 16-wide slabs, three register sets, the default; 0: round 2's loop);
        // gemm_ks: the in-workgroup K split where 64 x 64 tiles give at most ~one workgroup per CU
        else if (g_gemm_8w && tiles(128, 64) >= 200 && tiles(128, 64) <= 1024) cfg = g_gemm_pipe == 5 ? 143 : g_gemm_pipe == 4 ? 133 : g_gemm_pipe == 3 ? 113 : g_gemm_pipe ? 53 : 13;
        else if (g_gemm_ks && tiles(64, 64) <= 320 && K % 16 == 0) cfg = g_gemm_pipe == 3 ? 120 : g_gemm_pipe ? 60 : 20;
        else cfg = g_gemm_pipe == 5 ? 140 : g_gemm_pipe == 4 ? 125 : g_gemm_pipe == 3 ? 105 : g_gemm_pipe ? 45 : 5;
#else
        else cfg = 140;
#endif
    }
    switch (cfg) {
    // ---- the product's three shapes: 64 x 64 tiles on 16-wide slabs with the software-pipelined loop (PIPE = 3); 64 x 128 for
    //      very large problems; 128 x 32 for narrow N
    case 140: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 1, 16, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;     // 16-wide slabs: 24 KB of LDS
    case 6: NT_LAUNCH(2, 2, 1, 2, 32); break;
    case 8: NT_LAUNCH(4, 1, 1, 1, 32); break;
#if PS_GEMM_LAB
    case 1: NT_LAUNCH(2, 2, 2, 2, 16); break;
    case 2: NT_LAUNCH(2, 2, 1, 2, 16); break;
    case 3: NT_LAUNCH(2, 2, 1, 1, 16); break;
    case 4: NT_LAUNCH(4, 1, 1, 1, 16); break;
    case 5: NT_LAUNCH(2, 2, 1, 1, 32); break;
    case 7: NT_LAUNCH(2, 2, 2, 2, 32); break;
    case 9: NT_LAUNCH(2, 2, 2, 1, 32); break;
    case 10: NT_LAUNCH(2, 2, 1, 1, 64); break;
    case 11: NT_LAUNCH(4, 1, 1, 2, 32); break;
    case 12: NT_LAUNCH(1, 4, 2, 1, 32); break;
    case 13: NT_LAUNCH(4, 2, 1, 1, 32); break;      // 8 waves: 128 x 64
    case 14: NT_LAUNCH(2, 4, 1, 1, 32); break;      // 8 waves: 64 x 128
    case 15: NT_LAUNCH(4, 2, 1, 2, 32); break;      // 8 waves: 128 x 128
    case 16: NT_LAUNCH(2, 1, 1, 1, 32); break;      // 2 waves: 64 x 32
    case 17: NT_LAUNCH(1, 2, 1, 1, 32); break;      // 2 waves: 32 x 64
    case 20: NT_LAUNCH_KS(2, 2, 1, 1, 32, 2); break;   // 8 waves on 64 x 64: two wave groups split every K slab
    case 21: NT_LAUNCH_KS(2, 2, 1, 1, 64, 2); break;   // ... with 64-wide slabs
    case 22: NT_LAUNCH_KS(4, 1, 1, 1, 32, 2); break;   // 8 waves on 128 x 32 (narrow N)
    // software-pipelined slab loop (PIPE = 1), same tiles as 5 / 6 / 7 / 13 / 20 / 8
    case 45: NT_LAUNCH_P(2, 2, 1, 1, 32, 1); break;
    case 46: NT_LAUNCH_P(2, 2, 1, 2, 32, 1); break;
    case 47: NT_LAUNCH_P(2, 2, 2, 2, 32, 1); break;
    case 48: NT_LAUNCH_P(4, 1, 1, 1, 32, 1); break;
    case 53: NT_LAUNCH_P(4, 2, 1, 1, 32, 1); break;
    case 60: NT_LAUNCH_P(2, 2, 1, 1, 32, 2); break;
    case 105: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 1, 32, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;      // PIPE = 3
    case 113: PS_LAUNCH_EV((k_gemm_nt<4, 2, 1, 1, 32, 1, 3>), dim3(cdiv(M, 128) * cdiv(N, 64)), dim3(512), 0, st, stop_ev, a); break;
    case 120: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 1, 32, 2, 3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(512), 0, st, stop_ev, a); break;
    case 125: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 1, 32, 1, 4>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;      // PIPE = 4
    case 133: PS_LAUNCH_EV((k_gemm_nt<4, 2, 1, 1, 32, 1, 4>), dim3(cdiv(M, 128) * cdiv(N, 64)), dim3(512), 0, st, stop_ev, a); break;
    case 141: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 1, 16, 1, 4>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;
    case 143: PS_LAUNCH_EV((k_gemm_nt<4, 2, 1, 1, 16, 1, 3>), dim3(cdiv(M, 128) * cdiv(N, 64)), dim3(512), 0, st, stop_ev, a); break;    // 8 waves, 128 x 64: 46 KB
    case 144: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 2, 16, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 128)), dim3(256), 0, st, stop_ev, a); break;    // 64 x 128
    case 145: PS_LAUNCH_EV((k_gemm_nt<2, 1, 1, 1, 16, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 32)), dim3(128), 0, st, stop_ev, a); break;     // 2 waves: 64 x 32
    case 146: PS_LAUNCH_EV((k_gemm_nt<1, 2, 1, 1, 16, 1, 3>), dim3(cdiv(M, 32) * cdiv(N, 64)), dim3(128), 0, st, stop_ev, a); break;     // 2 waves: 32 x 64
    // round 4: fewer LDS fragment reads per MFMA at the SAME 64 x 64 output tile -- two waves, two accumulators each
    case 147: PS_LAUNCH_EV((k_gemm_nt<2, 1, 1, 2, 16, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(128), 0, st, stop_ev, a); break;     // waves 32 x 64
    case 148: PS_LAUNCH_EV((k_gemm_nt<1, 2, 2, 1, 16, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(128), 0, st, stop_ev, a); break;     // waves 64 x 32
    case 149: PS_LAUNCH_EV((k_gemm_nt<2, 1, 1, 2, 32, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(128), 0, st, stop_ev, a); break;     // ... 32-wide slabs
    case 151: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 2, 16, 1, 3>), dim3(cdiv(M, 64) * cdiv(N, 128)), dim3(256), 0, st, stop_ev, a); break;    // (= 144) 64 x 128, 4 waves
    case 142: PS_LAUNCH_EV((k_gemm_nt<2, 2, 1, 1, 64, 1, 4>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;     // 64-wide slabs: 70 KB
    // ... fragments read two groups ahead (PIPE = 2)
    case 85: NT_LAUNCH_P2(2, 2, 1, 1, 32, 1); break;
    case 86: NT_LAUNCH_P2(2, 2, 1, 2, 32, 1); break;
    case 93: NT_LAUNCH_P2(4, 2, 1, 1, 32, 1); break;
    case 90: NT_LAUNCH_P2(2, 2, 1, 1, 32, 2); break;
    // ... on 16x16x4 MFMAs (k_gemm_nt16): 64x64, 64x128, 128x128, 128x64 (8 waves)
    case 65: PS_LAUNCH_EV((k_gemm_nt16<2, 2, 1, 1, 32>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;
    case 66: PS_LAUNCH_EV((k_gemm_nt16<2, 2, 1, 2, 32>), dim3(cdiv(M, 64) * cdiv(N, 128)), dim3(256), 0, st, stop_ev, a); break;
    case 67: PS_LAUNCH_EV((k_gemm_nt16<2, 2, 2, 2, 32>), dim3(cdiv(M, 128) * cdiv(N, 128)), dim3(256), 0, st, stop_ev, a); break;
    case 73: PS_LAUNCH_EV((k_gemm_nt16<4, 2, 1, 1, 32>), dim3(cdiv(M, 128) * cdiv(N, 64)), dim3(512), 0, st, stop_ev, a); break;
    case 30: PS_LAUNCH_EV((k_gemm_nt_lds<3>), dim3(cdiv(M, 64) * cdiv(N, 64)), dim3(256), 0, st, stop_ev, a); break;   // operands DMA'd global -> LDS, 3 stages
#endif
    default:
        return ps_set_err(PS_E_UNSUPPORTED, "gemm_nt_cfg %d is not in this build%s", cfg, PS_GEMM_LAB ? "" : " (the rejected variants live in the lab build: tools/gemm_lab_build.sh)");
    }
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}

// FcLayer.forward of two consecutive relu layers in one launch (k_fc_fwd_pair).  Y1 = relu(A W1t^T) [M][N1] with leading
// dimension ldy1 (the second layer's input incl. its ones column and padding, untouched), Y2 = relu(Y1 W2t^T).  K2 = the
// second problem's K (Y1's padded width).  ctr: [ceil(M / 64) + 8] device words owned by the caller, zero at first use;
// *epoch: the caller's launch count for those counters.  Returns PS_E_UNSUPPORTED when the shapes do not fit (the caller
// then launches the two GEMMs one after the other).
// MEASURED SLOWER, off by default (round 3): the paired launch takes 48 us where the two launches take 20.4 + 3.5 + 13.7.  Its
// ablations say why, and correct an earlier reading of these GEMMs: with the second phase switched off the first alone takes
// 28.5 us inside this launch, and with the wait removed -- both problems' 768 workgroups running fully concurrently, three per
// CU -- the two GEMMs together take 38.5 us: no better than one after the other.  At 64 x 64 tiles these kernels are bound by
// their steady-state THROUGHPUT (~0.5 of the f32 MFMA rate whatever the concurrency), not by launch boundaries; overlapping
// them buys nothing, and the resident waiting workgroups cost the side chains their CU slots (the field sort started 50 us late).
int g_fwd_pair = 0;         // ps_tune_set("fwd_pair", 1): the first two forward GEMMs in one launch (k_fc_fwd_pair)
int gemm_nt_fwd_pair_ok(int M, int N1, int N2, int K1, int K2) {
    if (!PS_GEMM_LAB) return 0;
    if (!g_fwd_pair || g_gemm_nt_cfg != 0) return 0;
    if (M <= 0 || N1 <= 32 || N2 <= 32 || (K1 & 3) || (K2 & 3)) return 0;
    auto tiles = [&](int n) { return (long long)cdiv(M, 64) * cdiv(n, 64); };
    if (tiles(N1) >= 2048 || tiles(N2) >= 2048) return 0;       // (those shapes take the 64 x 128 tiles)
    return 1;
}
int gemm_nt_fwd_pair(const float *A, int lda, int a_rows, const float *W1t, int ldb1, int N1, float *Y1, int ldy1, int K1,
                     const float *W2t, int ldb2, int N2, float *Y2, int ldy2, int K2, int M, unsigned int *ctr, unsigned int *epoch,
                     unsigned int *xcc_err, hipStream_t st, LaunchOpts *lo, unsigned int *werr) {
    if (lo) lo->launched = false;
    if (!gemm_nt_fwd_pair_ok(M, N1, N2, K1, K2)) return ps_set_err(PS_E_UNSUPPORTED, "gemm_nt_fwd_pair: shapes (or not a lab build)");
#if PS_GEMM_LAB
    const LaunchOpts none;
    const LaunchOpts &o = lo ? *lo : none;
    PairArgs q;
    memset(&q, 0, sizeof q);
    q.p1 = NtArgs{A, lda, a_rows, W1t, ldb1, N1, Y1, ldy1, M, N1, K1, EPI_RELU, nullptr, 0, 0, nullptr, 0, g_gemm_ablate, stamp_next("fc_fwd_pair"),
                  o.flag, o.flag_val, nullptr, 0u, o.prio, wait_bound(werr, 105)};
    q.p2 = NtArgs{Y1, ldy1, M, W2t, ldb2, N2, Y2, ldy2, M, N2, K2, EPI_RELU, nullptr, 0, 0, nullptr, 0, 0, nullptr,
                  nullptr, 0u, nullptr, 0u, o.prio, wait_bound(werr, 105)};
    q.mt = cdiv(M, 64); q.tn1 = cdiv(N1, 64); q.tn2 = cdiv(N2, 64); q.lp_max = cdiv(q.mt, 8);
    q.ctr = ctr;
    *epoch += 1;
    q.target = *epoch * (unsigned int)q.tn1;
    q.xcc_err = xcc_err; q.xcc_tag = ctr + q.mt; q.epoch = *epoch & 0x0FFFFFFFu;      // (the 8 words behind the panel counters)
    const int grid = 8 * q.lp_max * (q.tn1 + q.tn2);
    // (fwd_pair = 2, round 5: the product's slab loop -- 16-wide slabs, PIPE = 3 -- inside the pair)
    if (g_fwd_pair == 2) PS_LAUNCH_EV((k_fc_fwd_pair<2, 2, 1, 1, 16, 3>), dim3(grid), dim3(256), 0, st, o.stop_event, q);
    else PS_LAUNCH_EV((k_fc_fwd_pair<2, 2, 1, 1, 32>), dim3(grid), dim3(256), 0, st, o.stop_event, q);
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
#endif
    return PS_OK;
}

int g_keys_early = 1;      // ps_tune_set("keys_early", 0): multi-hot: the sort's keys come from the gather again (the sort chain starts behind it)
int g_dw_late = 0;         // ps_tune_set("dw_late", 1): the first dW GEMM starts with the NEXT delta GEMM (the first delta GEMM runs alone)
int g_tn_prio = 1;          // ps_tune_set("tn_prio", 0 / 2 / 3): the dW GEMMs' wave priority: none / one / two levels under the main chain's; +8: dW_0 only
int g_main_prio = 1;        // ps_tune_set("main_prio", 0): no raised wave priority for the fused step's main-chain kernels
int g_gemm_pipe = 5;        // ps_tune_set("gemm_pipe", ...): 0 round 2's slab loop; 1 / 3 pipelined across the barrier on three LDS buffers with two / three register sets (32-wide slabs, 55 KB); 4 pipelined inside the slab, two buffers (37 KB); 5 (default): as 3 on 16-wide slabs (30 KB)
int g_gemm_ks = 0;          // ps_tune_set("gemm_ks", 1): 8 waves (K split inside the workgroup) on shapes with <= ~one 64 x 64 tile per CU
int g_gemm_8w = 0;          // ps_tune_set("gemm_8w", 1): 8-wave 128 x 64 tiles where they fit (faster alone, no gain in the step)
int g_radix_scan_free = 1;   // ps_tune_set("radix_scan_free", 0): a scan launch between the counts and the scatter of every radix pass again
int g_plan_mid = 1;         // ps_tune_set("plan_mid", 0): ps_shard_step's next plan head behind the running step's backward enqueue again (side chain 0; round 4)
int g_plan_early = 1;       // ps_tune_set("plan_early", 0): ps_shard_step's next plan in the running step's tail (main stream) again
int g_wide_slots = 1;       // ps_tune_set("wide_slots", 0): the sharded step all-reduces the wide part as dense G | C vectors (rounds 2-4) instead of per-worker slots
#if !PS_GEMM_LAB
// the product build has no row-panel forward (csrc/lab/kernels_panel.hip: built, bit-checked, measured, not adopted -- lab build only)
// and no fragment-order weights: ps_model.hip then takes the k_gemm_nt launches
int fwd_panel_shape_ok(int, int, int) { return 0; }
int launch_fwd_panel(const FwdPanelArgs &, const LastBwdArgs *, const HeadArgs *, int, hipStream_t, LaunchOpts *lo, unsigned int *) {
    if (lo) lo->launched = false;
    return ps_set_err(PS_E_UNSUPPORTED, "k_fwd_panel lives in the lab build (tools/gemm_lab_build.sh)");
}
int launch_pack_w(const float *, float *, int, int, hipStream_t) { return PS_OK; }
#endif
int g_fwd_panel = 0;        // ps_tune_set("fwd_panel", v), LAB build: 0 the FC forward as k_gemm_nt launches, 1 the two hidden layers of a 16-row panel in one launch (kernels_panel.hip), 2 with the head
int g_sort_layer = 0;       // ps_tune_set("sort_layer", l): the single-hot field sort is released by forward GEMM l's start (0: the first)
int g_sort_late = 0;        // ps_tune_set("sort_late", 1): the single-hot field sort behind the first delta GEMM's release instead of the first forward GEMM's
int g_tail_defer = 1;       // ps_tune_set("tail_defer", 0): the fused step's embedding update holds the join with the dense update again
int g_dw_split = 0;         // ps_tune_set("dw_split", 1): the first dW GEMM on side chain 0, the others on side chain 1
int g_tn_start_wait = 1;    // ps_tune_set("tn_start_wait", 0): a spinner launch in front of EVERY dW GEMM again
int g_tail_fused = 1;       // ps_tune_set("tail_fused", 0): the dense update between a spinner and a flag-setter launch, the main chain ends behind a spinner again
int g_end_wait = 1;         // ps_tune_set("end_wait", 0): the main chain joins side chain 0 behind a spinner launch again
int g_tail_dev = 1;         // ps_tune_set("tail_dev", 0): dense update last on the main chain again
int g_dev_wait = 1;         // ps_tune_set("dev_wait", 0): the dW chain waits for the head by event again
// A kernel that waits for a flag raised by a kernel of ANOTHER stream needs the two streams to make progress at the same
// time.  Environments that run one kernel at a time, whichever queue it comes from, break that: a profiler collecting
// hardware counters (rocprofv3 --pmc announces itself through ROCPROF_COUNTER_COLLECTION in the child's environment),
// HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL (every launch waits for the one before), a debugger's serialisation
// (HSA_ENABLE_DEBUG), and a runtime with fewer hardware queues than the step has chains (GPU_MAX_HW_QUEUES < 4: two
// streams of one model may then share a queue, the waiter in front of its releaser).  There every device-side wait is
// replaced by its event form when the library is loaded (the step's results are the same bit for bit,
// tests/test_gpu_schedule.py); anything this list misses ends in a bounded wait's timeout instead of a hang.
const char *g_dev_wait_off_reason = nullptr;
namespace {
bool env_on(const char *name) {
    const char *e = getenv(name);
    return e && *e && strcmp(e, "0") != 0 && strcasecmp(e, "false") != 0 && strcasecmp(e, "off") != 0;
}
const int g_serial_env_guard = []() {
    const char *why = nullptr;
    if (env_on("ROCPROF_COUNTER_COLLECTION")) why = "ROCPROF_COUNTER_COLLECTION";
    else if (env_on("HIP_LAUNCH_BLOCKING")) why = "HIP_LAUNCH_BLOCKING";
    else if (env_on("AMD_SERIALIZE_KERNEL")) why = "AMD_SERIALIZE_KERNEL";
    else if (env_on("HSA_ENABLE_DEBUG")) why = "HSA_ENABLE_DEBUG";
    else if (const char *q = getenv("GPU_MAX_HW_QUEUES")) { if (*q && atoi(q) < 4) why = "GPU_MAX_HW_QUEUES < 4"; }
    if (why) { g_dev_wait = 0; g_end_wait = 0; g_dev_wait_off_reason = why; }
    return 0;
}();
}  // namespace
unsigned long long g_spin_timeout_ticks = 200000000ull;      // 2 s of the device's 100 MHz wall clock; ps_tune_set("spin_timeout_ms")
WaitBound wait_bound(unsigned int *werr, unsigned int code) { return WaitBound{werr, werr ? g_spin_timeout_ticks : 0ull, code}; }

// A stream that reaches a hipStreamWaitEvent before the event has fired resumes 10-20 us after it (the dW chain
// started 10 us after the head on a good day and 18 on a bad one, which then pushed dW0 under the embedding update:
// a 170 / 178 us step, fixed per process).  A kernel that is already RUNNING when its condition comes true ends at
// once, and the next kernel of its stream starts ~3 us later like any in-order successor: the side chain parks this
// one-wave spinner in front of its first GEMM, and the main chain's next launch -- which starts only after the head
// has finished and released its writes -- flips the flag from its first workgroup.
// (epochs only grow: "reached or passed", wrap-safe, so that a flag already moved on can never strand a waiter)
__global__ void k_spin_until(const unsigned int *flag, unsigned int val, WaitBound b) {
    if (threadIdx.x == 0) spin_bounded(flag, val, b);
}
__global__ void k_flag_set(unsigned int *flag, unsigned int val) {
    __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// in stream order behind the work it stands for: that work has finished and released its writes when this runs
int launch_flag_set(unsigned int *flag, unsigned int val, hipStream_t st) {
    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(1), 0, st, flag, val);
    HIPCHK(hipGetLastError());
    return PS_OK;
}
__global__ void k_set_then_spin(unsigned int *set, unsigned int set_val, const unsigned int *flag, unsigned int val, WaitBound b) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(set, set_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_bounded(flag, val, b);
    }
}
int launch_set_then_spin(unsigned int *set, unsigned int set_val, const unsigned int *flag, unsigned int val, hipStream_t st, unsigned int *werr, unsigned int code) {
    hipLaunchKernelGGL(k_set_then_spin, dim3(1), dim3(64), 0, st, set, set_val, flag, val, wait_bound(werr, code));
    HIPCHK(hipGetLastError());
    return PS_OK;
}
__global__ void k_set_then_spin2(unsigned int *set, unsigned int set_val, const unsigned int *flag, unsigned int val, const unsigned int *flag2, unsigned int val2, WaitBound b) {
    if (threadIdx.x == 0) {
        if (set) __hip_atomic_store(set, set_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_bounded(flag, val, b);
        if (flag2) spin_bounded(flag2, val2, b);
    }
}
int launch_set_then_spin2(unsigned int *set, unsigned int set_val, const unsigned int *flag, unsigned int val, const unsigned int *flag2, unsigned int val2,
                          hipStream_t st, unsigned int *werr, unsigned int code) {
    hipLaunchKernelGGL(k_set_then_spin2, dim3(1), dim3(64), 0, st, set, set_val, flag, val, flag2, val2, wait_bound(werr, code));
    HIPCHK(hipGetLastError());
    return PS_OK;
}
int launch_spin_until(const unsigned int *flag, unsigned int val, hipStream_t st, unsigned int *werr, unsigned int code) {
    hipLaunchKernelGGL(k_spin_until, dim3(1), dim3(64), 0, st, flag, val, wait_bound(werr, code));
    HIPCHK(hipGetLastError());
    return PS_OK;
}
int g_ext_events = 1;       // ps_tune_set("ext_events", 0): cross-stream events by hipEventRecord again
int g_sort_ablate = 0;      // measurement only
int g_field_sort = 1;       // ps_tune_set("field_sort", 0): single-hot batches go through the general radix sort too
int g_last_rows = 0;        // ps_tune_set("last_rows", rows per k_last_bwd workgroup)
int g_gemm_tn_target = 0;   // ps_tune_set("gemm_tn_target", workgroups): override the workgroup target of the split choice
int gemm_tn_choose_split(int Kout, int N, int M) {
    const long long tiles = N <= 32 ? (long long)cdiv(Kout, 128) * cdiv(N, 32) : (long long)cdiv(Kout, 64) * cdiv(N, 64);
    // Every split is a partial slab written here and re-read by the dense update.  Alone on the chip dW0 (56 tiles)
    // is fastest at 8 splits (24.2 us vs 27.1 us at 4, tools/gemm_sweep2.py), but in the step the dW GEMMs share the
    // CUs with the data-gradient GEMMs on the main stream, and ~224 workgroups (dW0 4 splits, dW1 7) gives the same
    // or a slightly shorter step (tools/tn_split_step.py: 0.194-0.198 ms vs 0.200 ms at 448) at half the slab traffic
    const long long target = g_gemm_tn_target > 0 ? g_gemm_tn_target : 224;
    int s = (int)((target + tiles - 1) / tiles);
    const int max_s = M / 128 > 0 ? M / 128 : 1;        // keep >= 128 batch rows per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

#define TN_LAUNCH(WM, WN, TM, TN, BKT)                                                                      \
    hipLaunchKernelGGL((k_gemm_tn<WM, WN, TM, TN, BKT>),                                                    \
                       dim3(cdiv(Kout, WM * TM * 32) * cdiv(N, WN * TN * 32) * nsplit), dim3(256), 0, st, a)
#define TN_LAUNCH_P(WM, WN, TM, TN, BKT, KS)                                                                \
    hipLaunchKernelGGL((k_gemm_tn<WM, WN, TM, TN, BKT, KS, 3>),                                             \
                       dim3(cdiv(Kout, WM * TM * 32) * cdiv(N, WN * TN * 32) * nsplit), dim3(256 * KS), 0, st, a)
#define TN_LAUNCH_P4(WM, WN, TM, TN, BKT, KS)                                                               \
    hipLaunchKernelGGL((k_gemm_tn<WM, WN, TM, TN, BKT, KS, 4>),                                             \
                       dim3(cdiv(Kout, WM * TM * 32) * cdiv(N, WN * TN * 32) * nsplit), dim3(256 * KS), 0, st, a)
#define TN_LAUNCH_KS(WM, WN, TM, TN, BKT, KS)                                                               \
    hipLaunchKernelGGL((k_gemm_tn<WM, WN, TM, TN, BKT, KS>),                                                \
                       dim3(cdiv(Kout, WM * TM * 32) * cdiv(N, WN * TN * 32) * nsplit), dim3(256 * KS), 0, st, a)

int gemm_tn_splitk(const float *A, int lda, int a_cols, const float *D, int ldd, int d_cols,
                   float *Cpart, int ldc, int64_t part_stride, int Kout, int N, int M, int nsplit,
                   const int *skip_flag, hipStream_t st, const LaunchOpts *lo, unsigned int *werr) {
    if ((lda & 3) || (ldd & 3) || (a_cols & 3) || (d_cols & 3))
        return ps_set_err(PS_E_BAD_ARG, "gemm_tn: leading dims / cols must be multiples of 4");
    if (Kout <= 0 || N <= 0 || nsplit <= 0) return PS_OK;
    const int mchunk = (int)round_up(cdiv(M > 0 ? M : 1, nsplit), 32);
    TnArgs a{A, lda, a_cols, D, ldd, d_cols, Cpart, ldc, (long long)part_stride, Kout, N, M, mchunk, skip_flag, g_gemm_xcd, stamp_next("gemm_tn"), lo ? lo->prio : 0,
             lo ? lo->wait : nullptr, lo ? lo->wait_val : 0u, wait_bound(werr, 101)};
    int cfg = g_gemm_tn_cfg;
    // 8 waves per 64 x 64 tile, the slab's batch rows split over two wave groups: ~224 workgroups of 4 waves are ONE wave
    // per SIMD -- nothing hides a barrier or an LDS round trip; alone dW0 27.3 -> 25.3 us, dW1 17.5 -> 16.5, in the step
    // 0.1437 -> 0.1413 ms (A/B x2, round 3)
    if (cfg == 0) cfg = N <= 32 ? 5 : 6;
    switch (cfg) {
    // ---- the product's two shapes
    case 6: TN_LAUNCH_KS(2, 2, 1, 1, 32, 2); break;    // 8 waves on 64 x 64: two wave groups split every slab's batch rows
    case 5: TN_LAUNCH(4, 1, 1, 1, 32); break;          // narrow N: 128 x 32
#if PS_GEMM_LAB
    case 1: TN_LAUNCH(2, 2, 1, 1, 16); break;
    case 2: TN_LAUNCH(2, 2, 1, 1, 32); break;
    case 3: TN_LAUNCH(2, 2, 2, 2, 16); break;
    case 4: TN_LAUNCH(4, 1, 1, 1, 16); break;
    case 7: TN_LAUNCH_KS(2, 2, 1, 1, 64, 2); break;    // ... with 64-row slabs
    case 12: TN_LAUNCH_P(2, 2, 1, 1, 32, 1); break;    // software-pipelined slab loop (PIPE = 3), 4 waves
    case 16: TN_LAUNCH_P(2, 2, 1, 1, 32, 2); break;    // ... 8 waves (K split)
    case 17: TN_LAUNCH_P(2, 2, 1, 1, 64, 2); break;    // ... 64-row slabs
    case 22: TN_LAUNCH_P4(2, 2, 1, 1, 32, 1); break;   // pipelined inside the slab, two LDS buffers (PIPE = 4), 4 waves
    case 26: TN_LAUNCH_P4(2, 2, 1, 1, 32, 2); break;   // ... 8 waves (K split)
    case 27: TN_LAUNCH_P4(2, 2, 1, 1, 64, 2); break;   // ... 64-row slabs
    case 32: TN_LAUNCH_P(2, 2, 1, 1, 16, 1); break;    // PIPE = 3 on 16-row slabs (24 KB of LDS), 4 waves
#endif
    default:
        return ps_set_err(PS_E_UNSUPPORTED, "gemm_tn_cfg %d is not in this build%s", cfg, PS_GEMM_LAB ? "" : " (the rejected variants live in the lab build: tools/gemm_lab_build.sh)");
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}
