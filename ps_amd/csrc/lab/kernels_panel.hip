// kernels_panel.hip -- the FC forward chain of a 16-row panel in ONE launch (round 5):
//   FcLayer.forward x 2 (relu)           layer/FcLayer.java:74-91
//   [ head + FcLayer.backward of the out = 1 layer for the same rows     kernels_head.inc ]
// The FC chain is local to a batch row, so a workgroup that owns 16 rows needs nobody else's results from the gather's
// output to delta_2: no hand-over between workgroups, no kernel boundary between the layers.  The rows' activations live
// in LDS as [k / 4][row][4] images (1 KiB per 16-k step: conflict-free ds_read_b128, and the accumulator layout of
// v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand -- a lane's four registers are four consecutive output
// features of one row -- is one ds_write_b128 into the next layer's image).  The weights never touch LDS: they are
// streamed from L2 in MFMA-fragment order (FcParams.Wp, written by k_dense_update beside W' and Wt: every wave load is
// 1 KiB contiguous), D steps ahead of the MFMAs that use them, and the next layer's first steps are requested before the
// epilogue and the barrier between the layers.
// Price: weight re-use per CU is 16 rows -- (W0 + W1) x 256 workgroups = 353 MB from L2 per forward at configs[1]'s
// shape, 2x what 64 x 64 tiles read -- measured (tools/ubench/rowpanel_mlp.hip): 32.7 us for both layers against 25 with
// the loads hitting L1; k_gemm_nt's two launches take 18.8 + 12.4 us and a 3 us boundary in the step.  What the launch
// buys is the boundaries and the head: fwd0 | fwd1 | head_last_bwd become one kernel.
// Products are summed in another order than k_gemm_nt's (k ascending in steps of 4 per MFMA here, per 32 x 32 x 2 there):
// results agree to f32 rounding, not bit for bit; every model uses ONE of the two forms for all its steps (fused, split,
// sharded alike: ps_model.hip fwd_panel_on), so the schedules still agree bit for bit with each other.
// MEASURED, NOT THE PRODUCT'S FORWARD (round 5, profiles/r05_fwd_panel.txt): in the step the launch takes 40.4 us with the head against
// 18.9 + 2.8 + 12.4 + 2.3 + 7.6 = 44 us for fc_fwd0 | fc_fwd1 | head_last_bwd, and the step 0.1336 against 0.1333 ms -- the two
// layers run at 0.60 of the f32 MFMA rate here (the L2 -> CU stream of the weights: 24.7 us for both layers with the loads hitting
// L1, 32.7 from L2), and with ONE long-lived workgroup per CU nothing overlaps its x staging, its two epilogues, barriers and the
// head's serial arithmetic (13 us of 41 in which the CU issues no MFMA; tools/panel_timing.sh).  Compiled into the LAB build only
// (tools/gemm_lab_build.sh, -DPS_GEMM_LAB=1; ps_tune_set("fwd_panel", 1 | 2)); the product library carries stubs, allocates no Wp
// and its k_dense_update writes none.
#include <string.h>

#include "ps_common.h"
#include "kernels_emb.h"

#ifndef PS_GEMM_LAB
#define PS_GEMM_LAB 0
#endif
#if PS_GEMM_LAB

namespace {

#ifdef PS_PANEL_TIMING      // (tools/panel_timing.sh: wall-clock stamps of each workgroup's phases; the product build carries none)
__device__ unsigned long long g_panel_t[256 * 16];
#define PANEL_T(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_panel_t[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define HEAD_T(k) PANEL_T(8 + (k))
#else
#define PANEL_T(k) do { } while (0)
#define HEAD_T(k) do { } while (0)
#endif
#include "kernels_head.inc"

typedef float vf4 __attribute__((ext_vector_type(4)));

// One layer for this wave: T tiles of 16 output features, KS steps of 16 k, weights D steps ahead.
// wp: this wave's first tile of the packed weights [tile][ks][lane] (vf4); in: LDS image [k / 4][row] (vf4).
template <int KS, int T, int D>
__device__ __forceinline__ void panel_prefetch(const vf4 *__restrict__ wp, int lane, vf4 (&wr)[D][T]) {
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int t = 0; t < T; ++t) wr[d][t] = wp[((size_t)t * KS + d) * 64 + lane];
}
template <int KS, int T, int D>
__device__ __forceinline__ void panel_mm(const vf4 *__restrict__ wp, const vf4 *in, int lane, vf4 (&acc)[T], vf4 (&wr)[D][T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = vf4{0.f, 0.f, 0.f, 0.f};
    vf4 b = in[lane];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        vf4 bn = b;
        if (ks + 1 < KS) bn = in[(ks + 1) * 64 + lane];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const vf4 w = wr[ks % D][t];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], b[3], acc[t], 0, 0, 0);
            if (ks + D < KS) wr[ks % D][t] = wp[((size_t)t * KS + ks + D) * 64 + lane];
        }
        b = bn;
    }
}

constexpr int PANEL_WAVES = 8, PANEL_D = 4;

// A barrier that orders LDS only: __syncthreads() also waits for the wave's outstanding global stores (the rows of H1 / H2 just
// written, 1-2 us to drain) and loads (the next layer's first weight steps, requested on purpose before the barrier).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the head's loaders over LDS: a row of the [k / 4][row][4] image, and a plain vector
struct PanelLdsRow {
    const vf4 *img; int row;
    __device__ __forceinline__ float4 v4(int k) const { const vf4 t = img[(k >> 2) * 16 + row]; return make_float4(t[0], t[1], t[2], t[3]); }
    __device__ __forceinline__ float at(int k) const { return reinterpret_cast<const float *>(img)[(((k >> 2) * 16 + row) << 2) + (k & 3)]; }
};
struct PanelLdsVec {
    const float *p;
    __device__ __forceinline__ float4 v4(int k) const { return *reinterpret_cast<const float4 *>(p + k); }
    __device__ __forceinline__ float at(int k) const { return p[k]; }
};

// KS0 = Kpad0 / 16 steps of layer 0, N0 / N1 outputs of the two layers (multiples of 128: whole tiles per wave).
// HEAD: the head (kernels_head.inc head_one_t) and the out = 1 layer's backward (last_bwd_rows' arithmetic, row by row in order) of
// the same 16 rows, fed from the LDS image of layer 1's output; what they need from memory -- wide ids -> wide weights, the label,
// the out = 1 layer's weights -- is requested when the kernel starts and arrives while the GEMMs run.
template <int KS0, int N0, int N1, bool HEAD>
__global__ __launch_bounds__(PANEL_WAVES * 64) void k_fwd_panel(FwdPanelArgs a, LastBwdArgs q, HeadArgs h) {
    constexpr int T0 = N0 / 16 / PANEL_WAVES, T1 = N1 / 16 / PANEL_WAVES, KS1 = N0 / 16 + 1, KS2 = N1 / 16 + 1, D = PANEL_D;
    static_assert(KS2 <= KS0, "layer 1's output image re-uses layer 0's input image");
    __shared__ vf4 xs[KS0 * 64];           // layer 0's input, later layer 1's output (HEAD)
    __shared__ vf4 h1s[KS1 * 64];
    __shared__ float dsh[HEAD ? PS_PANEL_ROWS : 1];
    __shared__ __attribute__((aligned(16))) float wls[HEAD ? KS2 * 16 : 4];
    if (a.prio) __builtin_amdgcn_s_setprio(3);       // main-chain kernel of the fused step (see k_gemm_nt)
    EndWait end_wait(a.wait_flag, a.wait_val, a.bound);
    StampScope stamp(a.ts);
    if (a.flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * PS_PANEL_ROWS;
    const int r1 = row0 + PS_PANEL_ROWS < a.B ? row0 + PS_PANEL_ROWS : a.B;      // (row0 >= B: a workgroup that only writes its zero slab)
    const vf4 *__restrict__ w0 = reinterpret_cast<const vf4 *>(a.W0p) + (size_t)w * T0 * KS0 * 64;
    const vf4 *__restrict__ w1 = reinterpret_cast<const vf4 *>(a.W1p) + (size_t)w * T1 * KS1 * 64;
    PANEL_T(0);
    vf4 wr0[D][T0];
    panel_prefetch<KS0, T0, D>(w0, lane, wr0);
    HeadPre pre;
    float wcol = 0.f;
    const int hb = row0 + (tid >> 3) < r1 ? row0 + (tid >> 3) : r1 - 1;      // (tid < 128: eight lanes per row)
    if (HEAD) {
        if (tid < 128) head_prefetch_ids(h, hb, tid & 7, pre);
        if (tid < q.K) wcol = q.W[(size_t)tid * q.ldw];
        for (int i = tid; i < KS2 * 16; i += PANEL_WAVES * 64) wls[i] = i < h.k_last ? h.w_last[i] : 0.f;
    }
    // the panel's rows of layer 0's input (ones column and zero padding are in the buffer: ps_model.hip FcBuf.A) -> LDS image
    for (int i = tid; i < KS0 * 64; i += PANEL_WAVES * 64) {
        const int r = i / (KS0 * 4), g = i % (KS0 * 4);      // consecutive threads walk a row
        const int row = row0 + r < a.B ? row0 + r : a.B - 1;
        xs[g * 16 + r] = *reinterpret_cast<const vf4 *>(a.X + (size_t)row * a.ldx + g * 4);
    }
    if (HEAD && tid < 128) head_prefetch_weights(h, tid & 7, pre);      // (the ids have arrived with the rows)
    lds_barrier();
    PANEL_T(1);
    vf4 acc0[T0];
    panel_mm<KS0, T0, D>(w0, xs, lane, acc0, wr0);
    PANEL_T(2);
    vf4 wr1[D][T1];
    panel_prefetch<KS1, T1, D>(w1, lane, wr1);
    // epilogue 0: relu; lane (row = lane % 16, quarter = lane / 16) holds features 16 * tile + 4 * quarter .. + 3
    const int prow = lane & 15, grow = row0 + prow;
#pragma unroll
    for (int t = 0; t < T0; ++t) {
        vf4 v = acc0[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        const int g = (w * T0 + t) * 4 + (lane >> 4);
        h1s[g * 16 + prow] = v;
        if (grow < a.B) *reinterpret_cast<vf4 *>(a.H1 + (size_t)grow * a.ld1 + g * 4) = v;
    }
    if (tid < 64) {      // the ones column that carries layer 1's bias, and the padding behind it
        vf4 v = {0.f, 0.f, 0.f, 0.f};
        if ((tid >> 4) == 0) v[0] = 1.f;
        h1s[(N0 / 4 + (tid >> 4)) * 16 + (tid & 15)] = v;
    }
    lds_barrier();
    PANEL_T(3);
    vf4 acc1[T1];
    panel_mm<KS1, T1, D>(w1, h1s, lane, acc1, wr1);
    PANEL_T(4);
#pragma unroll
    for (int t = 0; t < T1; ++t) {
        vf4 v = acc1[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        const int g = (w * T1 + t) * 4 + (lane >> 4);
        if (HEAD) xs[g * 16 + prow] = v;        // (every wave is past the barrier behind layer 0: its input image is free)
        if (grow < a.B) *reinterpret_cast<vf4 *>(a.H2 + (size_t)grow * a.ld2 + g * 4) = v;
    }
    if (!HEAD) return;
    if (tid < 64) {      // the out = 1 layer's ones column and padding
        vf4 v = {0.f, 0.f, 0.f, 0.f};
        if ((tid >> 4) == 0) v[0] = 1.f;
        xs[(N1 / 4 + (tid >> 4)) * 16 + (tid & 15)] = v;
    }
    lds_barrier();
    PANEL_T(5);
    if (tid < 128) {
        const bool valid = row0 + (tid >> 3) < r1;
        const float d = head_one_t<true>(h, hb, tid & 63, valid, PanelLdsRow{xs, tid >> 3}, PanelLdsVec{wls}, &pre);
        if (valid && (tid & 7) == 0) dsh[tid >> 3] = d;
    }
    PANEL_T(6);
    lds_barrier();
    PANEL_T(7);
    // FcLayer.backward of the out = 1 layer for these rows (kernels_head.inc last_bwd_rows: the same sums in the same order)
    if (tid == 255) {
        float accb = 0.f;
        for (int b = row0; b < r1; ++b) accb += 1.0f * dsh[b - row0];
        q.part[(size_t)blockIdx.x * q.part_stride + (size_t)q.K * q.ldpart] = accb;
    }
    if (tid < q.K) {
        const int k = tid;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < PS_PANEL_ROWS; ++j) {
            if (row0 + j < r1) {
                const float x = reinterpret_cast<const float *>(xs)[(((k >> 2) * 16 + j) << 2) + (k & 3)], d = dsh[j];
                acc += x * d;                               // rows in order: the sequential batch sum
                if (k < q.dprev_cols) {
                    float v = wcol * d;                     // weights.transpose().mmul(delta) with one output
                    if (k < q.mask_cols) v *= x > 0.f ? 1.f : 0.f;   // the previous layer's relu'
                    q.dprev[(size_t)(row0 + j) * q.ldp + k] = v;
                }
            }
        }
        q.part[(size_t)blockIdx.x * q.part_stride + (size_t)k * q.ldpart] = acc;
    }
    PANEL_T(8);
}

// Wp[tile][ks][lane][m] = Wt[16 tile + lane % 16][16 ks + 4 (lane / 16) + m]   (Wt [N][Kpad]: zero beyond k = K)
__global__ __launch_bounds__(256) void k_pack_w(const float *__restrict__ Wt, float *__restrict__ Wp, int N, int Kpad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int KS = Kpad >> 4;
    if (i >= (int64_t)(N >> 4) * KS * 64) return;
    const int lane = (int)(i & 63);
    const int64_t ts = i >> 6;
    const int ks = (int)(ts % KS), tile = (int)(ts / KS);
    const int n = tile * 16 + (lane & 15), k = ks * 16 + 4 * (lane >> 4);
    reinterpret_cast<vf4 *>(Wp)[i] = *reinterpret_cast<const vf4 *>(Wt + (size_t)n * Kpad + k);
}

}  // namespace

#ifdef PS_PANEL_TIMING
extern "C" int ps_dbg_panel_timing(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_t), sizeof(unsigned long long) * 256 * 16) == hipSuccess ? 0 : -1;
}
#endif

int fwd_panel_shape_ok(int Kpad0, int N0, int N1) { return Kpad0 == 432 && N0 == 512 && N1 == 256; }

int launch_fwd_panel(const FwdPanelArgs &args, const LastBwdArgs *q, const HeadArgs *h, int head_wgs, hipStream_t st, LaunchOpts *lo, unsigned int *werr) {
    if (lo) lo->launched = false;
    if (args.B <= 0) return PS_OK;
    if (!fwd_panel_shape_ok(args.Kpad0, args.N0, args.N1) || (args.ldx & 3) || (args.ld1 & 3) || (args.ld2 & 3))
        return ps_set_err(PS_E_BAD_ARG, "launch_fwd_panel: shape %d x %d x %d not built", args.Kpad0, args.N0, args.N1);
    if (q && (q->chunk != PS_PANEL_ROWS || !h || !h->labels || !h->a_last || q->K != args.N1 || h->k_last > args.N1 + 16)) return ps_set_err(PS_E_BAD_ARG, "launch_fwd_panel: head needs %d rows per workgroup and labels", PS_PANEL_ROWS);
    const LaunchOpts none;
    const LaunchOpts &o = lo ? *lo : none;
    FwdPanelArgs a = args;
    a.ts = stamp_next(q ? "fwd_panel_head" : "fwd_panel");
    a.flag = o.flag; a.flag_val = o.flag_val; a.wait_flag = o.wait; a.wait_val = o.wait_val; a.prio = o.prio; a.bound = wait_bound(werr, 101);
    // (with the head: every slab k_dense_update sums is written, rows beyond B by workgroups that multiply a clamped row and store nothing)
    const int wgs = cdiv(a.B, PS_PANEL_ROWS);
    const dim3 grid(q && head_wgs > wgs ? head_wgs : wgs), block(PANEL_WAVES * 64);
    LastBwdArgs qq; HeadArgs hh;
    memset(&qq, 0, sizeof qq); memset(&hh, 0, sizeof hh);
    if (q) { qq = *q; hh = *h; PS_LAUNCH_EV((k_fwd_panel<27, 512, 256, true>), grid, block, 0, st, o.stop_event, a, qq, hh); }
    else PS_LAUNCH_EV((k_fwd_panel<27, 512, 256, false>), grid, block, 0, st, o.stop_event, a, qq, hh);
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}

int launch_pack_w(const float *Wt, float *Wp, int N, int Kpad, hipStream_t st) {
    if (!Wp || (N & 15) || (Kpad & 15)) return PS_OK;
    hipLaunchKernelGGL(k_pack_w, dim3(cdiv((int64_t)(N >> 4) * (Kpad >> 4) * 64, 256)), dim3(256), 0, st, Wt, Wp, N, Kpad);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

#else
#error "lab/kernels_panel.hip is compiled into the measurement build only (tools/gemm_lab_build.sh, -DPS_GEMM_LAB=1)"
#endif      // PS_GEMM_LAB
