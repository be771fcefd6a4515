// kernels_emb.hip -- the sparse half of the hot path on gfx950:
//   EmbeddingLayer/EmbeddingField.forward   layer/EmbeddingLayer.java:25-48, layer/EmbeddingField.java:66-78
//   LRLayer.forward + AddLayer + CrossEntropy layer/LRLayer.java:62-98, layer/AddLayer.java:33-48, loss/CrossEntropy.java:10-28
//   EmbeddingField.backward (twice) + KVStore.sum/update + Adam/Ftrl
//                                            layer/EmbeddingField.java:86-104, store/KVStore.java:192-268,
//                                            update/AdamUpdater.java:57-70, update/FtrlUpdater.java:51-76
// These are HBM-bound gather/scatter kernels: rows move as 16-B lanes
// (D/4 lanes per row, 64/(D/4) rows per wave instruction), outputs of one
// sample are contiguous so a wave stores whole 1-KiB runs, and the per-key
// reduction walks sorted runs in batch order (no float atomics; bit-exact
// against the oracle's sequential order).  Compiled with -ffp-contract=off:
// every reference op is individually rounded.
#include <string.h>

#include "ps_common.h"
#include "kernels_emb.h"
#include "ps_put.h"

#ifndef PS_GEMM_LAB
#define PS_GEMM_LAB 0
#endif

namespace {

template <int VEC> struct Vec;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <> struct Vec<4> {
    float4 v;
    __device__ __forceinline__ static Vec load(const float *p) { Vec r; r.v = *reinterpret_cast<const float4 *>(p); return r; }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = v; }
    // streaming variants (slc/nt): rows of a table far larger than the caches are read once, outputs are written once
    __device__ __forceinline__ static Vec load_nt(const float *p) {
        const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
        Vec r; r.v = make_float4(t.x, t.y, t.z, t.w); return r;
    }
    __device__ __forceinline__ void store_nt(float *p) const {
        f32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t *>(p));
    }
    __device__ __forceinline__ static Vec zero() { Vec r; r.v = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
    __device__ __forceinline__ float &at(int i) { return (&v.x)[i]; }
    __device__ __forceinline__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};
template <> struct Vec<1> {
    float v;
    __device__ __forceinline__ static Vec load(const float *p) { Vec r; r.v = *p; return r; }
    __device__ __forceinline__ void store(float *p) const { *p = v; }
    __device__ __forceinline__ static Vec load_nt(const float *p) { Vec r; r.v = __builtin_nontemporal_load(p); return r; }
    __device__ __forceinline__ void store_nt(float *p) const { __builtin_nontemporal_store(v, p); }
    __device__ __forceinline__ static Vec zero() { Vec r; r.v = 0.f; return r; }
    __device__ __forceinline__ float &at(int) { return v; }
    __device__ __forceinline__ float get(int) const { return v; }
};
#define VFOR(i) _Pragma("unroll") for (int i = 0; i < VEC; ++i)

// ---------------------------------------------------------------------------
// forward gather (+ bag sum, + relu, + dense concat, + sort keys)
// ---------------------------------------------------------------------------
// GATHER_ILP independent row loads per lane group are in flight at once: single-hot groups take
// GATHER_ILP consecutive bags (their outputs are contiguous), multi-hot groups walk their bag
// GATHER_ILP entries at a time.  One row per group and launch kept too little memory in flight
// to cover HBM latency (Little: ~8 MB chip-wide at 8 TB/s): measured 4.9 -> see profiles/.
#ifndef GATHER_ILP
#define GATHER_ILP 4
#endif
#define GATHER_ILP_MH 1   // measured on a 256 GB table, bags of 32: 1 -> 4.94, 2 -> 4.65, 4 -> 4.8, 8 -> 4.31 TB/s (more in flight per wave only costs occupancy)
// LRLayer.forward of sample b by its 8-lane group (layer/LRLayer.java:73-84): the head's wide part (kernels_head.inc head_one_t) on its own --
// the same loads, the same sequential f32 sum in field order, the same + bias, the same touched marks and error count: wide_z[b] has the
// bits the head would have computed.  As a role of the gather's launch the id -> weight round trip leaves the head, i.e. the step's main chain.
__device__ __forceinline__ void wide_forward_one(const EmbFwdArgs &a, const int b, const int lane, const bool valid) {
    const int l8 = lane & 7, gbase = lane & ~7;
    const bool early = a.F <= 32;
    float sumW = 0.f;
    bool wbad = false;
    for (int j0 = 0; j0 < a.F; j0 += 32) {
        int64_t wid[4];
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = j0 + 8 * r + l8;
            wid[r] = a.wide_ids[(size_t)b * a.F + (f < a.F ? f : a.F - 1)];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = j0 + 8 * r + l8;
            const bool bad = wid[r] < 0 || wid[r] >= a.wide_rows;
            if (bad && f < a.F) { if (early) wbad = true; else if (valid) atomicAdd(a.err, 1); }
            if (bad) wid[r] = 0;
            w[r] = a.wide_w[wid[r]];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = a.F - j0 - 8 * r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {                                    // field order; the 8 shuffles are independent
                const float v = __shfl(w[r], gbase + j);
                if (j < n) sumW += v;
            }
        }
        if (valid && a.wide_touched && a.wide_train) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (j0 + 8 * r + l8 < a.F) a.wide_touched[wid[r]] = 1;   // LRLayer.weights.put (never cleared)
        }
    }
    sumW += a.wide_bias[0];
    if (valid && wbad) atomicAdd(a.err, 1);
    if (valid && l8 == 0) a.wide_z[b] = sumW;
}

template <int VEC, bool MULTI, bool SLOT, int MHI>
__global__ __launch_bounds__(256) void k_emb_fwd(EmbFwdArgs a) {
    EndWait end_wait(a.end_wait, a.end_val, a.bound);     // (the join with the previous step's dense / replicated update)
    StampScope stamp(a.ts);
    start_wait(a.start_wait, a.start_val, a.bound);
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a.wide_blocks && (int)blockIdx.x >= a.wide_blk0) {
        const int b = ((int)blockIdx.x - a.wide_blk0) * 32 + (threadIdx.x >> 3);       // eight lanes per sample, like the head
        wide_forward_one(a, b < a.B ? b : a.B - 1, threadIdx.x & 63, b < a.B);
        return;
    }
    if (blockIdx.x >= a.gather_blocks) {
        // ConcatLayer.forward (layer/ConcatLayer.java:30-37): dense features behind the embeddings
        const int64_t t = gt - (int64_t)a.gather_blocks * 256;
        if (t < (int64_t)a.B * a.X) {
            const int b = (int)(t / a.X), x = (int)(t % a.X);
            a.out[(size_t)b * a.ld + a.F * a.D + x] = a.dense[t];
        }
        return;
    }
    const int64_t grp = gt / a.LPR;
    const int part = (int)(gt % a.LPR);
    const int64_t nb = (int64_t)a.B * a.F;
    // row of entry p of field f (clamped, counted in *err when out of range)
    auto row_of = [&](int64_t p, int f) -> int64_t {
        if (SLOT) return (int64_t)a.slot[p];              // sharded worker: slot in the pulled-row cache
        const int64_t rb = a.row_base[f], rn = a.row_base[f + 1] - rb;
        int64_t id = a.ids[p];
        if (id < 0 || id >= rn) { if (part == 0) atomicAdd(a.err, 1); id = 0; }
        return rb + id;
    };
    // sharded worker: the slots of this rank's own keys are read where the owner-side gather left them (EmbFwdArgs.W_alt)
    auto table_of = [&](int64_t row) -> const float * {
        return (SLOT && (uint32_t)row - a.alt_lo < a.alt_hi - a.alt_lo) ? a.W_alt : a.W;
    };
    if (!MULTI) {
        const int64_t bag0 = grp * GATHER_ILP;
        if (bag0 >= nb) return;
        int64_t rows[GATHER_ILP];
#pragma unroll
        for (int j = 0; j < GATHER_ILP; ++j) {
            const int64_t bag = bag0 + j < nb ? bag0 + j : nb - 1;
            rows[j] = row_of(bag, (int)(bag % a.F));
        }
        Vec<VEC> r[GATHER_ILP];
#pragma unroll
        for (int j = 0; j < GATHER_ILP; ++j) {                                                                       // rcopy (EmbeddingField.java:73)
            const float *src = table_of(rows[j]) + (size_t)rows[j] * a.D + part * VEC;
            r[j] = (a.nt & 1) ? Vec<VEC>::load_nt(src) : Vec<VEC>::load(src);
        }
#pragma unroll
        for (int j = 0; j < GATHER_ILP; ++j) {
            const int64_t bag = bag0 + j;
            if (bag < nb) {
                if (a.act == PS_ACT_RELU) { VFOR(i) r[j].at(i) = r[j].get(i) > 0.f ? r[j].get(i) : 0.f; }   // Relu.java:7-12
                float *dst = a.out + (size_t)(bag / a.F) * a.ld + (size_t)(bag % a.F) * a.D + part * VEC;
                if (a.nt & 2) r[j].store_nt(dst); else r[j].store(dst);
                if (part == 0 && a.key_out) a.key_out[bag] = (uint32_t)rows[j];
            }
        }
        return;
    }
    // a.order: bags FIELD-major and the workgroups of one XCD on one contiguous eighth of them -- an XCD's L2 then holds the hot rows of
    // ~F/8 fields instead of every field's (the rows of a field are shared by its bags, not by a sample's)
    int64_t bag = grp;
    if (a.order) {
        const unsigned int vb = (a.order & 1) ? (blockIdx.x & 7u) * ((unsigned)a.gather_blocks >> 3) + (blockIdx.x >> 3) : blockIdx.x;
        const int64_t bt = ((int64_t)vb * 256 + threadIdx.x) / a.LPR;
        bag = (a.order & 2) ? (bt < nb ? (bt % a.B) * a.F + bt / a.B : nb) : bt;
    }
    if (bag >= nb) return;
    const int b = (int)(bag / a.F), f = (int)(bag % a.F);
    const int64_t p0 = a.offsets[bag], p1 = a.offsets[bag + 1];
    Vec<VEC> s = Vec<VEC>::zero();
    constexpr int MH = MHI > 0 ? MHI : 1;
    if (MHI > 0) {
        // The lane group (LPR lanes, inside one wave) fetches LPR consecutive ids of its bag with ONE coalesced
        // load, turns them into rows, and hands them round by shuffle: the id -> row -> data chain is paid once
        // per LPR entries, MHI row loads are in flight per group, and the sort keys go out LPR at a time.
        const int lane = threadIdx.x & 63, gbase = lane - part;
        for (int64_t q = p0; q < p1; q += a.LPR) {
            const int64_t qi = q + part;
            const uint32_t myrow = (uint32_t)row_of(qi < p1 ? qi : p1 - 1, f);
            if (qi < p1 && a.key_out) {
                a.key_out[qi] = myrow;
                if (a.ent_bag) a.ent_bag[qi] = (uint32_t)bag;
            }
            const int cnt = (int)(p1 - q < a.LPR ? p1 - q : a.LPR);
            for (int j0 = 0; j0 < cnt; j0 += MH) {
                uint32_t rows[MH];
#pragma unroll
                for (int j = 0; j < MH; ++j) rows[j] = (uint32_t)__shfl((int)myrow, gbase + (j0 + j < cnt ? j0 + j : cnt - 1));
                Vec<VEC> r[MH];
#pragma unroll
                for (int j = 0; j < MH; ++j) {
                    const float *src = table_of(rows[j]) + (size_t)rows[j] * a.D + part * VEC;
                    r[j] = (a.nt & 1) ? Vec<VEC>::load_nt(src) : Vec<VEC>::load(src);
                }
#pragma unroll
                for (int j = 0; j < MH; ++j) {
                    if (j0 + j < cnt) {
                        if (q + j0 + j == p0) s = r[j];
                        else { VFOR(i) s.at(i) = r[j].get(i) + s.at(i); }      // sum pooling, strictly in bag order
                    }
                }
            }
        }
        if (a.act == PS_ACT_RELU) { VFOR(i) s.at(i) = s.get(i) > 0.f ? s.get(i) : 0.f; }
        s.store(a.out + (size_t)b * a.ld + (size_t)f * a.D + part * VEC);
        return;
    }
    for (int64_t q = p0; q < p1; q += GATHER_ILP_MH) {
        int64_t rows[GATHER_ILP_MH];
#pragma unroll
        for (int j = 0; j < GATHER_ILP_MH; ++j) rows[j] = row_of(q + j < p1 ? q + j : p1 - 1, f);
        Vec<VEC> r[GATHER_ILP_MH];
#pragma unroll
        for (int j = 0; j < GATHER_ILP_MH; ++j) r[j] = Vec<VEC>::load(table_of(rows[j]) + (size_t)rows[j] * a.D + part * VEC);
#pragma unroll
        for (int j = 0; j < GATHER_ILP_MH; ++j) {
            if (q + j < p1) {
                if (q + j == p0) s = r[j];
                else { VFOR(i) s.at(i) = r[j].get(i) + s.at(i); }      // sum pooling, strictly in bag order
                if (part == 0 && a.key_out) {
                    a.key_out[q + j] = (uint32_t)rows[j];
                    if (a.ent_bag) a.ent_bag[q + j] = (uint32_t)bag;
                }
            }
        }
    }
    if (a.act == PS_ACT_RELU) { VFOR(i) s.at(i) = s.get(i) > 0.f ? s.get(i) : 0.f; }
    s.store(a.out + (size_t)b * a.ld + (size_t)f * a.D + part * VEC);
}

#ifdef PS_HEAD_TIMING
__device__ unsigned long long g_head_t[256 * 8];
extern "C" int ps_dbg_head_timing(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_head_t), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
#define HEAD_T(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_head_t[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define HEAD_T(k) do { } while (0)
#endif
#include "kernels_head.inc"

__global__ __launch_bounds__(256) void k_head(HeadArgs a) {
    const int b = blockIdx.x * 32 + (threadIdx.x >> 3);
    (void)head_one(a, b < a.B ? b : a.B - 1, threadIdx.x & 63, b < a.B);
}

// FcLayer.backward of the out = 1 layer (kernels_head.inc last_bwd_rows), optionally with the head of the same rows in front

template <bool HEAD>
__global__ __launch_bounds__(256) void k_last_bwd(LastBwdArgs a, HeadArgs h) {
    __shared__ float dsh[HEAD ? HEAD_ROWS_MAX : 1];
    if (a.prio) __builtin_amdgcn_s_setprio(3);       // main-chain kernel of the fused step (see k_gemm_nt)
    StampScope stamp(a.ts);
    if (!HEAD && a.skip && *a.skip) return;
    last_bwd_rows<HEAD>(a, h, dsh, threadIdx.x, blockIdx.x);
}

// loss = sum(terms)/B, gbar = rowMeans(delta) ; sets the skip flag (model/DNN.java:58-63)
__global__ __launch_bounds__(1024) void k_loss_reduce(const float *terms, const float *dlast, int ldd, int B,
                                                      float *loss_out, float *gbar_out, int *skip, int force_no_skip) {
    __shared__ float s1[1024], s2[1024];
    const int tid = threadIdx.x;
    float a = 0.f, g = 0.f;
    for (int i = tid; i < B; i += 1024) { a += terms[i]; g += dlast[(size_t)i * ldd]; }
    s1[tid] = a; s2[tid] = g;
    __syncthreads();
    for (int off = 512; off; off >>= 1) {
        if (tid < off) { s1[tid] += s1[tid + off]; s2[tid] += s2[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float loss = s1[0] / B;
        loss_out[0] = loss;
        gbar_out[0] = s2[0] / (float)B;
        skip[0] = (!force_no_skip && (loss <= 0.01f || loss != loss)) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// updaters (shared by sparse rows and dense tensors), one element
// ---------------------------------------------------------------------------
// IEEE round-to-nearest sqrt and divide.  NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS
// the HIP header maps it to __ocml_native_sqrt_f32 (v_sqrt_f32, ~1 ulp) -- measured 1-ulp
// mismatches against the oracle.  llvm.sqrt.f32 / fdiv are correctly rounded under hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt (= Java's (float)Math.sqrt((double)x) and '/').
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float div_rn(float x, float y) { return x / y; }

__device__ __forceinline__ void adam_elem(const UpdParams &u, float g, float &w, float &M, float &V) {
    // update/AdamUpdater.java:61-69, op for op
    float m = g * u.c1;
    float mo = M * u.beta1;
    m = mo + m;
    float v = g * g;
    v = v * u.c2;
    float vo = V * u.beta2;
    v = vo + v;
    M = m; V = v;
    const float mm = div_rn(m, u.c1);
    const float vv = div_rn(v, u.c2);
    const float den = sqrt_rn(vv) + u.eps;
    float q = div_rn(mm, den);
    q = q * u.neg_alfa;
    w = q + w;
}

__device__ __forceinline__ void ftrl_elem(const UpdParams &u, float g, float &w, float &z, float &n) {
    // update/FtrlUpdater.java:64-74 (w from the OLD z,n; then z,n with the NEW w)
    if (fabsf(z) <= u.l1) {
        w = 0.f;
    } else {
        const float sign = z >= 0.f ? 1.f : -1.f;
        const float den = div_rn(u.l2 + (u.beta + sqrt_rn(n)), u.alfa);
        w = div_rn(-(z - sign * u.l1), den);
    }
    const float g2 = g * g;                                   // pow(dw,2): exact product, one rounding
    const float s = sqrt_rn(n + g2) - sqrt_rn(div_rn(n, u.alfa));
    const float t = g - s * w;
    z = t + z;
    n = g2 + n;
}

// ---------------------------------------------------------------------------
// long-run partials: tile c of CH consecutive sorted entries computes the (at
// most two) CH-chunks of long segments that START inside it
// ---------------------------------------------------------------------------
// BAG is a template parameter on purpose: a run-time "a.ent_bag ? load : p" inside the unrolled
// batches makes hipcc branch around every load and wait for each one separately (the guide's
// ".s-level trap (c)"): measured 30 us of serialized round trips per kernel before the hoist.
template <int VEC, bool BAG>
__device__ __forceinline__ Vec<VEC> load_g(const EmbBwdArgs &a, uint32_t p, int part) {
    // masked per-sample gradient of entry p: relu'(A)*delta slice
    // (EmbeddingField.java:91; the relu' mask was applied by the producing GEMM epilogue)
    uint32_t bag = p;
    if (BAG) bag = a.ent_bag[p];
    const uint32_t b = bag / (uint32_t)a.F, f = bag % (uint32_t)a.F;
    return Vec<VEC>::load(a.delta + (size_t)b * a.ldd + (size_t)f * a.D + part * VEC);
}

// acc (+)= g[s] + g[s+1] + ... in index order.  The adds are a strict chain (the reference
// order), but the loads are not: entries are fetched PS_EMB_ILP at a time -- 16 independent
// index loads, then 16 independent row loads -- so a run costs ~2 memory latencies per 16
// entries instead of 2 per entry (measured 30 us -> the latency chain was the whole kernel).
// Rows in flight per lane group (round 5: 32 -> 8 in the per-key reduce, 16 in the chunk partials).  Only the chunked order (multi-hot
// batches) gets here with more than 16 entries, and there the kernels are bound by how many keys are in flight, not by one key's chain:
// 32 rows per lane group cost k_emb_reduce_update 168 VGPRs (3 waves per SIMD), 16 cost 101 (4), 8 cost 71 (7).  At configs[4]'s shape
// 32 / 16 / 8: per-key reduce + Ftrl 71 / 64 / 56 us, chunk partials 36 / 28 / 30 us, the step 0.348 / 0.337 / 0.334 ms.  Same adds in the
// same order whatever the batch.
#ifndef PS_EMB_ILP
#define PS_EMB_ILP 8
#endif
#ifndef PS_EMB_ILP_PARTIALS
#define PS_EMB_ILP_PARTIALS 16
#endif
template <int VEC, bool BAG, int ILP>
__device__ __forceinline__ void run_sum(const EmbBwdArgs &a, uint32_t s, uint32_t e, int part, Vec<VEC> &acc, bool have) {
    for (uint32_t k = s; k < e; k += ILP) {
        uint32_t ent[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) ent[j] = a.sorted_ent[k + j < e ? k + j : e - 1];
        Vec<VEC> g[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) g[j] = load_g<VEC, BAG>(a, ent[j], part);
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            if (k + j < e) {
                if (have) { VFOR(i) acc.at(i) = g[j].get(i) + acc.at(i); }
                else { acc = g[j]; have = true; }
            }
        }
    }
}

template <int VEC, bool BAG, int ILP>
__device__ __forceinline__ Vec<VEC> chunk_sum(const EmbBwdArgs &a, uint32_t s, uint32_t e, int part) {
    Vec<VEC> acc = Vec<VEC>::zero();
    run_sum<VEC, BAG, ILP>(a, s, e, part, acc, false);     // first touch: put :91; then addi :94
    return acc;
}

// A key seen n <= NL times: exactly NL index loads and NL row loads (clamped duplicates beyond n),
// all independent, then both passes from registers.  NL is sized to n (1 / 4 / 16): issuing 16
// clamped loads for the typical n = 1 key made the kernel TA-bound on redundant requests.
template <int VEC, bool BAG, int NL>
__device__ __forceinline__ Vec<VEC> small_key(const EmbBwdArgs &a, uint32_t s0, uint32_t n, int part) {
    uint32_t ent[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) ent[j] = a.sorted_ent[s0 + ((uint32_t)j < n ? j : n - 1)];
    Vec<VEC> g[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) g[j] = load_g<VEC, BAG>(a, ent[j], part);
    Vec<VEC> S = g[0];                                              // put :91
#pragma unroll
    for (int j = 1; j < NL; ++j)
        if ((uint32_t)j < n) { VFOR(i) S.at(i) = g[j].get(i) + S.at(i); }       // addi :94, batch order
    VFOR(i) S.at(i) = div_rn(S.get(i), (float)n);                   // divi(N) :100  (pass 1)
    if (a.grad_mode == PS_GRAD_COMPAT) {
#pragma unroll
        for (int j = 0; j < NL; ++j)                                // pass 2: every g_k again
            if ((uint32_t)j < n) { VFOR(i) S.at(i) = g[j].get(i) + S.at(i); }
        VFOR(i) S.at(i) = div_rn(S.get(i), (float)(2 * n));         // divi(2n); then x2 (sum.addi self), /2 (cnt) exact
    }
    return S;
}

// the updater of `row` when the fields' updaters differ (FieldUpd, kernels_emb.h); callers test fu.ngroups > 1 first
template <class ARGS>
__device__ __forceinline__ UpdParams row_upd(const ARGS &a, uint32_t row) {
    int f = 0;
    while (f + 1 < a.fu.F && (int64_t)row >= a.fu.row_base[f + 1]) ++f;
    int g = a.fu.grp[f];
    for (int i = 0; i < a.fu.nover; ++i)          // exact-key rows win over their field (store/KVStore.java:242)
        if (a.fu.over_row[i] == row) g = a.fu.over_grp[i];
    return g ? a.fu.alt[g - 1] : a.upd;
}
// XCD-affine work order of the embedding backward.  The sorted entries run field by field, and an entry's delta row is the 64 bytes
// of ITS field in its sample's row of dx: a contiguous eighth of the sorted entries touches the dx columns of ~F/8 + 1 fields -- 1-2 MB
// of lines, resident in one XCD's 4 MB L2 -- where a round-robin deal makes every XCD's L2 fetch (and re-fetch, beside the streaming
// W / state rows) all of dx, each 128-byte line once per half.  Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md: a matter
// of speed only): the workgroups of one XCD take one contiguous eighth of the tiles (here) / of the keys by entry count
// (k_emb_reduce_update).  Grids are multiples of 8 when a.xcd is set.
__device__ __forceinline__ unsigned int emb_vblock(int xcd) {
    return xcd ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
}

// XCD-affine order BY FIELD PAIR (a.xcd == 3, round 6).  An entry's delta row is the D floats of ITS field in its sample's row of dx; at
// D = 16 two neighbouring fields share every 128-byte line.  Dealt by eighths of the keys (a.xcd == 1) the two halves of a line are consumed
// under two L2s wherever an eighth ends inside a pair, and the long-key role took its runs in list order, any XCD: every delta line was
// fetched about twice (PMC: 24.1 MB read for 16.6 algorithmic, profiles/r05_pmc_traffic.txt).  Now the runs -- short and long -- of fields
// 2x, 2x + 1, 2x + 16, 2x + 17, ... belong to the workgroups of XCD x (block b runs on XCD b % 8: observed, a matter of speed only), from the
// table the field sort leaves per field: [first run, runs, first long run, long runs].  t-th run (which = 0) / long run (which = 1) of XCD x:
__device__ __forceinline__ bool emb_xcd_pick(const uint32_t *__restrict__ ftab, int F, int x, int which, uint32_t t, uint32_t &idx) {
    for (int f0 = 2 * x; f0 < F; f0 += 16) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int f = f0 + e;
            if (f >= F) break;
            const uint32_t base = ftab[4 * f + 2 * which], cnt = ftab[4 * f + 2 * which + 1];
            if (t < cnt) { idx = base + t; return true; }
            t -= cnt;
        }
    }
    return false;
}

// the first key whose run starts at or behind entry p  (plain arguments: a lambda that captures the kernel's argument struct by
// reference makes hipcc copy the whole struct to scratch -- 856 bytes per lane, the single-hot update 135 us instead of 17)
__device__ __forceinline__ int64_t emb_key_at(const uint32_t *__restrict__ seg_id, const uint32_t *__restrict__ seg_start, int64_t nnz, int64_t nseg, int64_t p) {
    if (p <= 0) return 0;
    if (p >= nnz) return nseg;
    const uint32_t u = seg_id[p];
    return (int64_t)u + (seg_start[u] == (uint32_t)p ? 0 : 1);
}

template <int VEC, bool BAG>
__global__ __launch_bounds__(256) void k_emb_partials(EmbBwdArgs a) {
    StampScope stamp(a.ts_partials);
    if (a.flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.skip && *a.skip) return;
    const int64_t gt = (int64_t)emb_vblock(a.xcd) * 256 + threadIdx.x;
    const int lane64 = (int)(gt & 63);
    const int gpw = 64 / a.LPR;                         // lane groups per wave (groups never straddle waves)
    if (lane64 / a.LPR >= gpw) return;
    const int64_t c = (gt >> 6) * gpw + lane64 / a.LPR;
    const int part = lane64 % a.LPR;
    const uint32_t CH = PS_EMB_CHUNK;
    const uint32_t t0 = (uint32_t)(c * CH);
    if ((int64_t)t0 >= a.nnz) return;
    const uint32_t t1 = (uint32_t)((int64_t)t0 + CH < a.nnz ? t0 + CH : a.nnz) - 1;
    const uint32_t u0 = a.seg_id[t0];
    const uint32_t s0 = a.seg_start[u0], e0 = a.seg_start[u0 + 1];
    if (e0 - s0 > CH) {
        const uint32_t j = (t0 - s0 + CH - 1) / CH;
        const uint32_t s = s0 + j * CH;
        if (s <= t1 && s < e0) {
            const uint32_t e = s + CH < e0 ? s + CH : e0;
            const size_t slot = (size_t)2 * c + (j == 0 ? 1 : 0);
            chunk_sum<VEC, BAG, PS_EMB_ILP_PARTIALS>(a, s, e, part).store(a.partials + slot * a.D + part * VEC);
        }
    }
    const uint32_t u1 = a.seg_id[t1];
    if (u1 != u0) {
        const uint32_t s1 = a.seg_start[u1], e1 = a.seg_start[u1 + 1];
        if (e1 - s1 > CH) {
            const uint32_t e = s1 + CH;  // < e1
            chunk_sum<VEC, BAG, PS_EMB_ILP_PARTIALS>(a, s1, e, part).store(a.partials + ((size_t)2 * c + 1) * a.D + part * VEC);
        }
    }
}

// Second level for very long runs (> PS_EMB_SUPER_MIN chunks): the lane group whose tile holds the start of
// chunk j with j % PS_EMB_SUPER == 0 folds the chunk partials j .. j+31 (in chunk order) into one super partial,
// stored under the same slot index in partials2.  Same tile walk as k_emb_partials; almost every group exits.
template <int VEC>
__global__ __launch_bounds__(256) void k_emb_super(EmbBwdArgs a) {
    StampScope stamp(a.ts_super);
    if (a.skip && *a.skip) return;
    const int64_t gt = (int64_t)emb_vblock(a.xcd) * 256 + threadIdx.x;
    const int lane64 = (int)(gt & 63);
    const int gpw = 64 / a.LPR;
    if (lane64 / a.LPR >= gpw) return;
    const int64_t c = (gt >> 6) * gpw + lane64 / a.LPR;
    const int part = lane64 % a.LPR;
    const uint32_t CH = PS_EMB_CHUNK;
    const uint32_t t0 = (uint32_t)(c * CH);
    if ((int64_t)t0 >= a.nnz) return;
    const uint32_t t1 = (uint32_t)((int64_t)t0 + CH < a.nnz ? t0 + CH : a.nnz) - 1;
    auto fold = [&](uint32_t s0, uint32_t e0, uint32_t j, size_t out_slot) {
        const uint32_t nch = (e0 - s0 + CH - 1) / CH;
        const uint32_t j1 = j + PS_EMB_SUPER < nch ? j + PS_EMB_SUPER : nch;
        Vec<VEC> p[PS_EMB_SUPER];
#pragma unroll
        for (int k = 0; k < PS_EMB_SUPER; ++k) {
            const uint32_t jj = j + k < j1 ? j + k : j1 - 1;
            const uint32_t s = s0 + jj * CH;
            p[k] = Vec<VEC>::load(a.partials + ((size_t)2 * (s / CH) + (jj == 0 ? 1 : 0)) * a.D + part * VEC);
        }
        Vec<VEC> acc = p[0];
#pragma unroll
        for (int k = 1; k < PS_EMB_SUPER; ++k)
            if (j + k < j1) { VFOR(i) acc.at(i) = p[k].get(i) + acc.at(i); }
        acc.store(a.partials2 + out_slot * a.D + part * VEC);
    };
    const uint32_t u0 = a.seg_id[t0];
    const uint32_t s0 = a.seg_start[u0], e0 = a.seg_start[u0 + 1];
    if ((e0 - s0 + CH - 1) / CH > PS_EMB_SUPER_MIN) {
        const uint32_t j = (t0 - s0 + CH - 1) / CH;
        const uint32_t s = s0 + j * CH;
        if (s <= t1 && s < e0 && j % PS_EMB_SUPER == 0) fold(s0, e0, j, (size_t)2 * c + (j == 0 ? 1 : 0));
    }
    const uint32_t u1 = a.seg_id[t1];
    if (u1 != u0) {
        const uint32_t s1 = a.seg_start[u1], e1 = a.seg_start[u1 + 1];
        if ((e1 - s1 + CH - 1) / CH > PS_EMB_SUPER_MIN) fold(s1, e1, 0, (size_t)2 * c + 1);
    }
}

// The same super partials from the sort's LIST of the runs above PS_EMB_CHUNK * PS_EMB_SUPER_MIN entries (a.long_list: run id, first
// entry, end; *a.nlong of them) instead of a walk over every tile: a fixed small grid, workgroup i takes runs i, i + grid, ..., its lane
// groups the run's super groups.  (The tile walk is 6 k workgroups of which a few dozen fold anything: 19 us on the multi-hot step's
// main chain between the chunk partials and the per-key reduce.)
template <int VEC>
__global__ __launch_bounds__(256) void k_emb_super_list(EmbBwdArgs a) {
    StampScope stamp(a.ts_super);
    if (a.skip && *a.skip) return;
    const int lane64 = (int)(threadIdx.x & 63);
    const int gpw = 64 / a.LPR;
    if (lane64 / a.LPR >= gpw) return;
    const int part = lane64 % a.LPR;
    const uint32_t grp = (threadIdx.x >> 6) * gpw + lane64 / a.LPR, ngrp = 4 * gpw;
    const uint32_t CH = PS_EMB_CHUNK;
    const uint32_t nl = *a.nlong;
    for (uint32_t i = blockIdx.x; i < nl; i += gridDim.x) {
        const uint32_t *ll = a.long_list + 3 * (size_t)i;
        const uint32_t s0 = ll[1], e0 = ll[2];
        const uint32_t nch = (e0 - s0 + CH - 1) / CH;
        if (nch <= PS_EMB_SUPER_MIN) continue;
        for (uint32_t j = grp * PS_EMB_SUPER; j < nch; j += ngrp * PS_EMB_SUPER) {
            const uint32_t j1 = j + PS_EMB_SUPER < nch ? j + PS_EMB_SUPER : nch;
            Vec<VEC> p[PS_EMB_SUPER];
#pragma unroll
            for (int k = 0; k < PS_EMB_SUPER; ++k) {
                const uint32_t jj = j + k < j1 ? j + k : j1 - 1;
                const uint32_t sc = s0 + jj * CH;
                p[k] = Vec<VEC>::load(a.partials + ((size_t)2 * (sc / CH) + (jj == 0 ? 1 : 0)) * a.D + part * VEC);
            }
            Vec<VEC> acc = p[0];
#pragma unroll
            for (int k = 1; k < PS_EMB_SUPER; ++k)
                if (j + k < j1) { VFOR(i2) acc.at(i2) = p[k].get(i2) + acc.at(i2); }
            const uint32_t sj = s0 + j * CH;
            acc.store(a.partials2 + ((size_t)2 * (sj / CH) + (j == 0 ? 1 : 0)) * a.D + part * VEC);
        }
    }
}

// ---------------------------------------------------------------------------
// per-key reduce (+ double-backward factor) (+ fused updater)
// ---------------------------------------------------------------------------
// what KVStore.sum + KVStore.update(Map) do with one key's effective gradient S (store/KVStore.java:192-268):
// hand it out (split form / sharded push) and/or run the updater on the row in place.  Called by the LPR lanes
// of the key's lane group (all of them: the Ftrl skip test shuffles element 0 from part 0).
template <int VEC>
__device__ __forceinline__ void update_key(const EmbBwdArgs &a, const UpdParams &upd, uint32_t row, Vec<VEC> &S, int part) {
    float *wp = a.W + (size_t)row * a.D + part * VEC;
    float *sp = a.state + (size_t)row * 2 * a.D + part * VEC;
    if (upd.kind == PS_UPD_FTRL) {
        // FtrlUpdater.java:64-74 computes w from the OLD z, n: the row's weights are written, never read (a sixth of the
        // bytes of a key); :52 skips the whole key when dw[0] == 0 (element 0 lives in part 0)
        Vec<VEC> w = Vec<VEC>::zero(), s1 = Vec<VEC>::load(sp), s2 = Vec<VEC>::load(sp + a.D);
        const int lane = threadIdx.x & 63;
        const float g0 = __shfl(S.get(0), lane - part);
        if (g0 == 0.f) return;
        VFOR(i) ftrl_elem(upd, S.get(i), w.at(i), s1.at(i), s2.at(i));
        w.store(wp); s1.store(sp); s2.store(sp + a.D);
        return;
    }
    Vec<VEC> w = Vec<VEC>::load(wp);
    if (upd.kind == PS_UPD_SIMPLE) {
        VFOR(i) w.at(i) = (S.get(i) * -upd.eta) + w.get(i);      // update/SimpleUpdater.java:20-22
        w.store(wp);
        return;
    }
    Vec<VEC> s1 = Vec<VEC>::load(sp), s2 = Vec<VEC>::load(sp + a.D);
    if (upd.kind == PS_UPD_ADAM) {
        VFOR(i) adam_elem(upd, S.get(i), w.at(i), s1.at(i), s2.at(i));
    } else {
        // FtrlUpdater.java:52: skip the whole key when dw[0] == 0; element 0 lives in part 0
        const int lane = threadIdx.x & 63;
        const float g0 = __shfl(S.get(0), lane - part);
        if (g0 == 0.f) return;
        VFOR(i) ftrl_elem(upd, S.get(i), w.at(i), s1.at(i), s2.at(i));
    }
    w.store(wp); s1.store(sp); s2.store(sp + a.D);
}
template <int VEC>
__device__ __forceinline__ void finish_key(const EmbBwdArgs &a, uint32_t u, uint32_t row, uint32_t n, Vec<VEC> &S, int part) {
    if (a.grads_out) {
        S.store(a.grads_out + (size_t)u * a.D + part * VEC);
        if (part == 0) { a.uniq_row[u] = row; if (a.uniq_cnt) a.uniq_cnt[u] = n; }
    }
    if (!a.apply) return;
    if (a.fu.ngroups > 1) update_key<VEC>(a, row_upd(a, row), row, S, part);      // (a lane group's lanes share the row)
    else update_key<VEC>(a, a.upd, row, S, part);
}
// (Round 5, again, for the sequential order's short-key role only -- one key per lane group, so the row's W / state behind its gradient
//  are a fourth dependent round trip: 105 -> 141 VGPRs, 4 -> 3 waves per SIMD, the embedding update 16 -> 21.9 us, the step 0.1332 ->
//  0.1364 ms.  tools/ab_knobs.sh with seq_ablate says what bounds that launch in the step: without its long-key role 0.1339 ms, without its
//  short-key role 0.1304 -- the 2048 short-key workgroups behind 1664 long-key ones, four waves per SIMD.)
// (Round 4, measured and taken out again: requesting the row's W / state as soon as the row is known -- together with the key's
// delta rows instead of behind the store of its gradient -- changes nothing for single-hot batches (0.1329 against 0.1331 ms / step,
// A/B on one box: the long-key role decides that kernel) and costs the multi-hot step 35 us (0.390 -> 0.425 ms: twelve more
// live VGPRs in a role that already holds 32 rows in flight).)

// The REFERENCE order for a key seen n > PS_EMB_CHUNK times (layer/EmbeddingField.java:86-104: one addi per sample,
// strictly in batch order; then the second pass of App. A.6).  A strict f32 chain of n (compat: 2n) dependent adds
// cannot be split over lanes -- but everything around it can:
//   * the WORKGROUP of the 32-entry tile in which the key's run starts owns the key (at most one run above 32
//     entries can start in a tile); every other workgroup of the long-key range exits after three loads;
//   * waves 1..3 are loaders: index load -> delta row, SEQ_ILP entries per lane group in flight, one batch ahead of
//     the fold, rows parked in LDS TRANSPOSED ([component][entry]);
//   * wave 0 folds: lane d owns component d and walks its LDS row four entries per ds_read_b128 -- five
//     instructions per four entries on the only serial path of the kernel.
// One barrier per batch, two LDS buffers.  (A first version with one wave doing both, fetching rows as float4 and
// folding [entry][part], spent ~60 cycles per entry in its own instruction stream: 114 us for the 2240-entry keys of
// configs[1]; this one ~15.  tools/seq_ablate.py: with the fold switched off the loaders alone still take 28 of the 34 us
// -- 24 serial iterations of ~1.2 us, each bounded by the latency of the delta rows, freshly written by the GEMM of
// another XCD -- and with the loads switched off the fold alone takes the same; a DPP-broadcast variant that cut the
// fold's LDS read instructions fourfold measured SLOWER (46 us).)
#define SEQ_ILP 4
#define SEQ_TILE PS_EMB_SEQ_TILE  // SEQ mode: keys above this many entries go to a long-key workgroup (at most one such run
                                  // can start in a SEQ_TILE-entry tile); 16 keeps the short role at 64 row registers
#define SEQ_LONG_GRID 1024        // long-key workgroups when the sort handed over a list of the long runs (round 4: 512 -> 2048, ~1500 runs above 16 entries
                                  // in a configs[1] batch: one run per workgroup instead of three in a row; 0.1333 -> 0.1325 ms / step.  Round 6: 2048 -> 1024.
                                  // The launch is bound by how fast its workgroups are PLACED (tools/emb_timing.sh: 3712 workgroups, starts spread over
                                  // 17.6 of its 19.1 us, the short role's first workgroup placed at 12 us behind 2048 long-role ones of which 500 find no
                                  // run); with the list dealt by eighths since round 6 two runs in a row cost less than the 1024 placements: 0.1341 ->
                                  // 0.1323 ms / step, flat from 512 to 1024, the clamped id law 0.1401 either way: profiles/r06_emb_list_role.txt)
#define SUPER_LDS_FLOATS 4096     // the chunked order's super role: super partials of one round (16 KB)
#define SEQ_LDS_FLOATS 4096       // per buffer: D * (3 * (64 / LPR) * SEQ_ILP + 4) <= 768 * VEC + 4 * D
__device__ __forceinline__ uint32_t div_by(uint32_t x, uint32_t d, uint32_t magic, uint32_t &rem) {
    uint32_t q = __umulhi(x, magic);            // magic = floor(2^32 / d): q is the quotient or one less
    uint32_t r = x - q * d;
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// One long run (segment u, entries [s0, e0)), by the whole workgroup; every wave passes the same number of barriers.
template <int VEC, bool BAG>
__device__ __forceinline__ void long_key_run(const EmbBwdArgs &a, float *lds /* [2][SEQ_LDS_FLOATS] */, uint32_t u, uint32_t s0, uint32_t e0) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = e0 - s0;
    const int D = a.D, G = 64 / a.LPR;
    const uint32_t NBW = (uint32_t)G * SEQ_ILP, NB = 3 * NBW;   // entries per loader wave / per batch
    const uint32_t LDE = NB + 4;                                // LDS row stride: 16-B aligned rows, <= 2-way write conflicts
    const uint32_t nbatch = (n + NB - 1) / NB;
    const int npass = a.grad_mode == PS_GRAD_COMPAT ? 2 : 1;
    const uint32_t total = nbatch * (uint32_t)npass;            // both passes as one stream: the pipeline never drains
    const uint32_t fmagic = (uint32_t)(0x100000000ull / (uint32_t)a.F);
    if (w > 0) {
        // ---- loaders ----
        const int grp = lane / a.LPR, part = lane % a.LPR;
        const bool act = grp < G;                               // D = 10: lanes 60..63 idle
        const uint32_t lbase = (uint32_t)(w - 1) * NBW + (uint32_t)(act ? grp : 0);
        // TWO register sets: batch j lives in set j & 1, so the rows of two batches are in flight at any time and a batch
        // has two iterations to arrive.  (One set: every iteration waited a full row latency, ~1.2 us x 24 iterations
        // for the 2240-entry keys of configs[1] -- which WAS the embedding update: 29 us with the long-key role, 10.5
        // without, ps_tune_set("seq_ablate", 2).)  No load sits under a condition: past the end the clamped last batch
        // is fetched and parked into the buffer nobody reads again, so the compiler's wait counts stay exact.
        uint32_t entA[SEQ_ILP], entB[SEQ_ILP];
        Vec<VEC> rA[SEQ_ILP], rB[SEQ_ILP];
        auto load_idx = [&](uint32_t (&ent)[SEQ_ILP], uint32_t t) {
            const uint32_t tt = t < total ? t : total - 1;
            const uint32_t base = s0 + (tt % nbatch) * NB + lbase;
#pragma unroll
            for (int i = 0; i < SEQ_ILP; ++i) {
                const uint32_t p = base + (uint32_t)i * G;
                ent[i] = a.sorted_ent[p < e0 ? p : e0 - 1];
            }
        };
        auto load_rows = [&](Vec<VEC> (&r)[SEQ_ILP], const uint32_t (&ent)[SEQ_ILP]) {
#pragma unroll
            for (int i = 0; i < SEQ_ILP; ++i) {
                uint32_t bag = ent[i];
                if (BAG) bag = a.ent_bag[bag];
                uint32_t f = 0, b = bag;
                if (a.F > 1) b = div_by(bag, (uint32_t)a.F, fmagic, f);      // F = 1: the magic would be 2^32
                r[i] = Vec<VEC>::load(a.delta + (size_t)b * a.ldd + (size_t)f * D + (act ? part : 0) * VEC);
            }
        };
        auto park = [&](uint32_t t, const Vec<VEC> (&r)[SEQ_ILP]) {
            float *buf = lds + (t & 1u) * SEQ_LDS_FLOATS;
            if (act) {
#pragma unroll
                for (int i = 0; i < SEQ_ILP; ++i)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) buf[(uint32_t)(part * VEC + k) * LDE + lbase + (uint32_t)i * G] = r[i].get(k);
            }
        };
        load_idx(entA, 0); load_idx(entB, 1);
        load_rows(rA, entA); load_idx(entA, 2);                 // batch 0; then the indices of batch 2
        load_rows(rB, entB); load_idx(entB, 3);                 // batch 1; batch 3
        park(0, rA);
        load_rows(rA, entA); load_idx(entA, 4);                 // batch 2; batch 4
        __syncthreads();
        uint32_t t = 0;
        for (; t + 2 <= total; t += 2) {
            park(t + 1, rB);                                    // rows of batch t+1 (issued two iterations ago)
            load_rows(rB, entB); load_idx(entB, t + 5);         // batch t+3 from the indices fetched two iterations ago
            __syncthreads();
            park(t + 2, rA);
            load_rows(rA, entA); load_idx(entA, t + 6);         // batch t+4
            __syncthreads();
        }
        if (t < total) {
            park(t + 1, rB);
            __syncthreads();
        }
        return;                                                 // (the fold wave's last barrier is this loop's last one)
    }
    // ---- wave 0: the strict chain.  lane d owns components d, d + 64, ... ----
    constexpr int CPL = VEC == 4 ? 4 : 1;                       // D <= 256 (VEC 4) or <= 64 (VEC 1)
    float S[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) S[q] = 0.f;                   // 0 + g_1 = g_1 exactly (put :91)
    // the key's row, its W and its updater state are fetched NOW, while the loaders fill the first batch: at the end they
    // were two dependent round trips (row id -> values) on the only serial path of the kernel
    const uint32_t row = a.sorted_key[s0];
    float wv0[CPL], s10[CPL], s20[CPL];
    UpdParams U = a.upd;
    if (a.fu.ngroups > 1) U = row_upd(a, row);
    {
        const bool upd = a.apply != 0, st = upd && U.kind != PS_UPD_SIMPLE;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int d = lane + 64 * q, dc = d < D ? d : 0;
            wv0[q] = (upd && U.kind != PS_UPD_FTRL) ? a.W[(size_t)row * D + dc] : 0.f;     // (Ftrl writes w, never reads it)
            s10[q] = st ? a.state[(size_t)row * 2 * D + dc] : 0.f;
            s20[q] = st ? a.state[(size_t)row * 2 * D + D + dc] : 0.f;
        }
    }
    __syncthreads();
    for (uint32_t t = 0; t < total; ++t) {
        const uint32_t b = t % nbatch;
        const uint32_t cnt = n - b * NB < NB ? n - b * NB : NB;
        const float *buf = lds + (t & 1u) * SEQ_LDS_FLOATS;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int d = lane + 64 * q;
            if (d < D && !(a.ablate & 1)) {
                const float *row = buf + (uint32_t)d * LDE;
                float acc = S[q];
                uint32_t j = 0;
                for (; j + 32 <= cnt; j += 32) {               // 8 reads in flight, then 32 dependent adds
                    float4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(row + j + 4 * k);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc = v[k].x + acc; acc = v[k].y + acc; acc = v[k].z + acc; acc = v[k].w + acc; }   // addi :94
                }
                for (; j + 4 <= cnt; j += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(row + j);
                    acc = v.x + acc; acc = v.y + acc; acc = v.z + acc; acc = v.w + acc;
                }
                for (; j < cnt; ++j) acc = row[j] + acc;
                if (b == nbatch - 1) acc = div_rn(acc, (float)((t >= nbatch ? 2u : 1u) * n));     // divi(n) :100 ; pass 2: divi(2n)
                S[q] = acc;
            }
        }
        __syncthreads();
    }
    // KVStore.sum + update for this key (finish_key's arithmetic, one component per lane)
    if (a.out_slot) u = a.out_slot[a.sorted_ent[s0]];            // (see the short role)
    const float g0 = __shfl(S[0], 0);                           // FtrlUpdater.java:52 looks at dw[0]
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int d = lane + 64 * q;
        if (d >= D) continue;
        const float g = S[q];
        if (a.grads_out) {
            a.grads_out[(size_t)u * D + d] = g;
            if (d == 0) { a.uniq_row[u] = row; if (a.uniq_cnt) a.uniq_cnt[u] = n; }
        }
        if (!a.apply) continue;
        float *wp = a.W + (size_t)row * D + d;
        float wv = wv0[q];
        if (U.kind == PS_UPD_SIMPLE) { *wp = (g * -U.eta) + wv; continue; }
        float *sp = a.state + (size_t)row * 2 * D + d;
        float s1 = s10[q], s2 = s20[q];
        if (U.kind == PS_UPD_ADAM) adam_elem(U, g, wv, s1, s2);
        else { if (g0 == 0.f) continue; ftrl_elem(U, g, wv, s1, s2); }
        *wp = wv; sp[0] = s1; sp[D] = s2;
    }
}

// the short-key role of k_emb_reduce_update for ONE key (run u of the sorted entries), by the LPR lanes of a lane group
template <int VEC, bool BAG, bool SEQ>
__device__ __forceinline__ void reduce_one_key(const EmbBwdArgs &a, const int64_t u, const int part) {
    const uint32_t CH = PS_EMB_CHUNK;
    const uint32_t s0 = a.seg_start[u], e0 = a.seg_start[u + 1];
    const uint32_t n = e0 - s0;
    const uint32_t row = a.sorted_key[s0];
    Vec<VEC> S;
    if (n == 1) {
        // by far the most common key: one index, one row; g_eff = g exactly (App. A.6, n = 1)
        S = small_key<VEC, BAG, 1>(a, s0, n, part);
    } else if (n <= 4) {
        S = small_key<VEC, BAG, 4>(a, s0, n, part);
    } else if (SEQ) {
        if (n > 16) return;                                     // n > SEQ_TILE: a long-key workgroup owns this key
        S = small_key<VEC, BAG, 16>(a, s0, n, part);
    } else if (n <= PS_EMB_ILP) {
        S = small_key<VEC, BAG, PS_EMB_ILP>(a, s0, n, part);    // (one batch of loads, both passes from registers)
    } else if (n <= CH) {
        S = chunk_sum<VEC, BAG, PS_EMB_ILP>(a, s0, e0, part);
        VFOR(i) S.at(i) = div_rn(S.get(i), (float)n);
        if (a.grad_mode == PS_GRAD_COMPAT) {
            run_sum<VEC, BAG, PS_EMB_ILP>(a, s0, e0, part, S, true);
            VFOR(i) S.at(i) = div_rn(S.get(i), (float)(2 * n));
        }
    } else {
        const uint32_t nch = (n + CH - 1) / CH;
        // runs above 128 chunks were pre-folded 32 chunks at a time by k_emb_super (one lane group walking the
        // 1968 partials of a 63k-entry key WAS the kernel: ~250 us); their super partials sit in partials2
        const bool two = nch > PS_EMB_SUPER_MIN;
        if (a.super_blocks > 0 && nch > (uint32_t)a.list_min) return;      // (round 6: a workgroup of this launch's list role owns the key)
        const uint32_t step = two ? PS_EMB_SUPER : 1u;
        const float *src = two ? a.partials2 : a.partials;
        const uint32_t cnt = (nch + step - 1) / step;
        auto slot_of = [&](uint32_t j) -> size_t {
            const uint32_t s = s0 + j * CH;
            return (size_t)2 * (s / CH) + (j == 0 ? 1 : 0);
        };
        auto add_partials = [&](bool have) {                        // partials in order, 32 loads in flight
            for (uint32_t j0 = 0; j0 < cnt; j0 += PS_EMB_ILP) {
                Vec<VEC> p[PS_EMB_ILP];
#pragma unroll
                for (int j = 0; j < PS_EMB_ILP; ++j)
                    p[j] = Vec<VEC>::load(src + slot_of((j0 + j < cnt ? j0 + j : cnt - 1) * step) * a.D + part * VEC);
#pragma unroll
                for (int j = 0; j < PS_EMB_ILP; ++j) {
                    if (j0 + j < cnt) {
                        if (have) { VFOR(i) S.at(i) = p[j].get(i) + S.at(i); }
                        else { S = p[j]; have = true; }
                    }
                }
            }
        };
        add_partials(false);
        VFOR(i) S.at(i) = div_rn(S.get(i), (float)n);
        if (a.grad_mode == PS_GRAD_COMPAT) {
            add_partials(true);
            VFOR(i) S.at(i) = div_rn(S.get(i), (float)(2 * n));
        }
    }
    // (sharded step after the field sort: runs come field by field, the gradients go out in the plan's send order)
    const uint32_t uo = a.out_slot ? a.out_slot[a.sorted_ent[s0]] : (uint32_t)u;
    finish_key<VEC>(a, uo, row, n, S, part);
}

// Round 6: the runs above PS_EMB_SUPER_MIN chunks as a ROLE of this launch (workgroups [0, a.super_blocks)) instead of a launch of their own
// (k_emb_super_list, 8 us) in front of it -- at configs[4]'s shape that launch sat between two gaps of 14 and 7 us on the step's main chain,
// the chip being full of the next step's presort (profiles/r06_c4_gpu_timeline.txt).  One workgroup per run of the sort's list: its lane
// groups fold the run's chunk partials 32 at a time into super partials (k_emb_super_list's arithmetic, 8 loads in flight instead of 32: this
// kernel's register budget) into LDS, `cap` super partials per round; lane group 0 folds them in order (reduce_one_key's arithmetic) as they
// appear; compat mode walks the run a second time.  Same adds in the same order as the two-launch form.
// gs = 1 (runs of list_min < chunks <= PS_EMB_SUPER_MIN, the one-level order of reduce_one_key): the lane groups only FETCH the run's chunk
// partials, `cap` per round and one memory round trip for all of them, and lane group 0 folds them in chunk order from LDS -- one lane group of the
// short role walked them eight loads at a time (16 dependent round trips for 128 partials: the workgroups holding a 500..4000-entry key ended
// 20-40 us behind the median one and WERE the launch's tail, tools/emb_timing.sh).  Same adds in the same order.
template <int VEC>
__device__ __forceinline__ void super_key_run(const EmbBwdArgs &a, float *lds, int lds_floats, uint32_t u, uint32_t s0, uint32_t e0, const uint32_t gs) {
    const uint32_t CH = PS_EMB_CHUNK;
    const uint32_t n = e0 - s0, nch = (n + CH - 1) / CH, ngrp = (nch + gs - 1) / gs;
    const int lane64 = (int)(threadIdx.x & 63), gpw = 64 / a.LPR;
    const bool act = lane64 / a.LPR < gpw;
    const int part = lane64 % a.LPR;
    const uint32_t grp = (threadIdx.x >> 6) * gpw + lane64 / a.LPR, ngl = 4 * gpw;         // this lane group, lane groups per workgroup
    const uint32_t cap = (uint32_t)(lds_floats / a.D) < ngl ? (uint32_t)(lds_floats / a.D) : ngl;     // super partials per round (one per lane group at most)
    const int npass = a.grad_mode == PS_GRAD_COMPAT ? 2 : 1;
    Vec<VEC> S = Vec<VEC>::zero();
    bool have = false;
    for (int pass = 0; pass < npass; ++pass) {
        for (uint32_t g0 = 0; g0 < ngrp; g0 += cap) {
            const uint32_t g = g0 + grp;
            if (act && grp < cap && g < ngrp) {
                const uint32_t j = g * gs, j1 = j + gs < nch ? j + gs : nch;
                Vec<VEC> acc;
                bool h = false;
                if (gs == 1) {
                    const uint32_t sc = s0 + j * CH;
                    acc = Vec<VEC>::load(a.partials + ((size_t)2 * (sc / CH) + (j == 0 ? 1 : 0)) * a.D + part * VEC);
                } else
                for (uint32_t k0 = j; k0 < j1; k0 += PS_EMB_ILP) {
                    Vec<VEC> p[PS_EMB_ILP];
#pragma unroll
                    for (int k = 0; k < PS_EMB_ILP; ++k) {
                        const uint32_t jj = k0 + k < j1 ? k0 + k : j1 - 1;
                        const uint32_t sc = s0 + jj * CH;
                        p[k] = Vec<VEC>::load(a.partials + ((size_t)2 * (sc / CH) + (jj == 0 ? 1 : 0)) * a.D + part * VEC);
                    }
#pragma unroll
                    for (int k = 0; k < PS_EMB_ILP; ++k)
                        if (k0 + k < j1) {
                            if (h) { VFOR(i) acc.at(i) = p[k].get(i) + acc.at(i); }
                            else { acc = p[k]; h = true; }
                        }
                }
                acc.store(lds + (size_t)grp * a.D + part * VEC);
            }
            __syncthreads();
            if (grp == 0 && act) {
                const uint32_t cnt = ngrp - g0 < cap ? ngrp - g0 : cap;
                for (uint32_t q = 0; q < cnt; ++q) {
                    const Vec<VEC> v = Vec<VEC>::load(lds + (size_t)q * a.D + part * VEC);
                    if (have) { VFOR(i) S.at(i) = v.get(i) + S.at(i); }
                    else { S = v; have = true; }
                }
            }
            __syncthreads();
        }
        if (grp == 0 && act) { VFOR(i) S.at(i) = div_rn(S.get(i), (float)((pass == 0 ? 1u : 2u) * n)); }
    }
    if (grp == 0 && act) {
        const uint32_t row = a.sorted_key[s0];
        const uint32_t uo = a.out_slot ? a.out_slot[a.sorted_ent[s0]] : u;
        finish_key<VEC>(a, uo, row, n, S, part);
    }
}

// SEQ: the reference's summation order for every key.  Blocks [0, a.long_blocks) are the long-key waves above,
// the rest handle one key per lane group as before and leave keys above PS_EMB_CHUNK to them.
#ifdef PS_EMB_TIMING      // (tools/emb_timing.sh: every workgroup's start and end, wall clock; the product build carries none)
__device__ unsigned long long g_emb_t[8192 * 2];
struct EmbWgTimer {
    __device__ __forceinline__ EmbWgTimer() { if (threadIdx.x == 0 && blockIdx.x < 8192) g_emb_t[2 * blockIdx.x] = wall_clock64(); }
    __device__ __forceinline__ ~EmbWgTimer() { if (threadIdx.x == 0 && blockIdx.x < 8192) g_emb_t[2 * blockIdx.x + 1] = wall_clock64(); }
};
#else
struct EmbWgTimer {};
#endif
template <int VEC, bool BAG, bool SEQ>
__global__ __launch_bounds__(256) void k_emb_reduce_update(EmbBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float seq_lds[SEQ ? 2 * SEQ_LDS_FLOATS : SUPER_LDS_FLOATS];
    EndWait end_wait(a.end_wait, a.end_val, a.bound);       // (declared first: runs after the stamp's end; every return path)
#ifdef PS_EMB_TIMING
    EmbWgTimer wg_timer;
#endif
    // (no raised wave priority here: the dW GEMM and the dense update that run beside this kernel END the step's side
    //  chain -- with this kernel ahead of them the step got longer, 0.1530 against 0.1493 ms)
    StampScope stamp(a.ts, (SEQ && a.long_list) ? (unsigned int)a.long_blocks : 0u);
    if (SEQ && a.flag && blockIdx.x == 0 && threadIdx.x == 0)         // (chunked order: k_emb_partials is the first launch)
        __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.skip && *a.skip) return;
    if (SEQ && (int)blockIdx.x < a.long_blocks) {
        if (a.ablate & 2) return;                               // measurement: the kernel without its long-key role
        // (raised wave priority for these workgroups only: measured worse, 0.1515 against 0.1493 ms/step)
        if (a.long_list) {
            // the sort listed the runs above SEQ_TILE entries (nseg[1] of them, any order)
            const uint32_t nl = *a.nlong;
            const bool by_pair = a.xcd == 3;
            // (a.xcd == 1 | 2, a.lxcd: the list lies field by field -- XCD x takes its x-th EIGHTH, the fields the short role's eighth x
            //  roughly covers too: balanced, and only the pairs an eighth ends in are read under two L2s)
            const bool by_eighth = a.xcd && !by_pair && a.lxcd;
            const uint32_t x = blockIdx.x & 7u;
            const uint32_t step = (by_pair || by_eighth) ? (uint32_t)a.long_blocks >> 3 : (uint32_t)a.long_blocks;
            const uint32_t lo = by_eighth ? (uint32_t)((uint64_t)nl * x / 8) : 0u, hi = by_eighth ? (uint32_t)((uint64_t)nl * (x + 1) / 8) : nl;
            for (uint32_t t = (by_pair || by_eighth) ? blockIdx.x >> 3 : blockIdx.x; ; t += step) {
                uint32_t i = lo + t;
                if (by_pair) { if (!emb_xcd_pick(a.ftab, a.F, (int)x, 1, t, i)) break; }     // the XCD's own field pairs' long runs
                else if (i >= hi) break;
                const uint32_t *ll = a.long_list + 3 * (size_t)i;        // (run id, first entry, end): one load level
                long_key_run<VEC, BAG>(a, seq_lds, ll[0], ll[1], ll[2]);
                __syncthreads();                               // the next run reuses the LDS buffers
            }
            return;
        }
        // no list: the workgroup of the SEQ_TILE-entry tile in which a long run starts owns it
        const uint32_t CH = SEQ_TILE;
        const int64_t c = blockIdx.x;
        if (c * CH >= a.nnz) return;
        const uint32_t t0 = (uint32_t)(c * CH);
        const uint32_t t1 = (uint32_t)((int64_t)t0 + CH < a.nnz ? t0 + CH : a.nnz) - 1;
        const uint32_t u = a.seg_id[t1];
        const uint32_t s0 = a.seg_start[u], e0 = a.seg_start[u + 1];
        if (s0 < t0 || e0 - s0 <= CH) return;                  // the run starts in an earlier tile, or is a short key (block-uniform)
        long_key_run<VEC, BAG>(a, seq_lds, u, s0, e0);
        return;
    }
    if (!SEQ && (int)blockIdx.x < a.super_blocks) {
        const uint32_t nl = *a.nlong;
        for (uint32_t i = blockIdx.x; i < nl; i += (uint32_t)a.super_blocks) {
            const uint32_t *ll = a.long_list + 3 * (size_t)i;
            const uint32_t nch = (ll[2] - ll[1] + PS_EMB_CHUNK - 1) / PS_EMB_CHUNK;
            if (nch <= (uint32_t)a.list_min) continue;             // (a list with shorter runs in it: the sharded step's)
            super_key_run<VEC>(a, seq_lds, SUPER_LDS_FLOATS, ll[0], ll[1], ll[2], nch > PS_EMB_SUPER_MIN ? PS_EMB_SUPER : 1u);
        }
        return;
    }
    if (SEQ && (a.ablate & 4)) return;                          // measurement: the kernel without its short-key role
    // One lane group per key, GRID-STRIDE: the launcher cannot know the number of unique keys (it lives on the device) and
    // used to size the grid for the worst case, one key per entry -- at a multi-hot batch 50 k workgroups of which 40 k
    // found nothing to do; dispatching them was a sixth of the kernel.  Now a bounded grid walks the keys.
    const int sb = (int)blockIdx.x - (SEQ ? a.long_blocks : a.super_blocks);
    const int lane64 = (int)(threadIdx.x & 63);
    const int gpw = 64 / a.LPR;
    if (lane64 / a.LPR >= gpw) return;
    const int part = lane64 % a.LPR;
    const int64_t nseg = (int64_t)*a.nseg;
    // ONE call site of reduce_one_key (a second one made hipcc keep a private copy of the argument struct: 856 bytes of scratch per
    // lane in the SEQ instantiation, the single-hot update 135 us instead of 17)
    int64_t u = (((int64_t)sb * 4 + (threadIdx.x >> 6)) * gpw) + lane64 / a.LPR, uend = nseg;
    int64_t stride = (int64_t)a.short_blocks * 4 * gpw;            // lane groups in the short-key role's grid
    if (a.xcd) {
        // the keys whose run starts in eighth x of the sorted entries, walked by the workgroups of XCD x (emb_vblock's comment)
        const int x = (int)(blockIdx.x & 7u);
        // (a.xcd = 2: eighths of the ENTRIES instead of eighths of the keys)
        const int64_t u0 = a.xcd == 2 ? emb_key_at(a.seg_id, a.seg_start, a.nnz, nseg, a.nnz * x / 8) : nseg * x / 8;
        uend = a.xcd == 2 ? emb_key_at(a.seg_id, a.seg_start, a.nnz, nseg, a.nnz * (x + 1) / 8) : nseg * (x + 1) / 8;
        stride = (int64_t)(a.short_blocks >> 3) * 4 * gpw;
        u = u0 + ((int64_t)(sb >> 3) * 4 + (threadIdx.x >> 6)) * gpw + lane64 / a.LPR;
    }
    // (Round 5, measured and taken out again: the bounds of the run two keys ahead and the row of the next key requested while a key is
    //  reduced -- two of a key's four dependent round trips hidden for eight more VGPRs, 7 -> 6 waves per SIMD: the multi-hot step
    //  0.328-0.335 against 0.326 ms.  So was PS_EMB_SUPER_MIN 128 -> 16 (no lane group walks more than 16 partials: the workgroups that
    //  hold a 500..4000-entry key no longer end 25-35 us behind the median one, but k_emb_super_list then folds 570 runs instead of 78,
    //  7 -> 18 us in front of this launch: 0.327 ms).  tools/emb_timing.sh: the median workgroup of this launch ends at 28 us of 64, and
    //  moving its 230 MB of W / state in that time would take 8 TB/s -- the launch is within 1.5x of its HBM time.)
    if (SEQ && a.xcd == 3) {        // by field pair: u counts the runs of this XCD's fields (emb_xcd_pick)
        u = ((int64_t)(sb >> 3) * 4 + (threadIdx.x >> 6)) * gpw + lane64 / a.LPR;
        uend = (int64_t)1 << 40;
    }
    for (; u < uend; u += stride) {
        int64_t key = u;
        if (SEQ && a.xcd == 3) {
            uint32_t k32;
            if (!emb_xcd_pick(a.ftab, a.F, (int)(blockIdx.x & 7u), 0, (uint32_t)u, k32)) break;
            key = k32;
        }
        reduce_one_key<VEC, BAG, SEQ>(a, key, part);
    }
}


// ---------------------------------------------------------------------------
// PS owner side: apply a list of pushed (row, gradient) pairs
//   net/PServer.java:164-195 push -> KVStore.sum ; :197-214 psUpdate -> KVStore.update(updater,key)
//   BSP: g = (sum over the pushes of the key, in arrival order) / count, one updater step
//   async (:176-184): one updater step per push, in arrival order, no averaging
// ---------------------------------------------------------------------------
template <int VEC, bool IDENT>
__global__ __launch_bounds__(256) void k_rows_apply(RowsApplyArgs a) {
    if (a.skip && *a.skip) return;
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane64 = (int)(gt & 63);
    const int gpw = 64 / a.LPR;
    if (lane64 / a.LPR >= gpw) return;
    const int64_t u = (gt >> 6) * gpw + lane64 / a.LPR;
    const int part = lane64 % a.LPR;
    if (u >= (int64_t)*a.nseg) return;
    // identity: the list is already unique (one push per key), runs are single entries
    uint32_t s0 = (uint32_t)u, e0 = (uint32_t)u + 1;
    if (!IDENT) { s0 = a.seg_start[u]; e0 = a.seg_start[u + 1]; }
    const uint32_t row = a.sorted_key[s0];
    UpdParams U = a.upd;
    if (a.fu.ngroups > 1) U = row_upd(a, row);
    auto ent = [&](uint32_t k) -> size_t { return IDENT ? (size_t)k : (size_t)a.sorted_ent[k]; };
    float *wp = a.W + (size_t)row * a.D + part * VEC;
    float *sp = a.state + (size_t)row * 2 * a.D + part * VEC;
    Vec<VEC> w = Vec<VEC>::load(wp), s1 = Vec<VEC>::zero(), s2 = Vec<VEC>::zero();
    if (U.kind != PS_UPD_SIMPLE) { s1 = Vec<VEC>::load(sp); s2 = Vec<VEC>::load(sp + a.D); }
    const int lane = threadIdx.x & 63;
    auto apply = [&](const Vec<VEC> &g) {
        if (U.kind == PS_UPD_ADAM) { VFOR(i) adam_elem(U, g.get(i), w.at(i), s1.at(i), s2.at(i)); }
        else if (U.kind == PS_UPD_SIMPLE) { VFOR(i) w.at(i) = (g.get(i) * -U.eta) + w.get(i); }
        else {
            const float g0 = __shfl(g.get(0), lane - part);
            if (g0 != 0.f) { VFOR(i) ftrl_elem(U, g.get(i), w.at(i), s1.at(i), s2.at(i)); }
        }
    };
    if (a.is_async) {
        for (uint32_t k = s0; k < e0; ++k)
            apply(Vec<VEC>::load(a.grads + ent(k) * a.D + part * VEC));
    } else {
        Vec<VEC> S = Vec<VEC>::load(a.grads + ent(s0) * a.D + part * VEC);
        for (uint32_t k = s0 + 1; k < e0; ++k) {
            const Vec<VEC> g = Vec<VEC>::load(a.grads + ent(k) * a.D + part * VEC);
            VFOR(i) S.at(i) = g.get(i) + S.at(i);
        }
        VFOR(i) S.at(i) = div_rn(S.get(i), (float)(e0 - s0));
        apply(S);
    }
    w.store(wp);
    if (U.kind != PS_UPD_SIMPLE) { s1.store(sp); s2.store(sp + a.D); }
}

// ---------------------------------------------------------------------------
// PServer.push + psUpdate without a sort (net/PServer.java:164-214), for pushes that arrive
// grouped by worker with unique rows inside each worker's list (what ps_shard_plan sends):
//   mark : pos[worker][row] = entry, mask[row] |= 1 << worker        (atomicOr: order-free)
//   apply: the entry of the LOWEST pushing worker owns the row: walks the set bits in worker
//          order (= the arrival order the reference's synchronized push sees), mean or async,
//          one updater step, then clears mask[row] for the next step.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int push_peer_of(const PushApplyArgs &a, uint32_t e) {
    int p = 0;
    while (p + 1 < a.npeers && e >= a.peer_start[p + 1]) ++p;
    return p;
}

__global__ __launch_bounds__(256) void k_push_mark(PushApplyArgs a) {
    StampScope stamp(a.ts_mark);
    if (a.flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    const int p = push_peer_of(a, (uint32_t)e);
    const uint32_t r = a.rows_p[p][e - a.peer_start[p]];
    if ((int64_t)r >= a.R) { atomicAdd(a.err, 1); return; }
    a.pos[(size_t)p * a.R + r] = (uint32_t)e;
    atomicOr(&a.mask[r], 1u << p);
}

// ONE: a single pushing worker (its rows are unique): no mark pass, no mask -- every entry is its row's only push
// Mapped peer at N >= 2 (round 6): the worker-side gradient put as a ROLE of the owner push's first launch -- workgroups [0, PS_PUT_WGS) store this
// rank's gradients into the owners' receive regions, raise their flag words and workgroup 0 waits for the peers' (ps_put.h); the rest mark the
// pushed rows from the id lists, which are here already.  The two roles do not depend on each other and the kernel's end is the join: the
// apply launch behind it reads gradients that have all landed.  One launch and one boundary less on the step's critical chain than a put of its own.
__global__ __launch_bounds__(256) void k_push_mark_put(PushApplyArgs a, PeerPutArgs q) {
    StampScope stamp(a.ts_mark);
    if (blockIdx.x < PS_PUT_WGS) { peer_put_body(q, blockIdx.x); return; }
    const unsigned int mb = blockIdx.x - PS_PUT_WGS;
    if (a.flag && mb == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t e = (int64_t)mb * 256 + threadIdx.x;
    if (e >= a.n) return;
    const int p = push_peer_of(a, (uint32_t)e);
    const uint32_t r = a.rows_p[p][e - a.peer_start[p]];
    if ((int64_t)r >= a.R) { atomicAdd(a.err, 1); return; }
    a.pos[(size_t)p * a.R + r] = (uint32_t)e;
    atomicOr(&a.mask[r], 1u << p);
}

template <int VEC, bool ONE>
__global__ __launch_bounds__(256) void k_push_apply(PushApplyArgs a) {
    StampScope stamp(a.ts_apply);
    if (ONE && a.flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane64 = (int)(gt & 63);
    const int gpw = 64 / a.LPR;
    if (lane64 / a.LPR >= gpw) return;
    const int64_t e = (gt >> 6) * gpw + lane64 / a.LPR;
    const int part = lane64 % a.LPR;
    if (e >= a.n) return;
    const int pe = ONE ? 0 : push_peer_of(a, (uint32_t)e);
    const uint32_t row = a.rows_p[pe][e - a.peer_start[pe]];
    UpdParams U = a.upd;
    if (a.fu.ngroups > 1) U = row_upd(a, row);
    if ((int64_t)row >= a.R) { if (ONE && part == 0) atomicAdd(a.err, 1); return; }
    uint32_t m = ONE ? 1u : a.mask[row];
    if (m == 0u || pe != __ffs((int)m) - 1) return;     // another worker's entry leads this row
    const int cnt = __popc(m);
    float *wp = a.W + (size_t)row * a.D + part * VEC;
    float *sp = a.state + (size_t)row * 2 * a.D + part * VEC;
    Vec<VEC> w = Vec<VEC>::load(wp), s1 = Vec<VEC>::zero(), s2 = Vec<VEC>::zero();
    if (U.kind != PS_UPD_SIMPLE) { s1 = Vec<VEC>::load(sp); s2 = Vec<VEC>::load(sp + a.D); }
    const int lane = threadIdx.x & 63;
    auto apply = [&](const Vec<VEC> &g) {
        if (U.kind == PS_UPD_ADAM) { VFOR(i) adam_elem(U, g.get(i), w.at(i), s1.at(i), s2.at(i)); }
        else if (U.kind == PS_UPD_SIMPLE) { VFOR(i) w.at(i) = (g.get(i) * -U.eta) + w.get(i); }
        else {
            const float g0 = __shfl(g.get(0), lane - part);
            if (g0 != 0.f) { VFOR(i) ftrl_elem(U, g.get(i), w.at(i), s1.at(i), s2.at(i)); }
        }
    };
    Vec<VEC> S = Vec<VEC>::load(a.grads_p[pe] + (size_t)(e - a.peer_start[pe]) * a.D + part * VEC);     // the leader's own push comes first
    if (a.is_async) apply(S);
    m &= m - 1;
    while (m) {
        const int q = __ffs((int)m) - 1;
        m &= m - 1;
        const uint32_t ent = a.pos[(size_t)q * a.R + row];
        const Vec<VEC> g = Vec<VEC>::load(a.grads_p[q] + (size_t)(ent - a.peer_start[q]) * a.D + part * VEC);
        if (a.is_async) apply(g);
        else { VFOR(i) S.at(i) = g.get(i) + S.at(i); }
    }
    if (!a.is_async) {
        VFOR(i) S.at(i) = div_rn(S.get(i), (float)cnt);
        apply(S);
    }
    w.store(wp);
    if (U.kind != PS_UPD_SIMPLE) { s1.store(sp); s2.store(sp + a.D); }
    if (!ONE && part == 0) a.mask[row] = 0u;
}

// ---------------------------------------------------------------------------
// dense tensors: split-K reducer + /B + updater, writes W' ([in+1][out]) and its transpose
// ---------------------------------------------------------------------------
// One workgroup per 32 x 32 tile of a layer's [K+1][N] tensor, 4 elements per thread (rows ty, ty+8, ...).  W', the
// state and the slabs are read and written along n (128-byte rows per half wave); the transpose the forward GEMM reads
// goes through LDS so that its stores run along k the same way -- one 4-byte store per thread straight into Wt[n][k]
// touched a different cache line per lane and tripled the kernel's write traffic (profiles/ r01: 13.4 MB written for
// 5.6 MB of tensors).
__device__ __forceinline__ void wide_update_body(const WideUpdArgs &a, const int64_t r);
__global__ __launch_bounds__(256) void k_dense_update(DenseUpdArgs a) {
    StampScope stamp(a.ts);
    if (a.started_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.started_flag, a.started_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    start_wait(a.wait_flag, a.wait_val, a.bound);
    start_wait(a.wait_flag2, a.wait_val2, a.bound);
    if ((int)blockIdx.x >= a.tile_blocks) {                // the optional wide-table pass of the same launch
        wide_update_body(a.wide, (int64_t)((int)blockIdx.x - a.tile_blocks) * 256 + threadIdx.x);
        return;
    }
    if (a.skip && *a.skip) return;
    __shared__ float tile[32][33];
    int l = 0;
    while (l < a.nlayers - 1 && (int)blockIdx.x >= a.L[l].tile_end) ++l;
    const DenseLayer &L = a.L[l];
    const int tb = (int)blockIdx.x - L.tile_begin;
    const int tiles_n = (L.N + 31) >> 5;
    const int k0 = (tb / tiles_n) << 5, n0 = (tb % tiles_n) << 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = n0 + tx;
    const int row_lo = L.row_cnt > 0 ? L.row_lo : 0, row_hi = L.row_cnt > 0 ? L.row_lo + L.row_cnt - 1 : L.K;
    // The element's W, S1, S2 (and, on the multi-worker path, its flat gradient) are requested FIRST, unconditionally and from
    // clamped addresses, so that they travel with the slab loads below: as loads inside the per-element loop -- behind its
    // `continue`s and between four Adam evaluations -- each element paid its own memory round trip, and this kernel closes the
    // step (round 4: the dense update's tail is what the next step's first GEMM waits for).
    size_t wi4[4];
    int64_t t4[4];
    float w4[4] = {0.f, 0.f, 0.f, 0.f}, s14[4] = {0.f, 0.f, 0.f, 0.f}, s24[4] = {0.f, 0.f, 0.f, 0.f}, fg4[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const int nc = n < L.N ? n : L.N - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + ty + 8 * j, kc = k < row_lo ? row_lo : k > row_hi ? row_hi : k;
            wi4[j] = (size_t)kc * L.ldw + nc;
            t4[j] = L.elem_begin + (int64_t)kc * L.N + nc;
        }
        if (a.apply) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { w4[j] = L.W[wi4[j]]; s14[j] = L.S1[wi4[j]]; s24[j] = L.S2[wi4[j]]; }
        }
        if (a.flat_grad) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fg4[j] = a.flat_grad[t4[j]];
        }
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (!a.flat_grad && n < L.N) {
        const float *__restrict__ pn = L.part + n;
        for (int z0 = 0; z0 < L.nsplit; z0 += 4) {        // fixed slab order per element, 16 loads in flight
            float v[4][4];
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) {
                const int z = z0 + zz < L.nsplit ? z0 + zz : L.nsplit - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + ty + 8 * j;
                    v[zz][j] = pn[(size_t)z * L.part_stride + (size_t)(k < row_lo ? row_lo : k > row_hi ? row_hi : k) * L.ldp];
                }
            }
#pragma unroll
            for (int zz = 0; zz < 4; ++zz)
                if (z0 + zz < L.nsplit) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += v[zz][j];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + 8 * j;                    // k in [0, K] (K = bias row), n in [0, N)
        if (k < row_lo || k > row_hi || n >= L.N) continue;
        const int64_t t = t4[j];
        float g;
        if (a.flat_grad) {
            // multi-worker path: the (all-reduced) mean gradient was materialised flat
            g = fg4[j];
            if (a.flat_div > 0.f) g = div_rn(g, a.flat_div);
        } else {
            g = div_rn(s[j], (float)a.B);                 // divi(delta.columns) FcLayer.java:105 / rowMeans :103
        }
        if (a.grad_out) a.grad_out[t] = g;
        if (!a.apply) continue;
        const size_t wi = wi4[j];
        float w = w4[j], s1 = s14[j], s2 = s24[j];
        if (a.upd.kind == PS_UPD_ADAM) adam_elem(a.upd, g, w, s1, s2);
        else if (a.upd.kind == PS_UPD_SIMPLE) w = (g * -a.upd.eta) + w;
        else ftrl_elem(a.upd, g, w, s1, s2);   // per-tensor "dw[0]==0" skip is not meaningful for dense tensors
        L.W[wi] = w; L.S1[wi] = s1; L.S2[wi] = s2;
        tile[ty + 8 * j][tx] = w;
    }
    if (!a.apply) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + ty + 8 * j, kk = k0 + tx;
        if (nn < L.N && kk >= row_lo && kk <= row_hi) L.Wt[(size_t)nn * L.ldwt + kk] = tile[tx][ty + 8 * j];
    }
#if PS_GEMM_LAB
    if (L.Wp) {
        // ... and the fragment-order copy k_fwd_panel streams (kernels_panel.hip): the tile is 2 x 2 fragments of 16 features x 16 k,
        // one wave each -- lane (n % 16, (k % 16) / 4) holds four consecutive k of one feature, 1 KiB contiguous per wave
        const int lane = threadIdx.x & 63, fr = threadIdx.x >> 6;
        const int nl = (fr & 1) * 16 + (lane & 15), kl = (fr >> 1) * 16 + 4 * (lane >> 4);
        const int nn = n0 + nl, kk = k0 + kl;
        if (nn < L.N && kk < L.ldwt) {
            float *dst = L.Wp + ((((size_t)(nn >> 4) * (L.ldwt >> 4) + (kk >> 4)) * 64 + lane) << 2);
            if (kk >= row_lo && kk + 3 <= row_hi) *reinterpret_cast<float4 *>(dst) = make_float4(tile[kl][nl], tile[kl + 1][nl], tile[kl + 2][nl], tile[kl + 3][nl]);
            else {
#pragma unroll
                for (int m = 0; m < 4; ++m) if (kk + m >= row_lo && kk + m <= row_hi) dst[m] = tile[kl + m][nl];
            }
        }
    }
#endif
}

// materialise the flat dense gradient (sum of split partials / B) without updating
__device__ __forceinline__ void wide_update_body(const WideUpdArgs &a, const int64_t r) {
    if (a.skip && *a.skip) return;
    if (a.mode == 1 || a.mode == 5) {
        // sharded worker, before the all-reduce: G[k] = touched[k] * gbar, C[k] = touched[k]
        // (the PS averages a key over the workers that pushed it: net/PServer.java:164-214)
        // mode 5 (wide_grad_mode = intended): the keys' G / C were written by k_wide_intended; only the bias here
        const float gb = a.gbar[0];
        if (r < a.rows) { if (a.mode == 1) { const float t = a.touched[r] ? 1.f : 0.f; a.G[r] = t * gb; a.C[r] = t; } }
        else if (r == a.rows) a.G[2 * a.rows] = gb;          // wide.bias: every worker pushes it
        return;
    }
    if (a.mode == 6) {
        // sharded worker, before the all-reduce, slot form (WideUpdArgs.slots): r walks [bias | world x (gbar, words)]
        const float gb = a.gbar[0];
        const int64_t per = 1 + a.slot_words;
        if (r == 0) { a.slots[0] = gb; return; }              // wide.bias: every worker pushes it (summed, / world at the update)
        const int64_t q = r - 1;
        if (q >= per * a.world) return;
        const int w = (int)(q / per); const int64_t j = q % per;
        float v = 0.f;                                        // another worker's slot: zero, so that the sum is that worker's own value
        if (w == a.rank) {
            if (j == 0) v = gb;
            else {
                const int64_t k0 = (j - 1) * 24;
                uint32_t word = 0;
#pragma unroll
                for (int b = 0; b < 24; ++b) word |= (k0 + b < a.rows && a.touched[k0 + b < a.rows ? k0 + b : a.rows - 1]) ? (1u << b) : 0u;
                v = (float)word;                              // (< 2^24: exact)
            }
        }
        a.slots[1 + q] = v;
        return;
    }
    float g = a.gbar ? a.gbar[0] : 0.f;
    if (a.mode == 7) {
        // after the all-reduce, slot form: mean over the workers that touched the key, added in rank order
        if (r < a.rows) {
            const int64_t per = 1 + a.slot_words;
            const float *sl = a.slots + 1;
            const int64_t wi = 1 + r / 24; const int bit = (int)(r % 24);
            float G = 0.f, c = 0.f;
            for (int w = 0; w < a.world; ++w) {
                const uint32_t word = (uint32_t)sl[w * per + wi];
                if ((word >> bit) & 1u) { G += 1.f * sl[w * per]; c += 1.f; }
            }
            if (!(c > 0.f)) return;
            g = div_rn(G, c);
        } else if (r == a.rows) g = div_rn(a.slots[0], (float)a.nworkers);
    }
    if (a.mode == 2) {
        // after the all-reduce: mean over the workers that touched the key
        if (r < a.rows) { const float c = a.C[r]; if (!(c > 0.f)) return; g = div_rn(a.G[r], c); }
        else if (r == a.rows) g = div_rn(a.G[2 * a.rows], (float)a.nworkers);
    }
    if (r == a.rows) {
        // "wide.bias": 1x1, same rowMeans(delta)  (layer/LRLayer.java:106-107)
        if (a.upd.kind == PS_UPD_FTRL) { if (g != 0.f) ftrl_elem(a.upd, g, a.bias[0], a.bias_state[0], a.bias_state[1]); }
        else if (a.upd.kind == PS_UPD_ADAM) adam_elem(a.upd, g, a.bias[0], a.bias_state[0], a.bias_state[1]);
        else a.bias[0] = (g * -a.upd.eta) + a.bias[0];
        return;
    }
    if (r > a.rows || a.mode == 3) return;              // mode 3: "wide.bias" only (the keys go through k_wide_intended)
    // compat (layer/LRLayer.java:110-117): every key ever touched gets the same gbar
    if (a.mode == 0 && !a.touched[r]) return;
    float w = a.W[r], z = a.state[2 * r], n = a.state[2 * r + 1];
    if (a.upd.kind == PS_UPD_FTRL) { if (g == 0.f) return; ftrl_elem(a.upd, g, w, z, n); }
    else if (a.upd.kind == PS_UPD_ADAM) adam_elem(a.upd, g, w, z, n);
    else w = (g * -a.upd.eta) + w;
    a.W[r] = w; a.state[2 * r] = z; a.state[2 * r + 1] = n;
}

__global__ __launch_bounds__(256) void k_wide_update(WideUpdArgs a) {
    wide_update_body(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// ---------------------------------------------------------------------------
// init / row access
// ---------------------------------------------------------------------------
// Grid-stride: a HIP launch carries at most 2^32 - 1 threads, and a configs[3]-sized table (320 M rows x 64 =
// 2e10 elements) is far beyond that -- one thread per element silently initialised only the first 2^32 of them.
__global__ void k_init_emb(float *W, int64_t rows, int D, uint64_t seed, uint64_t table, float scale,
                           int64_t id_first, int64_t id_stride, const uint32_t *ids /* or the progression */) {
    const int64_t total = rows * D, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t r = t / D; const int d = (int)(t % D);
        const uint64_t id = ids ? (uint64_t)ids[r] : (uint64_t)(id_first + r * id_stride);
        W[t] = ps_init_value(seed, table, id, (uint64_t)d, scale);
    }
}

__global__ void k_init_dense(float *W, float *Wt, int K, int N, int ldw, int ldwt, uint64_t seed,
                             uint64_t table_w, float scale_w, uint64_t table_b, float scale_b) {
    // reference layouts: weights out x in column-major => flat index n + N*k; bias flat index n
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)(K + 1) * N) return;
    const int k = (int)(t / N), n = (int)(t % N);
    const float v = k < K ? ps_init_value(seed, table_w, (uint64_t)((int64_t)n + (int64_t)N * k), 0, scale_w)
                          : ps_init_value(seed, table_b, (uint64_t)n, 0, scale_b);
    W[(size_t)k * ldw + n] = v;
    Wt[(size_t)n * ldwt + k] = v;
}

__global__ void k_fill(float *p, int64_t n, float v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) p[t] = v;
}
__global__ void k_fill_col(float *p, int rows, int ld, int col, float v) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < rows) p[(size_t)t * ld + col] = v;
}

// rows[i][0..D) <-> table[row_of(ids[i])]; stride/offset select W or an interleaved state slot
__global__ void k_rows_copy(float *table, int64_t row_stride, int64_t col_off, const int64_t *rows_idx,
                            int64_t n, int D, float *buf, int to_table) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * D) return;
    const int64_t i = t / D; const int d = (int)(t % D);
    float *cell = table + rows_idx[i] * row_stride + col_off + d;
    if (to_table) *cell = buf[t]; else buf[t] = *cell;
}

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int g_mh_ilp16 = 0;
int g_emb_list_min = 16;    // ps_tune_set("emb_list_min", chunks): the chunked order's runs above this many 32-entry chunks go to the update launch's list role (128: only the runs with super partials, round 5)
int g_emb_list_grid = 256;  // ps_tune_set("emb_list_grid", workgroups) of that role
int g_super_in_update = 1;  // ps_tune_set("super_in_update", 0): the very long runs' super partials by a launch of their own (k_emb_super_list) again
int g_emb_lxcd = 1;         // ps_tune_set("emb_lxcd", 0): the long-key role takes the sort's list in order, any XCD (rounds 4-5)
int g_emb_xcd = 1;          // ps_tune_set("emb_xcd", 0): the embedding backward's keys / tiles dealt round robin instead of XCD by XCD (emb_vblock)
int g_fwd_order = 3;        // ps_tune_set("fwd_order", bits): multi-hot gather's bag order (EmbFwdArgs.order; 0: sample-major round robin -- 0.354 against 0.347 ms / step at configs[4]'s shape, the gather 47.8 -> 42.1 us)
int g_keys_grid = 0;        // ps_tune_set("keys_grid", workgroups): grid bound of the multi-hot key kernel (0: 1024)
int g_seq_long_grid = 0;        // ps_tune_set("seq_long_grid", workgroups): long-key workgroups of the sequential order (0: SEQ_LONG_GRID)
int g_emb_short_grid = 2048;    // ps_tune_set("emb_short_grid", workgroups): grid of the embedding update's one-key-per-lane-group role (multi-hot step: 1024/2048 0.361, 4096 0.3645, 8192 0.372 ms; the single-hot step does not care)
int g_seq_ablate = 0;    // measurement only (results wrong): 1 = the fold wave skips its LDS reads + adds, 2 = the loaders skip their global loads

// The LDS-staged form of the single-hot gather that BASELINE.json's north_star names ("coalesced CSR gather with
// LDS-staged rows"): every wave DMAs its rows global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, 4 KiB in
// flight per wave), then reads them back, applies relu and stores.  Kept as a measured alternative (ps_tune_set
// "gather_lds"): on the 256 GB table it is NOT faster than plain 16-byte vector loads -- the gather is bound by the
// DRAM/TLB behaviour of random 256-byte reads, not by registers or issue slots (numbers in DESIGN.md).
// A rejected measurement variant: compiled into the LAB build only (tools/gemm_lab_build.sh, -DPS_GEMM_LAB=1), like the
// rejected GEMM variants; the product library has no such kernel and ps_tune_set("gather_lds", 1) is refused there.
#ifndef PS_GEMM_LAB
#define PS_GEMM_LAB 0
#endif
#if PS_GEMM_LAB
namespace {
__global__ __launch_bounds__(256) void k_emb_fwd_lds(EmbFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float stage[4][GATHER_ILP][64 * 4];      // [wave][slot][lane * 4 floats]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t grp = gt / a.LPR;
    const int part = (int)(gt % a.LPR);
    const int64_t nb = (int64_t)a.B * a.F;
    const int64_t bag0 = grp * GATHER_ILP;
    int64_t rows[GATHER_ILP];
#pragma unroll
    for (int j = 0; j < GATHER_ILP; ++j) {
        const int64_t bag = bag0 + j < nb ? bag0 + j : nb - 1;
        const int f = (int)(bag % a.F);
        const int64_t rb = a.row_base[f], rn = a.row_base[f + 1] - rb;
        int64_t id = a.ids[bag];
        if (id < 0 || id >= rn) { if (part == 0) atomicAdd(a.err, 1); id = 0; }
        rows[j] = rb + id;
    }
#pragma unroll
    for (int j = 0; j < GATHER_ILP; ++j) {
        const float *src = a.W + (size_t)rows[j] * a.D + part * 4;
        // lane l's 16 bytes land at stage[w][j] + 16 * l (the LDS base is wave-uniform, the lane offset is implicit)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)&stage[w][j][0], 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < GATHER_ILP; ++j) {
        const int64_t bag = bag0 + j;
        if (bag < nb) {
            float4 v = *reinterpret_cast<const float4 *>(&stage[w][j][lane * 4]);
            if (a.act == PS_ACT_RELU) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
            *reinterpret_cast<float4 *>(a.out + (size_t)(bag / a.F) * a.ld + (size_t)(bag % a.F) * a.D + part * 4) = v;
        }
    }
}
}  // namespace
#endif      // PS_GEMM_LAB
int g_gather_lds = 0;      // measurement (lab build): 1 = the LDS-staged single-hot gather above

int g_gather_nt = -1;      // -1: automatic (streaming hints when the tables exceed the caches); 0..3: forced (bit 0 loads, bit 1 stores)

// Sort keys (and the bag of every entry) of a multi-hot batch straight from the ids: row = row_base[f] + id with the
// gather's clamping (errors are counted by the gather).  Lets the backward's sort start BESIDE the gather instead of
// behind it (configs[4]'s shape: the 80 us gather and the 136 us sort were back to back on the critical path).
// LANES lanes per bag, two entries per lane in flight (loads issued unconditionally from clamped addresses).  Beside the
// gather its duration is the gather's whatever LANES is (4 / 8 / 16 / 32 measured: 33-42 us), alone 16 us.
template <int LANES>
__global__ __launch_bounds__(256) void k_emb_keys(EmbFwdArgs a) {
    StampScope stamp(a.ts);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nb = (int64_t)a.B * a.F;
    int64_t bag = t / LANES;
    const int l = (int)(t % LANES);
    const bool live = bag < nb;
    if (!live) bag = nb - 1;
    const int f = (int)(bag % a.F);
    const int64_t rb = a.row_base[f], rn = a.row_base[f + 1] - rb;
    const int64_t p0 = a.offsets[bag], p1 = live ? a.offsets[bag + 1] : p0;
    for (int64_t p = p0 + l; p < p1; p += 2 * LANES) {
        const int64_t q = p + LANES;
        int64_t id0 = a.ids[p], id1 = a.ids[q < p1 ? q : p];
        if (id0 < 0 || id0 >= rn) id0 = 0;
        if (id1 < 0 || id1 >= rn) id1 = 0;
        a.key_out[p] = (uint32_t)(rb + id0);
        a.ent_bag[p] = (uint32_t)bag;
        if (q < p1) { a.key_out[q] = (uint32_t)(rb + id1); a.ent_bag[q] = (uint32_t)bag; }
    }
}

// ... for the segmented sort (kernels_sort.hip k_bag_scan): the pair (id inside the field, bag) of every entry at its place among
// its FIELD's entries -- pb[f] (the field's first slot, a multiple of tile) + pre[bag] (entries of the field in earlier samples)
// + position in the bag -- so that the sort needs no pass over the field bits
template <int LANES>
__global__ __launch_bounds__(256) void k_emb_keys_seg(EmbFwdArgs a, const uint32_t *__restrict__ pre, const uint32_t *__restrict__ ftotal, int tile,
                                                      uint32_t *__restrict__ kp, uint32_t *__restrict__ vp) {
    StampScope stamp(a.ts);
    __shared__ uint32_t pb[64];
    if (threadIdx.x == 0) {
        uint32_t p = 0;
        for (int f = 0; f < a.F; ++f) { pb[f] = p; p += (ftotal[f] + (uint32_t)tile - 1) / (uint32_t)tile * (uint32_t)tile; }
    }
    __syncthreads();
    const int64_t nb = (int64_t)a.B * a.F;
    // grid-stride over the bags (round 5: a bounded grid -- 3328 small workgroups at configs[4]'s shape were more than the chip holds at
    // once, and a kernel that is still being placed holds up the launches of the other streams behind it)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nb * LANES; t += (int64_t)gridDim.x * 256) {
        const int64_t bag = t / LANES;
        const int l = (int)(t % LANES);
        const int f = (int)(bag % a.F);
        const int64_t rn = a.row_base[f + 1] - a.row_base[f];
        const int64_t p0 = a.offsets[bag], p1 = a.offsets[bag + 1];
        const int64_t dst = (int64_t)pb[f] + pre[bag] - p0;         // entry p of the bag goes to dst + p
        for (int64_t p = p0 + l; p < p1; p += 2 * LANES) {
            const int64_t q = p + LANES;
            int64_t id0 = a.ids[p], id1 = a.ids[q < p1 ? q : p];
            if (id0 < 0 || id0 >= rn) id0 = 0;
            if (id1 < 0 || id1 >= rn) id1 = 0;
            kp[dst + p] = (uint32_t)id0;
            vp[dst + p] = (uint32_t)bag;
            if (q < p1) { kp[dst + q] = (uint32_t)id1; vp[dst + q] = (uint32_t)bag; }
        }
    }
}

int launch_emb_keys_seg(const EmbFwdArgs &a, const uint32_t *pre, const uint32_t *ftotal, int tile, uint32_t *kp, uint32_t *vp, hipStream_t st) {
    if (!a.offsets || !pre || !ftotal || !kp || !vp || a.F > 64) return ps_set_err(PS_E_BAD_ARG, "launch_emb_keys_seg: multi-hot batches of <= 64 fields only");
    const int64_t nb = (int64_t)a.B * a.F;
    if (nb <= 0) return PS_OK;
    EmbFwdArgs b = a;
    b.ts = stamp_next("emb_keys");
    hipLaunchKernelGGL(k_emb_keys_seg<8>, dim3(std::min<int64_t>(cdiv(nb * 8, 256), g_keys_grid > 0 ? g_keys_grid : 1024)), dim3(256), 0, st, b, pre, ftotal, tile, kp, vp);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_emb_keys(const EmbFwdArgs &a, hipStream_t st) {
    if (!a.offsets || !a.key_out || !a.ent_bag) return ps_set_err(PS_E_BAD_ARG, "launch_emb_keys: multi-hot batches only");
    const int64_t nb = (int64_t)a.B * a.F;
    if (nb <= 0) return PS_OK;
    EmbFwdArgs b = a;
    b.ts = stamp_next("emb_keys");
    hipLaunchKernelGGL(k_emb_keys<8>, dim3(cdiv(nb * 8, 256)), dim3(256), 0, st, b);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_emb_fwd(EmbFwdArgs a, hipStream_t st, LaunchOpts *lo, unsigned int *werr) {
    if (lo) lo->launched = false;
    a.end_wait = lo ? lo->wait : nullptr; a.end_val = lo ? lo->wait_val : 0u; a.bound = wait_bound(werr, 103);
    const hipEvent_t stop_ev = lo ? lo->stop_event : nullptr;
    // rows of tables far beyond the 256 MiB Infinity Cache are read once: non-temporal loads (measured on the 256 GB
    // table, tools/gather_nt.py: bags of 32 0.671 -> 0.707 of 8 TB/s, single-hot read+write 0.663 -> 0.694; nt stores: no effect)
    a.nt = a.table_bytes > ((size_t)1 << 30) ? 1 : 0;
    if (g_gather_nt >= 0) a.nt = g_gather_nt;
    a.ts = stamp_next("emb_fwd");
    const int vec = (a.D % 4 == 0) ? 4 : 1;
    a.LPR = a.D / vec;
    const bool multi = a.offsets != nullptr, slot = a.slot != nullptr;
    const int64_t nb = (int64_t)a.B * a.F;
    const int64_t groups = multi ? nb : (nb + GATHER_ILP - 1) / GATHER_ILP;
    a.gather_blocks = cdiv(groups * a.LPR, 256);
    // (a table far beyond the Infinity Cache has no rows to keep in an L2: the order the 256 GB gather was tuned on stays)
    a.order = (multi && !slot && a.gather_blocks >= 64 && a.table_bytes <= ((size_t)1 << 30)) ? g_fwd_order : 0;
    if (a.order & 1) a.gather_blocks = (a.gather_blocks + 7) & ~7;
    const int dense_blocks = a.dense ? cdiv((int64_t)a.B * a.X, 256) : 0;
    a.wide_blocks = (a.wide_ids && a.wide_z && a.B > 0) ? cdiv(a.B, 32) : 0;
    a.wide_blk0 = a.gather_blocks + dense_blocks;
    const int grid = a.gather_blocks + dense_blocks + a.wide_blocks;
    if (grid == 0) return PS_OK;
    // multi-hot: ids handed round a lane group by shuffle when the group sits inside one wave
    const int mhi = (multi && 64 % a.LPR == 0) ? (a.LPR <= 4 ? a.LPR : a.LPR == 8 ? 2 : (g_mh_ilp16 > 0 ? g_mh_ilp16 : 4)) : 0;      // (D = 64: four row loads in flight per 16-lane group -- 192.8 us against 195.8 with one on the 256 GB table, bags of 32, tools/gather_sweep.py)
#define EMB_FWD_MH(V, S)                                                                                           \
    do {                                                                                                           \
        if (mhi == 4) PS_LAUNCH_EV((k_emb_fwd<V, true, S, 4>), dim3(grid), dim3(256), 0, st, stop_ev, a);          \
        else if (mhi == 2) PS_LAUNCH_EV((k_emb_fwd<V, true, S, 2>), dim3(grid), dim3(256), 0, st, stop_ev, a);     \
        else if (mhi == 1) PS_LAUNCH_EV((k_emb_fwd<V, true, S, 1>), dim3(grid), dim3(256), 0, st, stop_ev, a);     \
        else PS_LAUNCH_EV((k_emb_fwd<V, true, S, 0>), dim3(grid), dim3(256), 0, st, stop_ev, a);                   \
    } while (0)
#define EMB_FWD_LAUNCH(V)                                                                                          \
    do {                                                                                                           \
        if (multi) { if (slot) EMB_FWD_MH(V, true); else EMB_FWD_MH(V, false); }                                   \
        else { if (slot) PS_LAUNCH_EV((k_emb_fwd<V, false, true, 0>), dim3(grid), dim3(256), 0, st, stop_ev, a);   \
               else PS_LAUNCH_EV((k_emb_fwd<V, false, false, 0>), dim3(grid), dim3(256), 0, st, stop_ev, a); }      \
    } while (0)
#if PS_GEMM_LAB
    if (g_gather_lds && vec == 4 && !multi && !slot && !a.key_out && !a.dense && !a.wide_blocks && 64 % a.LPR == 0 && !stop_ev && !a.end_wait)
        hipLaunchKernelGGL(k_emb_fwd_lds, dim3(grid), dim3(256), 0, st, a);       // (measurement variant: carries no event / end wait)
    else
#endif
    if (vec == 4) EMB_FWD_LAUNCH(4); else EMB_FWD_LAUNCH(1);
#undef EMB_FWD_LAUNCH
#undef EMB_FWD_MH
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}

int launch_head(const HeadArgs &a, float *loss_out, float *gbar_out, int *skip, int force_no_skip, hipStream_t st) {
    hipLaunchKernelGGL(k_head, dim3(cdiv(a.B, 32)), dim3(256), 0, st, a);   // eight lanes per sample
    HIPCHK(hipGetLastError());
    if (a.labels && loss_out) return launch_loss_reduce(a, loss_out, gbar_out, skip, force_no_skip, st);
    return PS_OK;
}

int launch_emb_bwd(EmbBwdArgs a, hipStream_t st, LaunchOpts *lo, unsigned int *werr) {
    if (lo) lo->launched = false;
    a.end_wait = lo ? lo->wait : nullptr; a.end_val = lo ? lo->wait_val : 0u; a.bound = wait_bound(werr, 102);
    const int vec = (a.D % 4 == 0) ? 4 : 1;
    a.LPR = a.D / vec;
    if (a.nnz <= 0) return PS_OK;
    const int64_t tiles = (a.nnz + PS_EMB_CHUNK - 1) / PS_EMB_CHUNK;
    const int gpw = 64 / a.LPR;                // lane groups per wave
    if (gpw < 1) return ps_set_err(PS_E_UNSUPPORTED, "embedding dim %d needs more than one wave per row", a.D);
    const int gp = cdiv((int64_t)cdiv(tiles, gpw) * 64, 256);
    const int gr = cdiv((int64_t)cdiv(a.nnz, gpw) * 64, 256);  // upper bound on unique keys; extra groups exit on *nseg
    const bool bag = a.ent_bag != nullptr;
    a.ablate = g_seq_ablate;
    a.ts = stamp_next("emb_bwd_update");
    a.flag = lo ? lo->flag : nullptr; a.flag_val = lo ? lo->flag_val : 0;      // "this launch has started" (LaunchOpts, ps_common.h)
    // one workgroup per SEQ_TILE-entry tile looks for a long run starting in it -- or, with the sort's list of the
    // long runs, a fixed grid walks that list
    a.long_blocks = !a.seq_order ? 0 : a.long_list ? (g_seq_long_grid > 0 ? g_seq_long_grid : SEQ_LONG_GRID) : cdiv(a.nnz, SEQ_TILE);
    // the short-key role: enough workgroups to fill the chip a few times over, never more than one lane group per entry
    a.short_blocks = gr < g_emb_short_grid ? gr : g_emb_short_grid;
    // XCD-affine order (emb_vblock): grids rounded up to multiples of 8 (surplus workgroups find nothing to do)
    // chunked order: the runs above PS_EMB_SUPER_MIN chunks as a role of the reduce launch (super_key_run) when the sort listed them
    a.super_blocks = (!a.seq_order && a.long_runs && a.long_list && g_super_in_update) ? (g_emb_list_grid > 0 ? g_emb_list_grid : 64) : 0;
    if (a.list_min <= 0 || a.list_min > PS_EMB_SUPER_MIN) a.list_min = PS_EMB_SUPER_MIN;
    if (!a.seq_order) { a.ts_partials = stamp_next("emb_partials"); if (a.long_runs && !a.super_blocks) a.ts_super = stamp_next("emb_super"); }
    a.xcd = (g_emb_xcd && a.short_blocks >= 64 && a.long_blocks % 8 == 0) ? g_emb_xcd : 0;
    // by field pair when the field sort left its table (single-hot batches; ps_tune_set("emb_xcd", 1 | 2): by eighths of the keys / entries again)
    if (a.xcd == 3 && !(a.seq_order && a.long_list && a.ftab && a.F <= 64)) a.xcd = 1;
    a.lxcd = g_emb_lxcd;
    if (a.xcd) a.short_blocks = (a.short_blocks + 7) & ~7;
    const int gpx = a.xcd ? (gp + 7) & ~7 : gp;
#define EMB_BWD_LAUNCH(V, BG)                                                                  \
    do {                                                                                       \
        if (a.seq_order) {                                                                     \
            hipLaunchKernelGGL((k_emb_reduce_update<V, BG, true>), dim3(a.long_blocks + a.short_blocks), dim3(256), 0, st, a); \
            break;                                                                             \
        }                                                                                      \
        hipLaunchKernelGGL((k_emb_partials<V, BG>), dim3(gpx), dim3(256), 0, st, a);            \
        if (a.super_blocks) {}                                                                  \
        else if (a.long_runs && a.long_list) hipLaunchKernelGGL((k_emb_super_list<V>), dim3(128), dim3(256), 0, st, a); \
        else if (a.long_runs) hipLaunchKernelGGL((k_emb_super<V>), dim3(gpx), dim3(256), 0, st, a);  \
        hipLaunchKernelGGL((k_emb_reduce_update<V, BG, false>), dim3(a.super_blocks + a.short_blocks), dim3(256), 0, st, a); \
    } while (0)
    if (vec == 4) { if (bag) EMB_BWD_LAUNCH(4, true); else EMB_BWD_LAUNCH(4, false); }
    else { if (bag) EMB_BWD_LAUNCH(1, true); else EMB_BWD_LAUNCH(1, false); }
#undef EMB_BWD_LAUNCH
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}

// A tensor whose gradient arrives in MANY slabs (the out = 1 layer: one slab per 32 batch rows, 128 of them at
// B = 4096, for 257 elements) would make k_dense_update's one-thread-per-element slab walk the longest chain of
// the whole launch (16 batches of 8 dependent-on-nothing loads, ~19 us for 257 threads).  One WAVE per element
// folds the slabs first: lane j takes slabs j, j+64, ..., then a butterfly (a fixed order), result in slab 0.
namespace {
__global__ __launch_bounds__(256) void k_dense_prereduce(float *__restrict__ part, int64_t part_stride, int ldp, int N, int64_t elems, int nsplit) {
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= elems) return;
    const int k = (int)(e / N), n = (int)(e % N);
    float *pp = part + (size_t)k * ldp + n;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int z = lane + 64 * j;
        v[j] = pp[(size_t)(z < nsplit ? z : nsplit - 1) * part_stride];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (lane + 64 * j < nsplit) s += v[j];
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) pp[0] = s;
}
}  // namespace

// Folds the slabs of layer l in place when there are many of them (see k_dense_prereduce) and marks the layer as
// one slab.  Separate from the update so that the step can run it as soon as the slabs exist (the out = 1 layer's come
// from the head's launch) instead of in front of the update at the very end of the side chain.
int dense_prereduce(DenseUpdArgs &a, int l, hipStream_t st) {
    DenseLayer &L = a.L[l];
    if (a.flat_grad || !(L.nsplit > 32 && L.nsplit <= 256)) return PS_OK;
    // A row window (the keyed push: "fc<i>.weights" / "fc<i>.bias" slabs hold only their own rows behind a shifted base)
    // is not a full (K+1) x N slab: the fold would read and write outside it, and its butterfly is not the arrival order
    // KVStore.sum adds in.  k_dense_update walks such slabs itself, in order.
    if (L.row_cnt > 0) return PS_OK;
    const int64_t elems = (int64_t)(L.K + 1) * L.N;
    hipLaunchKernelGGL(k_dense_prereduce, dim3(cdiv(elems, 4)), dim3(256), 0, st, const_cast<float *>(L.part), L.part_stride, L.ldp, L.N, elems, L.nsplit);
    HIPCHK(hipGetLastError());
    L.nsplit = 1;
    return PS_OK;
}

int launch_dense_update(const DenseUpdArgs &a0, hipStream_t st) {
    DenseUpdArgs a = a0;
    for (int l = 0; l < a.nlayers; ++l) PSCHK(dense_prereduce(a, l, st));
    int tiles = 0;
    for (int l = 0; l < a.nlayers; ++l) {
        DenseLayer &L = a.L[l];
        L.tile_begin = tiles;
        tiles += cdiv(L.K + 1, 32) * cdiv(L.N, 32);
        L.tile_end = tiles;
    }
    if (tiles + a.wide_blocks == 0) return PS_OK;
    a.ts = stamp_next("dense_update");
    a.tile_blocks = tiles;
    hipLaunchKernelGGL(k_dense_update, dim3(tiles + a.wide_blocks), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// ---------------------------------------------------------------------------
// wide_grad_mode = intended: per-key presence-weighted gradient: g(key) = (sum over the key's
// (sample, field) occurrences, in that order, of delta_sample) / B, Ftrl on the keys of THIS batch only.
// The occurrences come sorted by key (stable radix sort of the B*F wide ids) with their segments.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_wide_keys(const int64_t *__restrict__ ids, int64_t n, int64_t rows, uint32_t *__restrict__ keys, int *err) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t id = ids[i];
    if (id < 0 || id >= rows) { atomicAdd(err, 1); id = 0; }
    keys[i] = (uint32_t)id;
}
__global__ __launch_bounds__(256) void k_wide_intended(WideIntendedArgs a) {
    if (a.skip && *a.skip) return;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= (int64_t)*a.nseg) return;
    const uint32_t s0 = a.seg_start[u], e0 = a.seg_start[u + 1];
    const uint32_t key = a.sorted_key[s0];
    float S = 0.f;
    for (uint32_t k = s0; k < e0; ++k) S = a.delta[(size_t)(a.sorted_ent[k] / (uint32_t)a.F) * a.ldd] + S;   // (sample, field) order
    const float g = div_rn(S, (float)a.B);
    if (a.G) { a.G[key] = g; a.C[key] = 1.f; return; }      // sharded worker: pushed, not applied
    float w = a.W[key], z = a.state[2 * (size_t)key], n = a.state[2 * (size_t)key + 1];
    if (a.upd.kind == PS_UPD_FTRL) { if (g == 0.f) return; ftrl_elem(a.upd, g, w, z, n); }
    else if (a.upd.kind == PS_UPD_ADAM) adam_elem(a.upd, g, w, z, n);
    else w = (g * -a.upd.eta) + w;
    a.W[key] = w; a.state[2 * (size_t)key] = z; a.state[2 * (size_t)key + 1] = n;
}
}  // namespace

int launch_wide_keys(const int64_t *ids, int64_t n, int64_t rows, uint32_t *keys, int *err, hipStream_t st) {
    if (n <= 0) return PS_OK;
    hipLaunchKernelGGL(k_wide_keys, dim3(cdiv(n, 256)), dim3(256), 0, st, ids, n, rows, keys, err);
    HIPCHK(hipGetLastError());
    return PS_OK;
}
int launch_wide_intended(const WideIntendedArgs &a, int64_t n, hipStream_t st) {
    if (n <= 0) return PS_OK;
    hipLaunchKernelGGL(k_wide_intended, dim3(cdiv(n, 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// keyed pushes of wide keys: one thread per key (these lists are tiny: the gRPC facade's psUpdate)
__global__ __launch_bounds__(256) void k_wide_list(WideListArgs a) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= a.nkeys) return;
    const int64_t id = a.key_ids[u];
    const bool is_bias = id == a.rows;
    float *wp = is_bias ? a.bias : a.W + id;
    float *sp = is_bias ? a.bias_state : a.state + 2 * id;
    float w = *wp, z = sp[0], n = sp[1];
    auto apply = [&](float g) {
        if (a.upd.kind == PS_UPD_FTRL) { if (g != 0.f) ftrl_elem(a.upd, g, w, z, n); }      // FtrlUpdater.java:52
        else if (a.upd.kind == PS_UPD_ADAM) adam_elem(a.upd, g, w, z, n);
        else w = (g * -a.upd.eta) + w;
    };
    const uint32_t e0 = a.key_off[u], e1 = a.key_off[u + 1];
    if (a.is_async) {
        for (uint32_t e = e0; e < e1; ++e) apply(a.grads[e]);
    } else {
        float s = a.grads[e0];
        for (uint32_t e = e0 + 1; e < e1; ++e) s = a.grads[e] + s;      // KVStore.sum: addi in arrival order
        apply(div_rn(s, (float)(e1 - e0)));                              // KVStore.update: divi(sumCnt)
    }
    *wp = w; sp[0] = z; sp[1] = n;
}
int launch_wide_list(const WideListArgs &a, hipStream_t st) {
    if (a.nkeys <= 0) return PS_OK;
    hipLaunchKernelGGL(k_wide_list, dim3(cdiv(a.nkeys, 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int wide_update_blocks(const WideUpdArgs &a) { return cdiv(a.mode == 6 ? 1 + (int64_t)(1 + a.slot_words) * a.world : a.rows + 1, 256); }
int launch_wide_update(const WideUpdArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(k_wide_update, dim3(cdiv(a.rows + 1, 256)), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_init_emb(float *W, int64_t rows, int D, uint64_t seed, uint64_t table, float scale,
                    int64_t id_first, int64_t id_stride, const uint32_t *ids_dev, hipStream_t st) {
    if (rows * D == 0) return PS_OK;
    const int64_t blocks = (rows * D + 255) / 256;
    hipLaunchKernelGGL(k_init_emb, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, W, rows, D, seed, table, scale, id_first, id_stride, ids_dev);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_init_dense(float *W, float *Wt, int K, int N, int ldw, int ldwt, uint64_t seed,
                      uint64_t table_w, float scale_w, uint64_t table_b, float scale_b, hipStream_t st) {
    hipLaunchKernelGGL(k_init_dense, dim3(cdiv((int64_t)(K + 1) * N, 256)), dim3(256), 0, st, W, Wt, K, N, ldw, ldwt, seed, table_w, scale_w, table_b, scale_b);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_fill(float *p, int64_t n, float v, hipStream_t st) {
    if (n <= 0) return PS_OK;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, p, n, v);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_fill_col(float *p, int rows, int ld, int col, float v, hipStream_t st) {
    if (rows <= 0) return PS_OK;
    hipLaunchKernelGGL(k_fill_col, dim3(cdiv(rows, 256)), dim3(256), 0, st, p, rows, ld, col, v);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_rows_copy(float *table, int64_t row_stride, int64_t col_off, const int64_t *rows_idx_dev,
                     int64_t n, int D, float *buf_dev, int to_table, hipStream_t st) {
    if (n <= 0) return PS_OK;
    hipLaunchKernelGGL(k_rows_copy, dim3(cdiv(n * D, 256)), dim3(256), 0, st, table, row_stride, col_off, rows_idx_dev, n, D, buf_dev, to_table);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_rows_apply(RowsApplyArgs a, int64_t n, hipStream_t st) {
    const int vec = (a.D % 4 == 0) ? 4 : 1;
    a.LPR = a.D / vec;
    if (n <= 0) return PS_OK;
    const int gpw = 64 / a.LPR;
    if (gpw < 1) return ps_set_err(PS_E_UNSUPPORTED, "embedding dim %d needs more than one wave per row", a.D);
    const int g = cdiv((int64_t)cdiv(n, gpw) * 64, 256);
    if (vec == 4) {
        if (a.identity) hipLaunchKernelGGL((k_rows_apply<4, true>), dim3(g), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_rows_apply<4, false>), dim3(g), dim3(256), 0, st, a);
    } else {
        if (a.identity) hipLaunchKernelGGL((k_rows_apply<1, true>), dim3(g), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_rows_apply<1, false>), dim3(g), dim3(256), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int launch_push_apply(PushApplyArgs a, hipStream_t st, LaunchOpts *lo, const PeerPutArgs *put) {
    if (lo) lo->launched = false;
    if (put && a.npeers < 2) return ps_set_err(PS_E_BAD_ARG, "the gradient put rides on the mark launch: two workers at least");
    if (a.n <= 0 && !put) return PS_OK;
    a.flag = lo ? lo->flag : nullptr; a.flag_val = lo ? lo->flag_val : 0u;
    const int vec = a.D % 4 == 0 ? 4 : 1;
    a.LPR = a.D / vec;
    const int gpw = 64 / a.LPR;
    if (gpw < 1) return ps_set_err(PS_E_UNSUPPORTED, "embedding dim %d needs more than one wave per row", a.D);
    const int g = cdiv((int64_t)cdiv(a.n, gpw) * 64, 256);
    if (a.npeers == 1) {        // one pushing worker: one launch (its rows are unique, nothing to mark)
        a.ts_apply = stamp_next("push_apply");
        if (vec == 4) hipLaunchKernelGGL((k_push_apply<4, true>), dim3(g), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_push_apply<1, true>), dim3(g), dim3(256), 0, st, a);
    } else {
        a.ts_mark = stamp_next("push_mark"); a.ts_apply = stamp_next("push_apply");
        if (put) hipLaunchKernelGGL(k_push_mark_put, dim3(PS_PUT_WGS + cdiv(a.n, 256)), dim3(256), 0, st, a, *put);
        else hipLaunchKernelGGL(k_push_mark, dim3(cdiv(a.n, 256)), dim3(256), 0, st, a);
        if (a.n <= 0) { HIPCHK(hipGetLastError()); if (lo) lo->launched = true; return PS_OK; }      // (nothing was pushed to this owner: the put alone)
        if (vec == 4) hipLaunchKernelGGL((k_push_apply<4, false>), dim3(g), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_push_apply<1, false>), dim3(g), dim3(256), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}

int launch_last_bwd(const LastBwdArgs &a, int nsplit, hipStream_t st) {
    if (a.B <= 0 || nsplit <= 0) return PS_OK;
    HeadArgs none;
    memset(&none, 0, sizeof none);
    hipLaunchKernelGGL(k_last_bwd<false>, dim3(nsplit), dim3(256), 0, st, a, none);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// head + the out = 1 layer's backward of the same rows in one launch (no loss reduction: launch_loss_reduce)
int launch_head_last_bwd(const HeadArgs &h, const LastBwdArgs &a, int nsplit, hipStream_t st, LaunchOpts *lo) {
    if (lo) lo->launched = false;
    if (a.B <= 0 || nsplit <= 0) return PS_OK;
    if (a.chunk > HEAD_ROWS_MAX || !h.labels) return ps_set_err(PS_E_BAD_ARG, "launch_head_last_bwd: %d rows per workgroup / no labels", a.chunk);
    LastBwdArgs q = a;
    q.ts = stamp_next("head_last_bwd");
    PS_LAUNCH_EV(k_last_bwd<true>, dim3(nsplit), dim3(256), 0, st, lo ? lo->stop_event : nullptr, q, h);
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}
int head_last_bwd_fusable(int rows_per_wg) { return rows_per_wg <= HEAD_ROWS_MAX; }

int launch_loss_reduce(const HeadArgs &a, float *loss_out, float *gbar_out, int *skip, int force_no_skip, hipStream_t st) {
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, st, a.terms, a.dlast, a.ldd, a.B, loss_out, gbar_out, skip, force_no_skip);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// Wt[n][k] = W'[k][n] for k <= K (rows of W' incl. the bias row), n < N  (rebuild of the transposed copy after a load)
namespace {
__global__ __launch_bounds__(256) void k_transpose_w(const float *__restrict__ W, float *__restrict__ Wt, int Kpad, int ldw, int N) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)Kpad * N) return;
    const int k = (int)(t / N), n = (int)(t % N);
    Wt[(size_t)n * Kpad + k] = W[(size_t)k * ldw + n];
}
}  // namespace

int launch_transpose_w(const float *W, float *Wt, int Kpad, int ldw, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_transpose_w, dim3(cdiv((int64_t)Kpad * N, 256)), dim3(256), 0, st, W, Wt, Kpad, ldw, N);
    HIPCHK(hipGetLastError());
    return PS_OK;
}

#ifdef PS_EMB_TIMING
extern "C" int ps_dbg_emb_timing(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_emb_t), sizeof(unsigned long long) * 8192 * 2) == hipSuccess ? 0 : -1;
}
#endif
