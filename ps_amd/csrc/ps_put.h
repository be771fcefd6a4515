// ps_put.h -- the mapped-peer put (ps_comm.hip, "MAPPED PEER"): argument block and device body, shared by the launch of its own (k_peer_put,
// ps_comm.hip) and the owner push's first launch, which carries the gradient put as a ROLE at N >= 2 (k_push_mark_put, kernels_emb.hip)
#pragma once
#include "ps_store.h"
typedef float mp_f32x4 __attribute__((ext_vector_type(4)));
struct PeerPutArgs {
    int npeers, rank, LPR, D, self;
    int bcast;                                   // every peer gets the SAME rows (the flat gradient): source row = row inside the peer's part
    int ablate;                                  // measurement only (ps_tune_set("mapped_ablate")): 1 no stores, 2 plain stores, 4 no flags / no wait, 8 no source loads
    const float *src;                            // rows grouped by destination peer, in peer order
    uint32_t start[PS_MAX_MAPPED + 1];           // first row of peer p's part
    float *dst[PS_MAX_MAPPED];                   // peer p's receive buffer (mapped)
    long long dst_row[PS_MAX_MAPPED];            // first row there
    unsigned int *flag_peer[PS_MAX_MAPPED];      // peer p's PS_PUT_WGS flag words for (this kind, this rank): one per workgroup of this launch
    const unsigned int *flag_mine;               // this rank's flag words of this kind: [p][PS_PUT_WGS] raised by peer p's workgroups
    unsigned int epoch;
    WaitBound bound;
    unsigned long long *ts;
};
// The launch is ALWAYS PS_PUT_WGS workgroups (every rank polls that many words per peer).  What the launch costs is a chain of
// memory round trips, not bytes (3 MB per exchange at configs[2]) -- so the chain is kept short:
//   source loads (one batch of PUT_ILP per thread in flight; the header word that says where a peer wants its rows is requested
//   in front of them and needed only behind them) -> write-through stores, drained -> THIS workgroup's flag word at every peer
//   (no arrival counter: a returned atomic per workgroup on one address serialises at the memory side, and a "last workgroup"
//   adds a round trip) -> workgroup 0 polls the peers' PS_PUT_WGS words each.
// First version (one workgroup per 256 parts, acq_rel counter, flags by the last workgroup): 22 us per exchange on one GPU, as
// slow as the grouped ncclSend / ncclRecv it replaces; relaxed counter 16; 64 workgroups 14.5; this one: see DESIGN.md 6.
#ifdef __HIPCC__
// (wg: this workgroup's index among the PS_PUT_WGS of the put -- blockIdx.x in a launch of its own, the role's index in k_push_mark_put)
__device__ __forceinline__ void peer_put_body(const PeerPutArgs &a, const unsigned int wg) {
    __shared__ uint32_t start_s[PS_MAX_MAPPED + 1];
    __shared__ float *dst_s[PS_MAX_MAPPED];
    __shared__ unsigned int *flag_s[PS_MAX_MAPPED];
    const int tid = threadIdx.x;
    // every peer's destination once per workgroup, by ONE thread with a uniform index: scalar loads of the argument block.  (Indexed
    // by the lane, an argument array is read with vector loads from the kernel-argument buffer -- with the header word behind a pointer
    // loaded that way the launch's floor was 8 us with every load, store and flag switched off: tools/r06_put_ablate.py.)
    if (tid == 0) {
        for (int p = 0; p < a.npeers; ++p) { start_s[p] = a.start[p]; dst_s[p] = a.dst[p] + (size_t)a.dst_row[p] * a.D; flag_s[p] = a.flag_peer[p]; }
        start_s[a.npeers] = a.start[a.npeers];
    }
    __syncthreads();
    constexpr int PUT_ILP = 8;
    const int64_t total = (int64_t)start_s[a.npeers] * a.LPR, T = (int64_t)PS_PUT_WGS * 256;
    bool first = true;
    for (int64_t t0 = (int64_t)wg * 256 + tid; first || t0 < total; t0 += T * PUT_ILP) {
        mp_f32x4 v[PUT_ILP];
        int64_t off[PUT_ILP];
        int pp[PUT_ILP];
#pragma unroll
        for (int j = 0; j < PUT_ILP; ++j) {
            const int64_t t = t0 + (int64_t)j * T;
            pp[j] = -1;
            if (t < total) {
                const int64_t i = t / a.LPR;
                const int part = (int)(t % a.LPR);
                int p = 0;
                while (p + 1 < a.npeers && (uint32_t)i >= start_s[p + 1]) ++p;
                if (p != a.rank || a.self) {
                    if (!(a.ablate & 8)) v[j] = *reinterpret_cast<const mp_f32x4 *>(a.src + (size_t)(a.bcast ? i - start_s[p] : i) * a.D + part * 4);
                    off[j] = (i - start_s[p]) * a.D + part * 4;
                    pp[j] = p;
                }
            }
        }
        first = false;
#pragma unroll
        for (int j = 0; j < PUT_ILP; ++j)
            if (pp[j] >= 0) {
                float *q = dst_s[pp[j]] + off[j];
                if (a.ablate & 1) continue;
                if (a.ablate & 2) { *reinterpret_cast<mp_f32x4 *>(q) = v[j]; continue; }
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v[j]) : "memory");       // write-through: nothing stays in this XCD's L2
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores have been acknowledged by the memory they went to
    __syncthreads();
    if (a.ablate & 4) return;
    // this workgroup's word at every peer: sc0 sc1 stores, issued behind the barrier = behind every wave's drain
    if (tid < a.npeers && (tid != a.rank || a.self))
        __hip_atomic_store(flag_s[tid] + wg, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (wg != 0) return;
    // workgroup 0, done with its own part: every peer's PS_PUT_WGS words for this exchange (bounded; the next launch of the stream
    // starts behind this kernel's end and acquires what the peers stored)
    for (int idx = tid; idx < a.npeers * PS_PUT_WGS; idx += 256) {
        const int p = idx / PS_PUT_WGS;
        if (p != a.rank || a.self) (void)spin_bounded_sys(a.flag_mine + idx, a.epoch, a.bound);
    }
}

#endif
