// ps_shard.hip -- the device-side halves of the parameter-server exchange when
// the KVStore is sharded over N GPUs (one ps_store_t per GPU, rows routed by
// id mod N exactly as net/PSRouterClient.java routes keys through net/Mod.java):
//
//   worker  PSRouterClient.getList  net/PSRouterClient.java:60-85   -> ps_shard_plan (unique keys grouped by owner)
//   owner   PServer.getList         net/PServer.java:102-117        -> ps_shard_serve_pull (row gather)
//   worker  KVStore cache + train   store/KVStore.java:96           -> ps_shard_forward_backward (rows read from the pulled cache)
//   worker  PSClient.push per key   net/PSClient.java:154-174       -> ps_shard_grads (per-key gradients, already in send order)
//   owner   PServer.push + psUpdate net/PServer.java:164-214        -> ps_shard_apply_push (mean over pushing workers, one updater step)
//   dense   one RPC per tensor      net/PSClient.java:47-70,154-174 -> ps_shard_flat_grad / ps_shard_apply_flat (one all-reduce)
//
// The wire (RCCL all-to-all-v / all-reduce over xGMI) belongs to the host:
// bench.py drives it through torch.distributed, a Java host through its own
// binding; the buffers handed over here are plain device pointers.
#include <string.h>

#include "ps_store.h"

namespace {

int bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && (1ll << b) < n) ++b;
    return b;
}

// composite sort key of every entry: owner << sbits | owner-local row.  One thread per bag.
__global__ __launch_bounds__(256) void k_shard_keys(const int64_t *__restrict__ ids, const int64_t *__restrict__ offsets,
                                                    int64_t nbags, int F, int nshards, int sbits,
                                                    const int64_t *__restrict__ lrb /*[nshards][F+1]*/,
                                                    const int64_t *__restrict__ vocab /*[F]*/,
                                                    const uint8_t *__restrict__ owner_tab, const uint32_t *__restrict__ local_tab,
                                                    const int64_t *__restrict__ grow_base /* java_string routing, else NULL */,
                                                    uint32_t *__restrict__ keys, uint32_t *__restrict__ ent_bag, int *err,
                                                    uint8_t *__restrict__ stamp /* or NULL */, uint8_t epoch, unsigned long long *ts) {
    StampScope stamp_scope(ts);
    const int64_t bag = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (bag >= nbags) return;
    const int f = (int)(bag % F);
    const int64_t p0 = offsets ? offsets[bag] : bag, p1 = offsets ? offsets[bag + 1] : bag + 1;
    for (int64_t p = p0; p < p1; ++p) {
        int64_t id = ids[p];
        if (id < 0 || id >= vocab[f]) { atomicAdd(err, 1); id = 0; }
        int o; int64_t local;
        if (owner_tab) {                                        // net/Mod.java: owner = hash of the key STRING
            const int64_t g = grow_base[f] + id;
            o = owner_tab[g];
            local = lrb[(size_t)o * (F + 1) + f] + local_tab[g];
        } else {
            o = (int)(id % nshards);
            local = lrb[(size_t)o * (F + 1) + f] + id / nshards;
        }
        const uint32_t key = ((uint32_t)o << sbits) | (uint32_t)local;
        keys[p] = key;
        if (stamp) stamp[key] = epoch;      // presence: every writer stores the same byte (no atomics: 2 000 samples sharing one
                                            // key made atomicOr on its bitmap word a 30 us serial chain)
        if (ent_bag) ent_bag[p] = (uint32_t)bag;
    }
}

// One launch for the tail of the plan (it is a chain of tiny kernels: every launch is ~5 us of serial time):
// thread i stores entry i's slot (its unique key's index) and, when i < nseg, handles unique key i: the
// owner-local row to request, and where each owner's run starts.  Writes every owner_start entry.
__global__ __launch_bounds__(256) void k_shard_finish(const uint32_t *__restrict__ sorted_key, const uint32_t *__restrict__ sorted_ent,
                                                      const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_id,
                                                      const uint32_t *__restrict__ nseg, int64_t nnz, int sbits, int nshards,
                                                      uint32_t *__restrict__ send_rows, uint32_t *__restrict__ owner_start,
                                                      uint32_t *__restrict__ slot) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    slot[sorted_ent[i]] = seg_id[i];
    const uint32_t n = *nseg, u = (uint32_t)i;
    if (u >= n) return;
    const uint32_t key = sorted_key[seg_start[u]];
    const uint32_t prev = u ? sorted_key[seg_start[u - 1]] : 0u;
    send_rows[u] = key & ((1u << sbits) - 1u);
    const int o = (int)(key >> sbits);
    const int po = u ? (int)(prev >> sbits) : -1;
    for (int oo = po + 1; oo <= o; ++oo) owner_start[oo] = u;     // owners without keys start where the next one does
    if (u == n - 1)
        for (int oo = o + 1; oo <= nshards; ++oo) owner_start[oo] = n;
}

// ---------------------------------------------------------------------------
// The plan WITHOUT a sort.  What the pull needs before anything else can move -- the batch's UNIQUE keys grouped by
// owner in ascending (owner, row) order, how many go to each owner, and the unique index ("slot") of every entry --
// is a presence bitmap over the composite key space (owner << sbits | row: 4 M bits = 512 KiB at configs[2]) and a
// prefix sum of its popcounts: four small launches (~15 us) instead of the 12 launches of a 3-pass radix sort
// (~80 us), which the sharded step used to pay serially before its pull.  The per-key ENTRY LISTS the embedding
// backward needs still come from the stable sort -- moved to side stream 0, where it overlaps the exchange and the
// forward exactly as in the single-GPU step; both orders are ascending composite key, so the unique indices agree.
// ---------------------------------------------------------------------------
constexpr int PLAN_WPB = 256;      // bitmap words per workgroup

// stamp bytes of this step's epoch -> bitmap word (32 keys per thread), popcount per workgroup
__global__ __launch_bounds__(256) void k_plan_count(const uint8_t *__restrict__ stamp, uint8_t epoch, uint32_t *__restrict__ bitmap,
                                                    int64_t nwords, uint32_t *__restrict__ blk_sum, unsigned long long *ts) {
    StampScope stamp_scope(ts);
    __shared__ uint32_t red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * PLAN_WPB + tid;
    uint32_t c = 0;
    if (i < nwords) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(stamp + i * 32), hi = *reinterpret_cast<const uint4 *>(stamp + i * 32 + 16);
        const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) bits |= (((q[k] >> (8 * b)) & 0xFFu) == (uint32_t)epoch ? 1u : 0u) << (4 * k + b);
        bitmap[i] = bits;
        c = (uint32_t)__popc(bits);
    }
    for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) red[w] = c;
    __syncthreads();
    if (tid == 0) blk_sum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// word_prefix[w] = set bits in all earlier words; every set bit emits its unique key's owner-local row at its rank;
// owner o's run starts at the prefix of word (o << sbits) >> 5 (sbits >= 5: owner boundaries are word boundaries)
__global__ __launch_bounds__(256) void k_plan_emit(const uint32_t *__restrict__ bitmap, int64_t nwords, const uint32_t *__restrict__ blk_sum,
                                                   uint32_t *__restrict__ word_prefix, uint32_t *__restrict__ send_rows,
                                                   uint32_t *__restrict__ owner_start, uint32_t *__restrict__ nseg, int sbits, int nshards, unsigned long long *ts) {
    StampScope stamp_scope(ts);
    __shared__ uint32_t red[4], wsum[4], carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t acc = 0;
    for (int i = tid; i < (int)blockIdx.x; i += 256) acc += blk_sum[i];
    for (int off = 32; off; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (tid == 0) carry_s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * PLAN_WPB;
    const uint32_t wmask = (1u << (sbits - 5)) - 1u;             // word index inside one owner's range
    for (int j = 0; j < PLAN_WPB / 256; ++j) {
        const int64_t i = base + j * 256 + tid;
        const uint32_t bits = i < nwords ? bitmap[i] : 0u;
        const uint32_t v = (uint32_t)__popc(bits);
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t wb = carry_s;
        for (int ww = 0; ww < w; ++ww) wb += wsum[ww];
        const uint32_t excl = wb + inc - v;
        if (i < nwords) {
            word_prefix[i] = excl;
            if (((uint32_t)i & wmask) == 0u) owner_start[(uint32_t)i >> (sbits - 5)] = excl;
            uint32_t b = bits, r = excl;
            while (b) {
                const int bit = __ffs((int)b) - 1;
                b &= b - 1;
                send_rows[r++] = (((uint32_t)i << 5) | (uint32_t)bit) & ((1u << sbits) - 1u);
            }
        }
        __syncthreads();
        if (tid == 255) carry_s = wb + inc;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) { owner_start[nshards] = carry_s; *nseg = carry_s; }
}

// Round 4: count + emit + pack in ONE launch (the sharded step's host enqueues ~29 launches per step).
// Every workgroup turns its 256 stamp words into bitmap words and popcounts, publishes its total in pub[b] tagged with this
// plan's sequence number (a 64-bit agent-scope store), and adds up the totals of the workgroups in front of it -- they were
// dispatched earlier, so spinning on their words cannot deadlock (the field sort's look-back, kernels_sort.hip) -- which
// gives it the rank of its first set bit and, from the same words, where its owner's run starts (an owner's range of the
// key space is a whole number of workgroups: sbits - 5 >= 8).  Then every set bit emits its owner-local row three times:
// send_rows (the plan's list), the owner's wire block when it fits, the owner's full block.  The last workgroup has every
// total: it writes owner_start, nseg and the blocks' headers [count | overflow].
constexpr int PLAN_FUSED_MAX_BLOCKS = 2048;
struct PlanFusedArgs {
    const uint8_t *stamp; uint8_t epoch;
    uint32_t *bitmap; int64_t nwords;
    unsigned long long *pub; uint32_t seq;
    uint32_t *word_prefix, *send_rows, *owner_start, *nseg;
    int sbits, nshards;
    // the id blocks (NULL: plan only)
    uint32_t *blk; int64_t blk_words; uint32_t blk_cap;
    uint32_t *full; int64_t full_words;
    unsigned long long *ts;
};
__global__ __launch_bounds__(256) void k_plan_fused(PlanFusedArgs a) {
    StampScope stamp_scope(a.ts);
    __shared__ uint32_t red[4], red2[4], wsum[4], carry_s, obase_s;
    __shared__ uint32_t tot_s[PLAN_FUSED_MAX_BLOCKS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = (int)blockIdx.x, nblk = (int)gridDim.x;
    const int64_t i = (int64_t)b * PLAN_WPB + tid;
    uint32_t bits = 0;
    if (i < a.nwords) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(a.stamp + i * 32), hi = *reinterpret_cast<const uint4 *>(a.stamp + i * 32 + 16);
        const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) bits |= (((q[k] >> (8 * bb)) & 0xFFu) == (uint32_t)a.epoch ? 1u : 0u) << (4 * k + bb);
        a.bitmap[i] = bits;
    }
    const uint32_t v = (uint32_t)__popc(bits);
    uint32_t inc = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tid == 0) __hip_atomic_store(&a.pub[b], ((unsigned long long)a.seq << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // look back: totals of the workgroups in front of this one; those in front of my owner's first workgroup give its run's start
    const int wpo_blocks = 1 << (a.sbits - 5 - 8);      // workgroups per owner
    const int owner = b / wpo_blocks, ob = owner * wpo_blocks;
    const bool last = b == nblk - 1;
    uint32_t acc_b = 0, acc_o = 0;
    for (int j = tid; j < b; j += 256) {
        unsigned long long x;
        do { x = __hip_atomic_load(&a.pub[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(x >> 32) != a.seq);
        const uint32_t t = (uint32_t)x;
        acc_b += t;
        if (j < ob) acc_o += t;
        if (last) tot_s[j] = t;
    }
    for (int off = 32; off; off >>= 1) { acc_b += __shfl_down(acc_b, off); acc_o += __shfl_down(acc_o, off); }
    if (lane == 0) { red[w] = acc_b; red2[w] = acc_o; }
    __syncthreads();
    if (tid == 0) { carry_s = red[0] + red[1] + red[2] + red[3]; obase_s = red2[0] + red2[1] + red2[2] + red2[3]; if (last) tot_s[b] = total; }
    __syncthreads();
    uint32_t wb = carry_s;
    for (int ww = 0; ww < w; ++ww) wb += wsum[ww];
    const uint32_t excl = wb + inc - v, obase = obase_s;
    if (i < a.nwords) {
        a.word_prefix[i] = excl;
        uint32_t bb = bits, r = excl;
        uint32_t *const blk_o = a.blk ? a.blk + (size_t)owner * a.blk_words + PS_BLK_HDR : nullptr;
        uint32_t *const full_o = (a.full && a.full != a.blk) ? a.full + (size_t)owner * a.full_words + PS_BLK_HDR : nullptr;
        while (bb) {
            const int bit = __ffs((int)bb) - 1;
            bb &= bb - 1;
            const uint32_t row = (((uint32_t)i << 5) | (uint32_t)bit) & ((1u << a.sbits) - 1u), pos = r - obase;
            a.send_rows[r] = row;
            if (blk_o && pos < a.blk_cap) blk_o[pos] = row;
            if (full_o) full_o[pos] = row;
            ++r;
        }
    }
    if (!last) return;
    // the last workgroup: owner_start[0..nshards], nseg, the blocks' headers
    __shared__ uint32_t os_s[PS_PUSH_MAX_PEERS + 1];
    __shared__ uint32_t ovf_s;
    if (tid == 0) ovf_s = 0u;
    for (int o = w; o <= a.nshards; o += 4) {           // one wave per owner boundary
        const int end = o * wpo_blocks < nblk ? o * wpo_blocks : nblk;
        uint32_t s = 0;
        for (int j = lane; j < end; j += 64) s += tot_s[j];
        for (int off = 32; off; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) os_s[o] = s;
    }
    __syncthreads();
    if (tid <= a.nshards) a.owner_start[tid] = os_s[tid];
    if (tid == 0) *a.nseg = os_s[a.nshards];
    if (a.blk) {
        if (tid < a.nshards && os_s[tid + 1] - os_s[tid] > a.blk_cap) atomicOr(&ovf_s, 1u);
        __syncthreads();
        if (tid < a.nshards) {
            const uint32_t cnt = os_s[tid + 1] - os_s[tid];
            // (word 2: this worker's first cache slot of owner tid's rows -- the mapped-peer pull stores them there, ps_comm.hip)
            a.blk[(size_t)tid * a.blk_words] = cnt; a.blk[(size_t)tid * a.blk_words + 1] = ovf_s; a.blk[(size_t)tid * a.blk_words + 2] = os_s[tid];
            if (a.full != a.blk) { a.full[(size_t)tid * a.full_words] = cnt; a.full[(size_t)tid * a.full_words + 1] = ovf_s; a.full[(size_t)tid * a.full_words + 2] = os_s[tid]; }
        }
    }
}

__global__ __launch_bounds__(256) void k_plan_slots(const uint32_t *__restrict__ keys, int64_t nnz, const uint32_t *__restrict__ bitmap,
                                                    const uint32_t *__restrict__ word_prefix, uint32_t *__restrict__ slot, unsigned long long *ts) {
    StampScope stamp_scope(ts);
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint32_t key = keys[p], wd = key >> 5;
    slot[p] = word_prefix[wd] + (uint32_t)__popc(bitmap[wd] & ((1u << (key & 31u)) - 1u));
}

// rows_out[i][:] = W[rows[i]][:]   (PServer.getList: the rows for a key list)
// The key list arrives grouped by requesting worker, every worker's part where the id exchange left it (PeerLists):
// entry i of worker p = rows_p[p][i - start[p]].
struct PeerLists {
    int npeers;
    uint32_t start[PS_PUSH_MAX_PEERS + 1];
    const uint32_t *rows_p[PS_PUSH_MAX_PEERS];
};
// end_wait: the first workgroup ends only once *end_wait reached end_val (the sharded step: "the slots of this step's
// plan are written" -- the gather is the launch in front of the forward, and holds the join with side chain 0 for it)
// gs.keys != NULL (round 5): workgroups [gather_blocks, gridDim.x) compute the slots of the step this gather serves (k_plan_slots'
// arithmetic) -- they wait at their start for "the plan head is done" (long since: it ran beside the previous step's forward)
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ W, PeerLists pl, int64_t n,
                                                     int D, int LPR, int64_t total_rows, float *__restrict__ out, int *err, unsigned long long *ts,
                                                     const unsigned int *end_wait, unsigned int end_val, WaitBound bound, GatherSlots gs, int gather_blocks, GatherPut gp) {
    EndWait end_wait_scope(end_wait, end_val, bound);
    StampScope stamp_scope(ts);
    if ((int)blockIdx.x >= gather_blocks) {
        start_wait(gs.wait, gs.wait_val, bound);
        const int64_t p = (int64_t)((int)blockIdx.x - gather_blocks) * 256 + threadIdx.x;
        if (p < gs.nnz) {
            const uint32_t key = gs.keys[p], wd = key >> 5;
            gs.slot[p] = gs.word_prefix[wd] + (uint32_t)__popc(gs.bitmap[wd] & ((1u << (key & 31u)) - 1u));
        }
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = t / LPR;
    const int part = (int)(t % LPR);
    if (!gp.on) {
        if (i >= n) return;
        int p = 0;
        while (p + 1 < pl.npeers && (uint32_t)i >= pl.start[p + 1]) ++p;
        int64_t r = pl.rows_p[p][i - pl.start[p]];
        if (r >= total_rows) { if (part == 0) atomicAdd(err, 1); r = 0; }
        if (VEC == 4) *reinterpret_cast<float4 *>(out + (size_t)i * D + part * 4) = *reinterpret_cast<const float4 *>(W + (size_t)r * D + part * 4);
        else out[(size_t)i * D + part] = W[(size_t)r * D + part];
        return;
    }
    // ---- mapped peer, fused (VEC == 4 only: the launcher checks): worker p's rows go straight into its cache ----
    typedef float gp_f32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t start_s[PS_PUSH_MAX_PEERS + 1];
    __shared__ const uint32_t *rows_s[PS_PUSH_MAX_PEERS];
    __shared__ float *dst_s[PS_PUSH_MAX_PEERS];
    __shared__ int last_s;
    if (threadIdx.x == 0) {             // one thread, uniform index: scalar loads of the argument block (k_peer_put's comment)
        for (int p = 0; p < pl.npeers; ++p) { start_s[p] = pl.start[p]; rows_s[p] = pl.rows_p[p]; dst_s[p] = gp.dst[p]; }
        start_s[pl.npeers] = pl.start[pl.npeers];
    }
    __syncthreads();
    if (i < n) {
        int p = 0;
        while (p + 1 < pl.npeers && (uint32_t)i >= start_s[p + 1]) ++p;
        int64_t r = rows_s[p][i - start_s[p]];
        if (r >= total_rows) { if (part == 0) atomicAdd(err, 1); r = 0; }
        const gp_f32x4 v = *reinterpret_cast<const gp_f32x4 *>(W + (size_t)r * D + part * 4);
        if (p != gp.rank || gp.self) {
            float *q = dst_s[p] + (size_t)(i - start_s[p]) * D + part * 4;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");      // write-through, like k_peer_put's
        } else {
            *reinterpret_cast<gp_f32x4 *>(out + (size_t)i * D + part * 4) = v;      // this rank's own rows: read in place by its forward
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        // (relaxed: the payload is write-through and drained -- nothing for a release to write back; ps_comm.hip k_peer_put)
        const unsigned int old = __hip_atomic_fetch_add(gp.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == (unsigned int)gather_blocks - 1u ? 1 : 0;
        if (last_s) __hip_atomic_store(gp.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_s) return;
    // the last gather workgroup: every workgroup's rows have landed -- this rank's PS_PUT_WGS words at every peer (the same words a
    // put launch's workgroups raise one by one: a peer polls them all, whichever form its sender used), then the peers' words here
    for (int idx = threadIdx.x; idx < pl.npeers * PS_PUT_WGS; idx += 256) {
        const int p = idx / PS_PUT_WGS;
        if (p != gp.rank || gp.self) __hip_atomic_store(gp.flag_peer[p] + (idx - p * PS_PUT_WGS), gp.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int idx = threadIdx.x; idx < pl.npeers * PS_PUT_WGS; idx += 256) {
        const int p = idx / PS_PUT_WGS;
        if (p != gp.rank || gp.self) (void)spin_bounded_sys(gp.flag_mine + idx, gp.epoch, bound);
    }
}

}  // namespace
int g_slots_in_gather = 1;      // ps_tune_set("slots_in_gather", 0): the plan's slot kernel behind a spinner on side chain 0 + a flag setter again (round 4)
int g_plan_sort = 0;       // ps_tune_set("plan_sort", 1): the sort-based plan (A/B runs, tests of both paths)
int g_shard_sort_defer = 1;      // ps_tune_set("shard_sort_defer", 0): the plan's field sort right behind its slots again (round 3)
int g_plan_fused = 1;      // ps_tune_set("plan_fused", 0): count / emit / pack as three launches (round 3)
namespace {

int ensure_push_ws(ps_store *s, int64_t n) {
    RtGuard rt_guard;
    if (n <= s->push_cap) return PS_OK;
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    sort_ws_free(s->push_ws);
    fr(s->push_keys); fr(s->push_ents); fr(s->push_seg_start); fr(s->push_seg_id); fr(s->push_nseg);
    s->push_keys = s->push_ents = s->push_seg_start = s->push_seg_id = s->push_nseg = nullptr;
    s->push_cap = 0;            // (a failed allocation below leaves an empty workspace, not a stale capacity)
    const int64_t cap = n + n / 4 + 1024;
    PSCHK(sort_ws_alloc(s->push_ws, cap));
    HIPCHK(hipMalloc((void **)&s->push_keys, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc((void **)&s->push_ents, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc((void **)&s->push_seg_start, sizeof(uint32_t) * (size_t)(cap + 2)));
    HIPCHK(hipMalloc((void **)&s->push_seg_id, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc((void **)&s->push_nseg, sizeof(uint32_t) * 4));
    s->push_cap = cap;
    return PS_OK;
}

int ensure_shard_state(ps_model *m, int nshards) {
    RtGuard rt_guard;
    ps_model::Shard &sh = m->sh;
    ps_store *s = m->s;
    if (sh.slot && sh.nshards == nshards) return PS_OK;
    if (sh.slot) return ps_set_err(PS_E_STATE, "the shard count of a model cannot change (%d -> %d)", sh.nshards, nshards);
    if (s->emb.nshards != nshards) return ps_set_err(PS_E_BAD_ARG, "store holds shard %d/%d, plan wants %d shards", s->emb.shard, s->emb.nshards, nshards);
    const int F = s->emb.F;
    sh.nshards = nshards;
    std::vector<int64_t> lrb((size_t)nshards * (F + 1), 0);
    int64_t maxrows = 1;
    for (int o = 0; o < nshards; ++o) {
        for (int f = 0; f < F; ++f) {
            const int64_t cnt = s->emb.java_route() ? s->emb.owner_cnt[(size_t)o * F + f]
                                                    : (s->emb.rows[f] > o ? (s->emb.rows[f] - o + nshards - 1) / nshards : 0);
            lrb[(size_t)o * (F + 1) + f + 1] = lrb[(size_t)o * (F + 1) + f] + cnt;
        }
        if (lrb[(size_t)o * (F + 1) + F] > maxrows) maxrows = lrb[(size_t)o * (F + 1) + F];
    }
    sh.sbits = bits_for(maxrows);
    if (sh.sbits + bits_for(nshards) > 32) return ps_set_err(PS_E_UNSUPPORTED, "owner and local row do not fit one 32-bit sort key");
    const int64_t nc = m->nnz_cap;
    PSCHK(store_dev_alloc(s, (void **)&sh.lrb_dev, sizeof(int64_t) * lrb.size() + sizeof(int64_t) * F, false));
    HIPCHK(hipMemcpyAsync(sh.lrb_dev, lrb.data(), sizeof(int64_t) * lrb.size(), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(sh.lrb_dev + lrb.size(), s->emb.rows.data(), sizeof(int64_t) * F, hipMemcpyHostToDevice, s->stream));
    if (s->emb.java_route() && !s->emb.owner_dev) {
        EmbTables &e = s->emb;
        const size_t G = e.owner_h.size();
        PSCHK(store_dev_alloc(s, (void **)&e.owner_dev, G + 16, false));
        PSCHK(store_dev_alloc(s, (void **)&e.local_dev, sizeof(uint32_t) * G + 16, false));
        PSCHK(store_dev_alloc(s, (void **)&e.grow_base_dev, sizeof(int64_t) * (size_t)(F + 1), false));
        HIPCHK(hipMemcpyAsync(e.owner_dev, e.owner_h.data(), G, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(e.local_dev, e.local_h.data(), sizeof(uint32_t) * G, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(e.grow_base_dev, e.grow_base.data(), sizeof(int64_t) * (size_t)(F + 1), hipMemcpyHostToDevice, s->stream));
    }
    {   // the sort-free plan: bitmap over the composite key space (falls back to the sort above 2^29 keys = 64 MiB of bitmap)
        const int64_t kspace = (int64_t)nshards << sh.sbits;
        if (sh.sbits >= 5 && kspace <= ((int64_t)1 << 29)) {
            sh.bm_words = kspace >> 5;
            PSCHK(store_dev_alloc(s, (void **)&sh.bitmap, sizeof(uint32_t) * (size_t)sh.bm_words, true));
            PSCHK(store_dev_alloc(s, (void **)&sh.stamp, (size_t)kspace + 32, true));
            PSCHK(store_dev_alloc(s, (void **)&sh.word_prefix, sizeof(uint32_t) * (size_t)sh.bm_words, false));
            PSCHK(store_dev_alloc(s, (void **)&sh.blk_sum, sizeof(uint32_t) * (size_t)(cdiv(sh.bm_words, PLAN_WPB) + 1), false));
            PSCHK(store_dev_alloc(s, (void **)&sh.plan_pub, sizeof(unsigned long long) * (size_t)(cdiv(sh.bm_words, PLAN_WPB) + 1), true));
        }
    }
    PSCHK(store_dev_alloc(s, (void **)&sh.slot, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(store_dev_alloc(s, (void **)&sh.keys2, sizeof(uint32_t) * (size_t)(nc + 1), false));      // (m->keys' twin: ps_store.h)
    PSCHK(store_dev_alloc(s, (void **)&sh.send_rows, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(store_dev_alloc(s, (void **)&sh.owner_start, sizeof(uint32_t) * (size_t)(nshards + 2), true));
    sh.flat_elems = m->dense_elems + (m->cfg.kind == PS_MODEL_WIDEDEEP ? 2 * s->wide.rows + 1 : 0);
    PSCHK(store_dev_alloc(s, (void **)&sh.flat, sizeof(float) * (size_t)(sh.flat_elems + 4), true));      // (+4: the mapped-peer reduction reads it in 16-byte rows)
    HIPCHK(hipHostMalloc((void **)&sh.owner_start_host, sizeof(uint32_t) * (size_t)(nshards + 2), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&sh.plan_ev, hipEventDisableTiming));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

}  // namespace

int shard_ensure_state(ps_model *m, int nshards) { return ensure_shard_state(m, nshards); }

extern "C" int ps_store_set_stream(ps_store_t *s, void *hip_stream) {
    // Adopt the host framework's stream (e.g. torch.cuda.current_stream().cuda_stream) so the
    // kernels here and the RCCL collectives the host enqueues are ordered without host syncs.
    if (!s) return ps_set_err(PS_E_BAD_ARG, "store is NULL");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->stream = (hipStream_t)hip_stream;      // the store's own stream is simply left idle
    return PS_OK;
}

// the plan's kernels; readback: also copy owner_start to pinned memory and record plan_ev (ps_shard_plan_finish)
// The slot of every entry (the forward reads it ~50 us later) and the entry lists of the backward (stable sort + runs),
// on side stream 0 beside the exchange and the forward.  k_plan_slots before the sort in stream order: the general
// sort ping-pongs through m->keys.
// can the plan's field sort leave plan_slots_and_lists for the step's forward (Shard::sort_due)?
static bool plan_sort_deferrable(const ps_model *m, hipStream_t ss) {
    const bool fsort = !m->cur_offsets && g_field_sort && field_sort_fits(m->cur_B, m->cfg.F);
    return fsort && ss != m->s->stream && m->dev_ok && g_shard_sort_defer && ss == m->side[0] && !m->cfg.use_graph && !m->profile && m->multi_stream;
}
// by_gather: no launch at all here -- the slots are computed by the next owner-side gather's launch (Shard::slots_due), the sort by
// the step's forward (the caller has checked plan_sort_deferrable)
static int plan_slots_and_lists(ps_model *m, int nshards, int64_t nnz, hipStream_t st, hipStream_t ss, bool by_gather = false) {
    ps_store *s = m->s;
    ps_model::Shard &sh = m->sh;
    const int F = m->cfg.F;
    const bool off_main = ss != s->stream;        // (the forward, on the training stream, waits for the slots)
    sh.slots_due = false;
    if (by_gather) {
        sh.slots_due = true; sh.slots_keys = m->keys; sh.slots_nnz = nnz;
        sh.slot_ev = nullptr;
    } else {
    if (off_main) sh.slot_ev = m->events[m->next_event++ % m->events.size()];
    // (the event the forward waits for rides on the slot kernel's launch: no record packet on the side chain)
    PS_LAUNCH_EV(k_plan_slots, dim3(cdiv(nnz, 256)), dim3(256), 0, ss, (off_main && g_ext_events) ? sh.slot_ev : nullptr, m->keys, nnz, sh.bitmap,
                 sh.word_prefix, sh.slot, stamp_next("plan_slots"));
    if (off_main && !g_ext_events) HIPCHK(hipEventRecord(sh.slot_ev, ss));
    }
    sh.slot_flag = false;
    const bool fsort = !m->cur_offsets && g_field_sort && field_sort_fits(m->cur_B, F);
    // the sort leaves this call when the step's forward can launch it behind its first GEMM's start (Shard::sort_due)
    const bool defer_sort = fsort && off_main && m->dev_ok && g_shard_sort_defer && ss == m->side[0] && !m->cfg.use_graph && !m->profile && m->multi_stream;
    if (by_gather && !defer_sort) return ps_set_err(PS_E_STATE, "the slots were left to the gather but the sort cannot be deferred");
    unsigned int *fs_flag = nullptr;
    if (off_main && m->dev_ok && !by_gather) {       // ... and a device flag: ps_shard_step hangs this join on its owner-side gather's launch
        if (++m->start_epoch == 0) ++m->start_epoch;
        sh.slot_epoch = m->start_epoch;
        // (raised by the START of the field sort behind the slot kernel -- in order, so the slots are written -- rather than
        //  by a flag-setter launch of its own; a deferred sort comes too late for that)
        if (fsort && !defer_sort) fs_flag = m->start_flag + 10;
        else PSCHK(launch_flag_set(m->start_flag + 10, sh.slot_epoch, ss));
        sh.slot_flag = true;
    }
    HIPCHK(hipGetLastError());
    m->field_sorted = false;
    sh.sort_due = false;
    if (defer_sort) {
        int kb = sh.sbits + bits_for(nshards);
        const bool based = nshards == 1 && !s->emb.java_route();
        if (based) {
            int64_t span = 1;
            for (int f = 0; f < F; ++f) span = std::max(span, s->emb.rows[f]);
            kb = bits_for(span);
        }
        sh.sort_due = true; sh.sort_kb = kb; sh.sort_based = based;
        m->sorted_keys = m->fs_keys; m->sorted_ents = m->fs_ents;
        m->field_sorted = true;
    } else if (fsort) {
        // single-hot: one launch sorts the F fields on their own (kernels_sort.hip) instead of the 11-launch radix
        // chain.  A composite key (owner, local row) belongs to one field only, so the runs are the same; they come
        // field by field rather than in send order, and the embedding backward writes each run's gradient at the
        // plan's slot of its first entry (EmbBwdArgs.out_slot).  One shard: the owner-0 field bases shorten the key.
        if (++m->fs_epoch == 0) ++m->fs_epoch;
        int kb = sh.sbits + bits_for(nshards);
        const bool based = nshards == 1 && !s->emb.java_route();
        if (based) {
            int64_t span = 1;
            for (int f = 0; f < F; ++f) span = std::max(span, s->emb.rows[f]);
            kb = bits_for(span);
        }
        PSCHK(field_sort_segments(m->keys, based ? sh.lrb_dev : nullptr, kb, m->cur_B, F, PS_EMB_SEQ_TILE, m->fs_keys, m->fs_ents,
                                  m->seg_start, m->seg_id, m->seg_nseg_scratch, m->long_list, m->fs_pub, m->fs_epoch, ss, fs_flag, sh.slot_epoch));
        m->sorted_keys = m->fs_keys; m->sorted_ents = m->fs_ents;
        m->field_sorted = true;
    } else {
        PSCHK(radix_sort_pairs(m->ws, m->keys, m->ents, nnz, sh.sbits + bits_for(nshards), true, &m->sorted_keys, &m->sorted_ents, ss));
        PSCHK(build_segments(m->ws, m->sorted_keys, nnz, m->seg_start, m->seg_id, m->seg_nseg_scratch, ss, m->long_list, PS_EMB_SEQ_TILE));
    }
    m->long_list_valid = true; m->nlong_ptr = m->seg_nseg_scratch + 1;
    m->side0_pending = off_main;
    return PS_OK;
}

// the plan's field sort, enqueued by the step's forward on side chain 0: behind a spinner on its first GEMM's start
// (behind_fwd_flag; the launch that raises start_flag[4] = fwd_epoch is already on the training stream), or at once
int shard_launch_deferred_sort(ps_model *m, bool behind_fwd_flag) {
    ps_model::Shard &sh = m->sh;
    if (!sh.sort_due) return PS_OK;
    sh.sort_due = false;
    hipStream_t ss = m->side[0];
    if (behind_fwd_flag) PSCHK(launch_spin_until(m->start_flag + 4, m->fwd_epoch, ss, m->s->werr(), 15));
    if (++m->fs_epoch == 0) ++m->fs_epoch;
    PSCHK(field_sort_segments(m->keys, sh.sort_based ? sh.lrb_dev : nullptr, sh.sort_kb, m->cur_B, m->cfg.F, PS_EMB_SEQ_TILE, m->fs_keys, m->fs_ents,
                              m->seg_start, m->seg_id, m->seg_nseg_scratch, m->long_list, m->fs_pub, m->fs_epoch, ss));
    m->side0_pending = true;
    return PS_OK;
}

int shard_flush_deferred_flag(ps_model *m) {
    ps_model::Shard &sh = m->sh;
    if (!sh.deferred) return PS_OK;
    sh.deferred = false;
    return launch_flag_set(sh.def_flag, sh.def_val, m->side[0]);
}

// early plans (see shard_plan_enqueue): the second half, enqueued by the caller once the launch that raises
// start_flag[6] = sh.pub_epoch ("the running step's embedding backward has finished") is on the main stream
int shard_plan_enqueue_tail(ps_model *m, int nshards, hipStream_t st) {
    ps_model::Shard &sh = m->sh;
    if (!sh.tail_due) return PS_OK;
    sh.tail_due = false;
    hipStream_t ss = m->side[0];
    // Round 5 (slots_in_gather): behind a plan head on the list chain nothing is launched here at all -- the slots are computed by
    // the next step's owner-side gather (in order behind the running step's backward and push on the training stream; its slot
    // workgroups wait for "plan head done" themselves), the field sort by that step's forward.  Three launches less per step.
    if (g_slots_in_gather && sh.head_on_list && sh.tail_flag_due && plan_sort_deferrable(m, ss)) {
        PSCHK(shard_flush_deferred_flag(m));          // (side chain 0's "small kernels done": a flag-setter launch of its own now)
        sh.tail_flag_due = false;                     // (nobody waits for the push's start any more)
        return plan_slots_and_lists(m, nshards, sh.tail_nnz, st, ss, true);
    }
    // (one launch: raises side chain 0's pending "small kernels done" -- it is in order behind them -- then waits for the running
    //  step's push and, when the plan head ran on the list chain, for that chain's "plan head done")
    unsigned int *set = nullptr; unsigned int set_val = 0;
    if (sh.deferred) { set = sh.def_flag; set_val = sh.def_val; sh.deferred = false; }
    PSCHK(launch_set_then_spin2(set, set_val, m->start_flag + 6, sh.pub_epoch, sh.head_on_list ? m->start_flag + 7 : nullptr, sh.plan_epoch, ss, m->s->werr(), 6));
    return plan_slots_and_lists(m, nshards, sh.tail_nnz, st, ss);
}

// the plan's kernels; readback: also copy owner_start to pinned memory and record plan_ev (ps_shard_plan_finish).
// early (ps_shard_step's pipeline, called between a step's backward and its push): the kernels that only read the ids
// -- keys, presence map, unique lists, counts -- go to side chain 0 behind a spinner on "the running step's first forward
// GEMM has started" (its exchanges, which read the previous plan's lists, are done then), so they run while that step
// trains instead of in its tail; the main stream only parks a spinner on "plan done".  What overwrites lists the
// running backward still reads (slots, entry lists) is enqueued later by shard_plan_enqueue_tail; the run count is
// double-buffered (nseg_cur).
// hook (round 5, ps_comm.hip shard_step_begin_hook): called BETWEEN the forward and the backward of the running step, for a
// device-resident single-hot batch that shard_plan_hook_ok accepted -- the running step's batch stays staged (m->cur_*: its
// backward is not enqueued yet), the plan takes its inputs from `batch` itself, and its kernels go to the list chain side[2].
bool shard_plan_hook_ok(const ps_model *m, const ps_batch_t *batch) {
    const ps_model::Shard &sh = m->sh;
    return g_plan_mid && g_plan_early && batch && batch->on_device && !batch->offsets && batch->B > 0 && batch->B <= m->Bcap && batch->ids &&
           (int64_t)batch->B * m->cfg.F <= m->nnz_cap && sh.bitmap && sh.keys2 && g_plan_sort == 0 && m->dev_ok && dev_waits_ok(m->s) &&
           !m->cfg.use_graph && !m->profile && m->multi_stream && g_field_sort && field_sort_fits(batch->B, m->cfg.F) && g_plan_fused && sh.plan_pub &&
           cdiv(sh.bm_words, PLAN_WPB) <= PLAN_FUSED_MAX_BLOCKS && sh.sbits - 5 >= 8 && sh.nshards <= PS_PUSH_MAX_PEERS && m->side[2];
}
int shard_plan_enqueue(ps_model *m, const ps_batch_t *batch, int nshards, hipStream_t st, bool readback, bool early, bool order_after_main, bool hook) {
    ps_store *s = m->s;
    PSCHK(ensure_shard_state(m, nshards));
    ps_model::Shard &sh = m->sh;
    const bool fwd_flag = m->fwd_flag_valid;       // (of the step enqueued before this call)
    m->fwd_flag_valid = false;
    // side chain 0's pending "small kernels done" flag (ps_store.h defer_flag5) rides on the opening spinner of an EARLY plan that
    // goes to side chain 0, or on its tail's spinner there when the head goes to the list chain.  Any other plan raises it first,
    // before anything here can wait for the training stream (a host batch's staging does; the ordering event of a late plan does)
    // -- the running step's last delta GEMM holds its slot until that flag is up.
    if (sh.deferred && !hook) {
        const bool will_be_early = early && batch && batch->on_device && !batch->offsets && batch->B > 0 && sh.bitmap && g_plan_sort == 0 && fwd_flag &&
                                   g_plan_early && dev_waits_ok(s) && !m->cfg.use_graph && !m->profile && m->multi_stream && !readback && g_field_sort &&
                                   field_sort_fits(batch->B, m->cfg.F);
        if (!will_be_early) PSCHK(shard_flush_deferred_flag(m));
    }
    if (!hook) {
        PSCHK(stage_batch(m, batch, true));
        if (st != s->stream && !batch->on_device) HIPCHK(hipStreamSynchronize(s->stream));   // host batch: uploads ran on the store's stream
    }
    const int F = m->cfg.F;
    // (hook: a device-resident single-hot batch, checked by shard_plan_hook_ok -- what stage_batch would have made of it)
    const int plan_B = hook ? batch->B : m->cur_B;
    const int64_t *const plan_ids = hook ? batch->ids : m->cur_ids, *const plan_offsets = hook ? nullptr : m->cur_offsets;
    const int64_t nbags = (int64_t)plan_B * F, nnz = hook ? nbags : m->cur_nnz;
    sh.plan_nnz = nnz;
    const bool bm = sh.bitmap != nullptr && nnz > 0 && g_plan_sort == 0;
    early = early && bm && fwd_flag && g_plan_early && m->dev_ok && !m->cfg.use_graph && !m->profile && m->multi_stream && !readback &&
            batch->on_device && !plan_offsets && g_field_sort && field_sort_fits(plan_B, F);
    static const bool plan_debug = getenv("PS_PLAN_DEBUG") != nullptr;      // measurement: which way did the plan go
    if (plan_debug) fprintf(stderr, "[plan] early=%d hook=%d (bitmap %d, fwd flag %d, device batch %d, single-hot %d, field sort fits %d)\n", (int)early, (int)hook, (int)bm,
                            (int)fwd_flag, (int)batch->on_device, (int)!plan_offsets, (int)field_sort_fits(plan_B, F));
    if (hook && !early) return ps_set_err(PS_E_STATE, "the plan head was handed to the forward's hook but cannot run early");
    // where the id-only kernels go: an early plan runs beside the step that trains -- on the list chain (the caller's stream in
    // overlap mode: the id exchange follows in order, nothing waits across streams), else on side chain 0
    const bool on_list = early && st == m->side[2] && m->side[2] != nullptr;
    hipStream_t ps = early ? (on_list ? st : m->side[0]) : st;
    sh.head_on_list = on_list;
    if (!early) PSCHK(shard_flush_deferred_flag(m));      // (normally flushed above already)
    if (!early && order_after_main && st != s->stream) {
        // (not early, on another stream than the training stream: the plan overwrites lists the running step's backward
        //  still reads -- behind everything enqueued on the training stream so far)
        hipEvent_t e = m->events[m->next_event++ % m->events.size()];
        HIPCHK(hipEventRecord(e, s->stream));
        HIPCHK(hipStreamWaitEvent(st, e, 0));
    }
    if (early && sh.deferred && ps == m->side[0]) {       // (side chain 0's pending "small kernels done" rides on this spinner's launch)
        sh.deferred = false;
        PSCHK(launch_set_then_spin(sh.def_flag, sh.def_val, m->start_flag + 4, m->fwd_epoch, ps, s->werr(), 14));
    } else {
        if (!on_list) PSCHK(shard_flush_deferred_flag(m));      // (on the list chain: the tail's spinner on side chain 0 takes it)
        if (early) PSCHK(launch_spin_until(m->start_flag + 4, m->fwd_epoch, ps, s->werr(), 14));
    }
    m->nseg_cur = early ? (m->nseg_cur == m->nseg_dev ? m->nseg_dev + 4 : m->nseg_dev) : m->nseg_dev;
    if (bm && ++sh.epoch == 0) {          // the byte stamps wrap every 255 plans: start over from a clean map
        HIPCHK(hipMemsetAsync(sh.stamp, 0, (size_t)sh.bm_words * 32, ps));
        sh.epoch = 1;
    }
    // On the list chain the keys are written while the running step's field sort (side chain 0, beside its forward) may still read
    // its own: the other buffer.  (Every reader takes the pointer when it is enqueued: the sort of the step this plan is for, its
    // slot kernel -- both enqueued after this swap.)
    if (on_list) std::swap(m->keys, sh.keys2);
    hipLaunchKernelGGL(k_shard_keys, dim3(cdiv(nbags, 256)), dim3(256), 0, ps, plan_ids, plan_offsets, nbags, F, nshards,
                       sh.sbits, sh.lrb_dev, sh.lrb_dev + (size_t)nshards * (F + 1), s->emb.owner_dev, s->emb.local_dev, s->emb.grow_base_dev, m->keys,
                       plan_offsets ? m->ent_bag : (uint32_t *)nullptr, s->err_dev, bm ? sh.stamp : (uint8_t *)nullptr, sh.epoch, stamp_next("shard_keys"));
    HIPCHK(hipGetLastError());
    sh.packed = false;
    if (bm) {
        const int nblk = cdiv(sh.bm_words, PLAN_WPB);
        // count + emit (+ the id blocks of ps_shard_step) in one launch when the key space allows the look-back
        // (its last workgroup keeps the owner boundaries in os_s[PS_PUSH_MAX_PEERS + 1] and writes them with tid <= nshards:
        //  more owners than that -- ps_shard_plan_launch takes any shard count, stores shard up to 255 ways -- go through
        //  k_plan_count + k_plan_emit, which handle any count)
        const bool fused = g_plan_fused && nblk <= PLAN_FUSED_MAX_BLOCKS && sh.sbits - 5 >= 8 && sh.plan_pub && nshards <= PS_PUSH_MAX_PEERS;
        PlanFusedArgs fa;
        memset(&fa, 0, sizeof fa);
        if (fused) {
            if (++sh.plan_seq == 0) ++sh.plan_seq;
            fa.stamp = sh.stamp; fa.epoch = sh.epoch; fa.bitmap = sh.bitmap; fa.nwords = sh.bm_words; fa.pub = sh.plan_pub; fa.seq = sh.plan_seq;
            fa.word_prefix = sh.word_prefix; fa.send_rows = sh.send_rows; fa.owner_start = sh.owner_start; fa.nseg = m->nseg_cur;
            fa.sbits = sh.sbits; fa.nshards = nshards;
            fa.blk = sh.pack_blk; fa.blk_words = sh.blk_words; fa.blk_cap = (uint32_t)sh.blk_cap; fa.full = sh.pack_full; fa.full_words = sh.full_words;
            fa.ts = stamp_next("plan_fused");
            sh.packed = sh.pack_blk != nullptr;
        } else
        hipLaunchKernelGGL(k_plan_count, dim3(nblk), dim3(256), 0, ps, sh.stamp, sh.epoch, sh.bitmap, sh.bm_words, sh.blk_sum, stamp_next("plan_count"));
        if (early) {
            if (fused) hipLaunchKernelGGL(k_plan_fused, dim3(nblk), dim3(256), 0, ps, fa);
            else
            hipLaunchKernelGGL(k_plan_emit, dim3(nblk), dim3(256), 0, ps, sh.bitmap, sh.bm_words, sh.blk_sum, sh.word_prefix, sh.send_rows,
                               sh.owner_start, m->nseg_cur, sh.sbits, nshards, stamp_next("plan_emit"));
            HIPCHK(hipGetLastError());
            if (++m->start_epoch == 0) ++m->start_epoch;
            sh.plan_epoch = m->start_epoch;
            if (st != ps) {         // (the caller continues on another stream: park it behind "plan done")
                PSCHK(launch_flag_set(m->start_flag + 7, sh.plan_epoch, ps));
                PSCHK(launch_spin_until(m->start_flag + 7, sh.plan_epoch, st, s->werr(), 7));      // (enqueued after the launch that releases it)
            }
            sh.tail_due = true; sh.tail_nnz = nnz;
            return PS_OK;
        }
        // Everything after the emit leaves the main chain (plan_slots_and_lists).  The emit's launch carries the event the
        // side stream waits for, the slot kernel's launch the one the forward waits for (no record packets on either chain).
        hipStream_t ss = (m->profile || !m->multi_stream) ? st : m->side[0];
        hipEvent_t e1 = nullptr;
        if (ss != st) e1 = m->events[m->next_event++ % m->events.size()];
        if (fused) PS_LAUNCH_EV(k_plan_fused, dim3(nblk), dim3(256), 0, st, g_ext_events ? e1 : nullptr, fa);
        else
        PS_LAUNCH_EV(k_plan_emit, dim3(nblk), dim3(256), 0, st, g_ext_events ? e1 : nullptr, sh.bitmap, sh.bm_words, sh.blk_sum, sh.word_prefix,
                     sh.send_rows, sh.owner_start, m->nseg_cur, sh.sbits, nshards, stamp_next("plan_emit"));
        if (ss != st) {
            if (!g_ext_events) HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipStreamWaitEvent(ss, e1, 0));
        }
        PSCHK(plan_slots_and_lists(m, nshards, nnz, st, ss));
    } else {
        PSCHK(radix_sort_pairs(m->ws, m->keys, m->ents, nnz, sh.sbits + bits_for(nshards), true, &m->sorted_keys, &m->sorted_ents, st));
        PSCHK(build_segments(m->ws, m->sorted_keys, nnz, m->seg_start, m->seg_id, m->nseg_dev, st, m->long_list, PS_EMB_SEQ_TILE));
        m->long_list_valid = true; m->nlong_ptr = m->nseg_dev + 1; m->field_sorted = false;
    }
    if (bm) {
        // (send_rows, owner_start, slot, nseg came from the bitmap)
    } else if (nnz > 0) {
        hipLaunchKernelGGL(k_shard_finish, dim3(cdiv(nnz, 256)), dim3(256), 0, st, m->sorted_keys, m->sorted_ents, m->seg_start, m->seg_id,
                           m->nseg_dev, nnz, sh.sbits, nshards, sh.send_rows, sh.owner_start, sh.slot);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemsetAsync(sh.owner_start, 0, sizeof(uint32_t) * (nshards + 2), st));
    }
    if (readback) {
        HIPCHK(hipMemcpyAsync(sh.owner_start_host, sh.owner_start, sizeof(uint32_t) * (nshards + 1), hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(sh.plan_ev, st));
        sh.plan_pending = true;
    }
    return PS_OK;
}

extern "C" int ps_shard_plan_launch(ps_model_t *m, const ps_batch_t *batch, int nshards, void *hip_stream) {
    RoctxRange roctx_range("ps_shard_plan_launch");
    if (!m || !batch || nshards < 1) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    PSCHK(store_enter(m->s));
    return shard_plan_enqueue(m, batch, nshards, hip_stream ? (hipStream_t)hip_stream : m->s->stream, true);
}

extern "C" int ps_shard_plan_finish(ps_model_t *m, int64_t *counts_out, uint32_t **send_rows_dev, int64_t *n_unique) {
    if (!m || !counts_out) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    ps_model::Shard &sh = m->sh;
    if (!sh.plan_pending) return ps_set_err(PS_E_STATE, "ps_shard_plan_launch first");
    PSCHK(store_enter(m->s));
    HIPCHK(hipEventSynchronize(sh.plan_ev));
    sh.plan_pending = false;
    for (int o = 0; o < sh.nshards; ++o) counts_out[o] = (int64_t)sh.owner_start_host[o + 1] - (int64_t)sh.owner_start_host[o];
    sh.U = sh.owner_start_host[sh.nshards];
    if (send_rows_dev) *send_rows_dev = sh.send_rows;
    if (n_unique) *n_unique = sh.U;
    return PS_OK;
}

extern "C" int ps_shard_plan(ps_model_t *m, const ps_batch_t *batch, int nshards, void *hip_stream,
                             int64_t *counts_out, uint32_t **send_rows_dev, int64_t *n_unique) {
    if (!counts_out) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    PSCHK(ps_shard_plan_launch(m, batch, nshards, hip_stream));
    return ps_shard_plan_finish(m, counts_out, send_rows_dev, n_unique);
}

// PServer.getList for key lists that lie grouped by requesting worker (rows_p[p]: counts[p] owner-local rows)
int shard_serve_pull_lists(ps_store *s, const uint32_t *const *rows_p, const int64_t *counts, int npeers, float *rows_out_dev, LaunchOpts *lo, const GatherSlots *gs, const GatherPut *gp) {
    if (lo) lo->launched = false;
    if (!s->emb.W) return ps_set_err(PS_MISSING, "no embedding tables");
    if (npeers < 1 || npeers > PS_PUSH_MAX_PEERS) return ps_set_err(PS_E_BAD_ARG, "1..%d workers", PS_PUSH_MAX_PEERS);
    PeerLists pl;
    memset(&pl, 0, sizeof pl);
    pl.npeers = npeers;
    int64_t n = 0;
    for (int p = 0; p < npeers; ++p) { pl.start[p] = (uint32_t)n; pl.rows_p[p] = rows_p[p]; n += counts[p]; }
    pl.start[npeers] = (uint32_t)n;
    GatherSlots g0;
    memset(&g0, 0, sizeof g0);
    if (gs && gs->keys && gs->nnz > 0) g0 = *gs;
    const bool slots = g0.keys != nullptr;
    if (n == 0 && !slots) return PS_OK;
    if (n == 0) {       // (nothing to gather: the slot kernel on its own, behind the wait its workgroups would have held)
        if (g0.wait) PSCHK(launch_spin_until(g0.wait, g0.wait_val, s->stream, s->werr(), 19));
        hipLaunchKernelGGL(k_plan_slots, dim3(cdiv(g0.nnz, 256)), dim3(256), 0, s->stream, g0.keys, g0.nnz, g0.bitmap, g0.word_prefix, g0.slot, stamp_next("plan_slots"));
        HIPCHK(hipGetLastError());
        return PS_OK;
    }
    const int D = s->emb.D, vec = D % 4 == 0 ? 4 : 1, LPR = D / vec;
    GatherPut gp0;
    memset(&gp0, 0, sizeof gp0);
    if (gp && gp->on && vec == 4) gp0 = *gp;
    else if (gp && gp->on) return ps_set_err(PS_E_UNSUPPORTED, "the fused mapped-peer gather needs D %% 4 == 0");
    const unsigned int *ew = lo ? lo->wait : nullptr;
    const unsigned int ev = lo ? lo->wait_val : 0u;
    const WaitBound wb = wait_bound(s->werr(), 104);
    const int gb = (int)cdiv(n * LPR, 256), sb = slots ? (int)cdiv(g0.nnz, 256) : 0;
    if (vec == 4)
        hipLaunchKernelGGL(k_gather_rows<4>, dim3(gb + sb), dim3(256), 0, s->stream, s->emb.W, pl, n, D, LPR, s->emb.total_rows, rows_out_dev, s->err_dev, stamp_next("gather_rows"), ew, ev, wb, g0, gb, gp0);
    else
        hipLaunchKernelGGL(k_gather_rows<1>, dim3(gb + sb), dim3(256), 0, s->stream, s->emb.W, pl, n, D, LPR, s->emb.total_rows, rows_out_dev, s->err_dev, stamp_next("gather_rows"), ew, ev, wb, g0, gb, gp0);
    HIPCHK(hipGetLastError());
    if (lo) lo->launched = true;
    return PS_OK;
}

extern "C" int ps_shard_serve_pull(ps_store_t *s, const uint32_t *rows_dev, int64_t n, float *rows_out_dev) {
    RoctxRange roctx_range("ps_shard_serve_pull");
    if (!s || n < 0 || (n > 0 && (!rows_dev || !rows_out_dev))) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (!s->emb.W) return ps_set_err(PS_MISSING, "no embedding tables");
    if (n == 0) return PS_OK;
    PSCHK(store_enter(s));
    return shard_serve_pull_lists(s, &rows_dev, &n, 1, rows_out_dev, nullptr);
}

extern "C" int ps_shard_forward_backward(ps_model_t *m, const float *cache_dev, float *loss) {
    RoctxRange roctx_range("ps_shard_forward_backward");
    if (!m || (!cache_dev && m->sh.U > 0)) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (!m->sh.slot) return ps_set_err(PS_E_STATE, "ps_shard_plan first");
    ps_store *s = m->s;
    PSCHK(store_enter(s));
    m->sh.active = true;
    m->sh.cache = cache_dev;
    if (m->sh.slot_ev) { HIPCHK(hipStreamWaitEvent(s->stream, m->sh.slot_ev, 0)); m->sh.slot_ev = nullptr; }   // the plan's slots (side stream)
    int rc = enqueue_forward(m, true, true);
    // (ps_shard_step's pipeline: the NEXT step's plan head goes to the list chain here -- behind this forward's launches, whose
    //  first GEMM releases it, and in front of the backward's, so that the host enqueues it ~60 us earlier than behind them)
    // (the plan head flips the run-count buffer, nseg_cur, to the NEXT plan's: this step's backward, enqueued below, still reads its own)
    uint32_t *const nseg_this = m->nseg_cur;
    if (rc == PS_OK && m->sh.hook_batch) rc = shard_step_begin_hook(m);
    m->sh.hook_batch = nullptr; m->sh.hook_comm = nullptr;
    uint32_t *const nseg_next = m->nseg_cur;
    m->nseg_cur = nseg_this;
    if (rc == PS_OK) rc = enqueue_backward(m, false);     // gradients only: the owners apply them (the flat buffer's wide part
                                                          // [G | C | bias] is filled by the dense gradient's launch)
    m->nseg_cur = nseg_next;
    m->sh.active = false;
    PSCHK(rc);
    m->fwd_done = true; m->bwd_done = true;
    return finish_step(m, loss);
}

extern "C" int ps_shard_grads(ps_model_t *m, float **grads_dev, int64_t *n_unique) {
    if (!m || !grads_dev) return ps_set_err(PS_E_BAD_ARG, "null argument");
    if (!m->bwd_done) return ps_set_err(PS_E_STATE, "no gradients: ps_shard_forward_backward first");
    *grads_dev = m->grads_out;          // [U][D], in the order of ps_shard_plan's send_rows
    if (n_unique) *n_unique = m->sh.U;
    return PS_OK;
}

// the sort-free push's per-row mask and per-worker position tables (allocated once: never between two collectives)
int shard_push_reserve(ps_store *s, int npeers) {
    const int64_t R = s->emb.total_rows;
    if (npeers <= 1 || !shard_push_grouped_ok(s, npeers)) return PS_OK;   // one worker: no mask / position tables; else the sorted path
    if (s->push_mask && s->push_pos_peers >= npeers) return PS_OK;
    RtGuard rt_guard;
    hipStream_t st = s->stream;
    if (!s->push_mask) {
        HIPCHK(hipMalloc((void **)&s->push_mask, sizeof(uint32_t) * (size_t)(R + 1)));
        HIPCHK(hipMemsetAsync(s->push_mask, 0, sizeof(uint32_t) * (size_t)(R + 1), st));
    }
    if (s->push_pos_peers < npeers) {
        if (s->push_pos) { HIPCHK(hipStreamSynchronize(st)); (void)hipFree(s->push_pos); s->push_pos = nullptr; }
        HIPCHK(hipMalloc((void **)&s->push_pos, sizeof(uint32_t) * (size_t)npeers * (size_t)(R + 1)));
        s->push_pos_peers = npeers;
    }
    return PS_OK;
}

extern "C" int ps_shard_apply_push(ps_store_t *s, const uint32_t *rows_dev, const float *grads_dev, int64_t n,
                                   const int64_t *peer_counts, int npeers, int is_async) {
    return shard_apply_push(s, rows_dev, grads_dev, n, peer_counts, npeers, is_async, true);
}

// can the sort-free push (worker-grouped lists, kernels_emb.hip k_push_mark / k_push_apply) serve this store?
int g_push_grouped_max_mb = 4000;        // ps_tune_set("push_grouped_max_mb"): the position table's size limit (tests: force the sorted push)
bool shard_push_grouped_ok(const ps_store *s, int npeers) {
    if (npeers < 1 || npeers > PS_PUSH_MAX_PEERS) return false;
    if (npeers == 1) return true;         // one worker: no mark pass, no tables
    return (double)npeers * (double)s->emb.total_rows * 4.0 <= 1.0e6 * (double)g_push_grouped_max_mb;
}

// PServer.push + psUpdate for lists that lie grouped by pushing worker, every worker's part where it is (rows_p[p],
// grads_p[p]: counts[p] entries); lo: the first launch announces its start (LaunchOpts.flag)
int shard_apply_push_lists(ps_store *s, const uint32_t *const *rows_p, const float *const *grads_p, const int64_t *counts, int npeers,
                           int is_async, bool bump_step, LaunchOpts *lo, const PeerPutArgs *put) {
    if (lo) lo->launched = false;
    if (!s->emb.W || !s->emb.state) return ps_set_err(PS_MISSING, "no embedding tables with updater state");
    // (tables reserved earlier -- a model that decided for this push at its first begin -- stay good whatever the knob says now)
    const bool reserved = npeers >= 1 && npeers <= PS_PUSH_MAX_PEERS && (npeers == 1 || (s->push_mask && s->push_pos_peers >= npeers));
    if (!reserved && !shard_push_grouped_ok(s, npeers)) return ps_set_err(PS_E_UNSUPPORTED, "the sort-free push needs 1..%d workers and a position table under 4 GB", PS_PUSH_MAX_PEERS);
    PushApplyArgs a;
    memset(&a, 0, sizeof a);
    PSCHK(store_fill_field_upd(s, &a.upd, &a.fu));
    int64_t n = 0;
    for (int p = 0; p < npeers; ++p) {
        if (counts[p] < 0) return ps_set_err(PS_E_BAD_ARG, "negative peer count");
        a.peer_start[p] = (uint32_t)n; a.rows_p[p] = rows_p[p]; a.grads_p[p] = grads_p[p];
        n += counts[p];
    }
    a.peer_start[npeers] = (uint32_t)n;
    if (n > 0 || put) {        // (put: this rank's gradient put rides on the launch even when nobody pushed anything to this owner)
        if (npeers > 1 && !reserved) PSCHK(shard_push_reserve(s, npeers));      // a no-op after the first step (ps_shard_step_begin reserves up front)
        a.D = s->emb.D; a.is_async = is_async ? 1 : 0; a.npeers = npeers; a.n = n; a.R = s->emb.total_rows;
        a.mask = s->push_mask; a.pos = s->push_pos;
        a.W = s->emb.W; a.state = s->emb.state; a.err = s->err_dev;
        PSCHK(launch_push_apply(a, s->stream, lo, put));
    }
    if (bump_step) s->global_step++;     // psUpdate: globalStep.incrementAndGet()  (net/PServer.java:213)
    return PS_OK;
}

int shard_apply_push(ps_store *s, const uint32_t *rows_dev, const float *grads_dev, int64_t n, const int64_t *peer_counts,
                     int npeers, int is_async, bool bump_step) {
    RoctxRange roctx_range("ps_shard_apply_push");
    if (!s || n < 0 || (n > 0 && (!rows_dev || !grads_dev))) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (!s->emb.W || !s->emb.state) return ps_set_err(PS_MISSING, "no embedding tables with updater state");
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    const int64_t R = s->emb.total_rows;
    // worker-grouped lists (what ps_shard_plan + all-to-all-v deliver): no sort, two kernels
    const bool grouped = peer_counts && shard_push_grouped_ok(s, npeers);
    if (n > 0 && grouped) {
        int64_t tot = 0;
        const uint32_t *rp[PS_PUSH_MAX_PEERS];
        const float *gp[PS_PUSH_MAX_PEERS];
        for (int p = 0; p < npeers; ++p) {
            if (peer_counts[p] < 0) return ps_set_err(PS_E_BAD_ARG, "negative peer count");
            rp[p] = rows_dev + tot; gp[p] = grads_dev + (size_t)tot * s->emb.D;
            tot += peer_counts[p];
        }
        if (tot != n) return ps_set_err(PS_E_BAD_ARG, "peer counts sum to %lld, n is %lld", (long long)tot, (long long)n);
        return shard_apply_push_lists(s, rp, gp, peer_counts, npeers, is_async, bump_step, nullptr);
    } else if (n > 0) {
        PSCHK(ensure_push_ws(s, n));
        // stable sort by row: within a key the pushes stay in arrival (= source worker) order
        HIPCHK(hipMemcpyAsync(s->push_keys, rows_dev, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st));
        uint32_t *sk = nullptr, *se = nullptr;
        PSCHK(radix_sort_pairs(s->push_ws, s->push_keys, s->push_ents, n, bits_for(R), true, &sk, &se, st));
        PSCHK(build_segments(s->push_ws, sk, n, s->push_seg_start, s->push_seg_id, s->push_nseg, st));
        RowsApplyArgs r;
        memset(&r, 0, sizeof r);
        r.D = s->emb.D; r.is_async = is_async ? 1 : 0; r.identity = 0;
        r.sorted_key = sk; r.sorted_ent = se; r.seg_start = s->push_seg_start; r.nseg = s->push_nseg;
        r.grads = grads_dev; r.W = s->emb.W; r.state = s->emb.state;
        PSCHK(store_fill_field_upd(s, &r.upd, &r.fu));
        PSCHK(launch_rows_apply(r, n, st));
    }
    if (bump_step) s->global_step++;     // psUpdate: globalStep.incrementAndGet()  (net/PServer.java:213)
    return PS_OK;
}

extern "C" int ps_shard_flat_grad(ps_model_t *m, float **flat_dev, int64_t *nfloats) {
    if (!m || !flat_dev || !nfloats) return ps_set_err(PS_E_BAD_ARG, "null argument");
    if (!m->sh.flat) return ps_set_err(PS_E_STATE, "ps_shard_plan first");
    *flat_dev = m->sh.flat;
    *nfloats = m->sh.flat_elems;
    return PS_OK;
}

extern "C" int ps_shard_apply_flat(ps_model_t *m, int nworkers) {
    RoctxRange roctx_range("ps_shard_apply_flat");
    if (!m || nworkers < 1) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (!m->sh.flat) return ps_set_err(PS_E_STATE, "ps_shard_plan first");
    PSCHK(store_enter(m->s));
    return shard_apply_flat(m, nworkers, m->s->stream);
}

int shard_apply_flat(ps_model *m, int nworkers, hipStream_t st) {
    ps_store *s = m->s;
    const int nfc = m->cfg.nfc;
    ps_updater_t u;
    DenseUpdArgs d;
    memset(&d, 0, sizeof d);
    d.nlayers = nfc; d.B = m->cur_B; d.apply = 1; d.flat_grad = m->sh.flat; d.flat_div = (float)nworkers;
    PSCHK(store_resolve_updater(s, "fc0.weights", &u));
    d.upd = make_upd_params(u);
    int64_t off = 0;
    for (int l = 0; l < nfc; ++l) {
        FcParams &p = s->fc[l];
        DenseLayer &L = d.L[l];
        L.W = p.W; L.Wt = p.Wt; L.Wp = p.Wp; L.S1 = p.S1; L.S2 = p.S2;
        L.K = p.K; L.N = p.N; L.ldw = p.ldw; L.ldwt = p.Kpad;
        L.elem_begin = off; off += (int64_t)(p.K + 1) * p.N; L.elem_end = off;
    }
    if (m->cfg.kind == PS_MODEL_WIDEDEEP) {          // the wide part of the flat buffer, same launch
        WideUpdArgs &w = d.wide;
        w.rows = s->wide.rows; w.W = s->wide.W; w.state = s->wide.state; w.touched = s->wide.touched;
        w.bias = s->wide.bias; w.bias_state = s->wide.bias_state; w.mode = 2; w.nworkers = nworkers;
        w.G = m->sh.flat + m->dense_elems; w.C = w.G + s->wide.rows;
        if (m->sh.slot_world) {
            if (nworkers != m->sh.slot_world) return ps_set_err(PS_E_BAD_ARG, "the wide slots hold %d workers, not %d", m->sh.slot_world, nworkers);
            w.mode = 7; w.slots = m->sh.flat + m->dense_elems; w.slot_words = (int)m->sh.slot_words; w.world = m->sh.slot_world; w.rank = m->sh.slot_rank;
        }
        PSCHK(store_resolve_updater(s, "wide.weights", &u));
        w.upd = make_upd_params(u);
        d.wide_blocks = wide_update_blocks(w);
    }
    PSCHK(launch_dense_update(d, st));
    return PS_OK;
}
