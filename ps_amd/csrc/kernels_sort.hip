// kernels_sort.hip -- stable LSD radix sort of (key,val) u32 pairs and the
// segment (run-of-equal-keys) builder that the deterministic embedding
// backward stands on (per-key sums in batch order, no
// float atomics).  gfx950: 64-wide waves, ballot-based stable ranking.
//
// Every kernel first issues ALL of its global loads unconditionally (indices
// clamped, not branched): a load inside an `if` of an unrolled loop makes
// hipcc wait for each one separately, and at these sizes (1e5 keys) a kernel
// is nothing but a chain of memory round trips.
#include "ps_common.h"

namespace {

constexpr int RS_TPB = 256;
#ifndef PS_RS_IPT
#define PS_RS_IPT 32        // round 5: 8192-pair tiles (was 16 = 4096).  Half as many, twice as large workgroups: every pass of the multi-hot
                            // step's sort takes LONGER alone (scatter 39 -> 68 us) and the step is 10 us shorter (0.3766 -> 0.3664 ms) -- the kernels
                            // beside it (forward GEMMs, per-key reduce) lose less to 416 resident workgroups than to 806 that are still being placed
#endif
constexpr int RS_IPT = PS_RS_IPT;       // keys per thread (tile = 256 x RS_IPT pairs); -DPS_RS_IPT=8 for A/B builds
constexpr int RS_TILE = RS_TPB * RS_IPT;  // 8192 keys per workgroup
//  the run-head bits of a thread's keys live in one uint32_t (k_seg_fused, k_seg_emit); k_seg_scatter<., RS_IPT> keeps 2 x RS_TILE
// words + its cursors in LDS: 75 KB at 32 -- inside gfx950's 160 KB, beyond any 64 KB part's
static_assert(RS_IPT >= 1 && RS_IPT <= 32, "PS_RS_IPT: at most 32 keys per thread (one head bit each in a uint32_t)");
static_assert((size_t)RS_TILE * 8 + 16384 <= 160 * 1024, "the staged scatter's LDS footprint must fit gfx950's 160 KB");
constexpr int RS_WAVE_SPAN = RS_TILE / 4; // 2048 consecutive keys per wave
constexpr int RS_SB_LOG = 5, RS_SB = 1 << RS_SB_LOG;   // tiles per superblock (second level of the digit counts)

// per-block digit histogram; counts is digit-major: counts[digit * nblk + blk].  DB = digit bits: 8 (256 buckets)
// or 11 (2048: two passes instead of three for the 22-bit keys of a 3 M-pair multi-hot batch; the ballot ranking of
// the scatter costs the same per key bit, the memory passes and launches are two thirds)
template <int DB>
__global__ __launch_bounds__(RS_TPB) void k_radix_hist(const uint32_t *__restrict__ keys, int64_t n,
                                                       int shift, uint32_t *__restrict__ counts,
                                                       int nblk, uint32_t *__restrict__ hi, int nsb, unsigned long long *ts) {
    constexpr int ND = 1 << DB, DPT = ND / RS_TPB;
    __shared__ uint32_t h[ND];
    StampScope stamp(ts);
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < DPT; ++q) h[tid + q * RS_TPB] = 0;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    uint32_t k[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = base + j * RS_TPB + tid;
        k[j] = keys[idx < n ? idx : n - 1];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = base + j * RS_TPB + tid;
        if (idx < n) atomicAdd(&h[(k[j] >> shift) & (uint32_t)(ND - 1)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
        const int d = tid + q * RS_TPB;
        const uint32_t c = h[d];
        // second level (hi != nullptr: the scan-free passes): keys of the digit per superblock of RS_SB tiles.  Integer
        // atomics, ~RS_SB per address; the scatter adds the superblocks before its own and the tiles before it inside
        // its superblock -- no scan launch between the two kernels of a pass.  Both levels are TILE-major there
        // ([tile][digit]: one wave's stores, atomics and the scatter's loads are consecutive words); the scan wants
        // digit-major rows.
        if (hi) {
            counts[(size_t)blockIdx.x * ND + d] = c;
            if (c) atomicAdd(&hi[(size_t)(blockIdx.x >> RS_SB_LOG) * ND + d], c);
        } else {
            counts[(size_t)d * nblk + blockIdx.x] = c;
        }
    }
}

// counts is digit-major [256][nblk].  Workgroup d turns row d into its exclusive prefix (positions of digit d's
// keys of block b among all keys with digit d) and leaves the row total in totals[d].  One workgroup per digit:
// the scan scales with the key count (a single-workgroup scan of 256*nblk counters was THE cost at 3e6 keys).
__global__ __launch_bounds__(RS_TPB) void k_scan_rows(uint32_t *__restrict__ counts, int nblk, uint32_t *__restrict__ totals, unsigned long long *ts) {
    __shared__ uint32_t wsum[4];
    StampScope stamp(ts);
    __shared__ uint32_t carry_s;
    uint32_t *row = counts + (size_t)blockIdx.x * nblk;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += RS_TPB) {
        const int i = base + tid;
        const uint32_t v = i < nblk ? row[i] : 0u;
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
        if (i < nblk) row[i] = wb + inc - v;
        __syncthreads();
        if (tid == RS_TPB - 1) carry_s = wb + inc;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry_s;
}

// Stable scatter through LDS.  Wave w of the block owns keys [blk*4096 + w*1024, +1024) and ranks them in index
// order (equal digits inside a batch of 64 by ballots, across batches and waves by per-(wave,digit) cursors);
// the ranked pairs are first placed at their position INSIDE THE BLOCK's sorted order in LDS, then written out
// by consecutive threads, so one store instruction covers runs of one digit instead of 64 scattered words.
template <bool IOTA, bool STAGED, int DB>
__global__ __launch_bounds__(RS_TPB) void k_radix_scatter(const uint32_t *__restrict__ kin,
                                                          const uint32_t *__restrict__ vin,
                                                          uint32_t *__restrict__ kout,
                                                          uint32_t *__restrict__ vout, int64_t n,
                                                          int shift,
                                                          const uint32_t *__restrict__ offs,
                                                          const uint32_t *__restrict__ totals,
                                                          int nblk, const uint32_t *__restrict__ hi, int nsb, unsigned long long *ts) {
    StampScope stamp(ts);
    constexpr int ND = 1 << DB, DPT = ND / RS_TPB;   // thread t owns digits DPT*t .. DPT*t + DPT-1 (digit order = thread order)
    constexpr uint32_t DMASK = ND - 1;
    __shared__ uint32_t cur[4][ND];
    __shared__ uint32_t gdelta[ND];        // global position - block-local position, per digit
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t sk[STAGED ? RS_TILE : 1], sv[STAGED ? RS_TILE : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 4 * ND; i += RS_TPB) ((uint32_t *)cur)[i] = 0;
    const int64_t bbase = (int64_t)blockIdx.x * RS_TILE;
    const int64_t wbase = bbase + (int64_t)w * RS_WAVE_SPAN;
    uint32_t k[RS_IPT], v[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const int64_t ci = idx < n ? idx : n - 1;
        k[j] = kin[ci];
        v[j] = IOTA ? (uint32_t)idx : vin[ci];
    }
    uint32_t in_digit[DPT], dtot[DPT];
    if (hi) {
        // scan-free: offs holds the raw per-tile counts, hi the per-superblock sums.  Keys of the digit in earlier tiles =
        // the superblocks before mine + the tiles before me inside mine; the digit's total = all superblocks.  All loads
        // independent, from clamped addresses.
        const int sb = blockIdx.x >> RS_SB_LOG, b_in = blockIdx.x & (RS_SB - 1);
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int d = tid * DPT + q;
            const uint32_t *lo = offs + (size_t)sb * RS_SB * ND + d;
            const uint32_t *hh = hi + d;
            uint32_t pre = 0, tot = 0;
            // two batches of RS_SB loads, one after the other: all 2*RS_SB in flight at once cost 64 more live VGPRs
            // (135 in all: three waves per SIMD instead of four, and a second round of workgroups at 779 tiles)
            {
                uint32_t lv[RS_SB];
#pragma unroll
                for (int i = 0; i < RS_SB; ++i) lv[i] = lo[(size_t)(i < b_in ? i : 0) * ND];
#pragma unroll
                for (int i = 0; i < RS_SB; ++i) if (i < b_in) pre += lv[i];
            }
            asm volatile("" : "+v"(pre) :: "memory");
            {
                uint32_t hv[RS_SB];
#pragma unroll
                for (int i = 0; i < RS_SB; ++i) hv[i] = hh[(size_t)(i < nsb ? i : 0) * ND];
#pragma unroll
                for (int i = 0; i < RS_SB; ++i) {
                    if (i < nsb) tot += hv[i];
                    if (i < sb) pre += hv[i];
                }
            }
            for (int s0 = RS_SB; s0 < nsb; s0 += 8) {            // more than RS_SB superblocks (> 4 M keys)
                uint32_t h8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) h8[i] = hh[(size_t)(s0 + i < nsb ? s0 + i : 0) * ND];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (s0 + i < nsb) tot += h8[i];
                    if (s0 + i < sb) pre += h8[i];
                }
            }
            in_digit[q] = pre; dtot[q] = tot;
        }
    } else {
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            in_digit[q] = offs[(size_t)(tid * DPT + q) * nblk + blockIdx.x];   // keys of that digit in earlier blocks
            dtot[q] = totals[tid * DPT + q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        if (idx < n) atomicAdd(&cur[w][(k[j] >> shift) & DMASK], 1u);
    }
    // exclusive scan of the ND digit totals (global digit bases) and of this block's digit counts (local bases)
    uint32_t tsum = 0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) tsum += dtot[q];
    uint32_t ginc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(ginc, off);
        if (lane >= off) ginc += t;
    }
    if (lane == 63) wtot[w] = ginc;
    __syncthreads();
    uint32_t gbase = ginc - tsum;
    for (int ww = 0; ww < w; ++ww) gbase += wtot[ww];
    uint32_t c0[DPT], c1[DPT], c2[DPT], c3[DPT];
    uint32_t bsum = 0;
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
        const int d = tid * DPT + q;
        c0[q] = cur[0][d]; c1[q] = cur[1][d]; c2[q] = cur[2][d]; c3[q] = cur[3][d];
        bsum += c0[q] + c1[q] + c2[q] + c3[q];
    }
    uint32_t linc = bsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(linc, off);
        if (lane >= off) linc += t;
    }
    __syncthreads();                       // wtot is re-used
    if (lane == 63) wtot[w] = linc;
    __syncthreads();
    uint32_t lbase = linc - bsum;
    for (int ww = 0; ww < w; ++ww) lbase += wtot[ww];
#pragma unroll
    for (int q = 0; q < DPT; ++q) {
        const int d = tid * DPT + q;
        cur[0][d] = lbase; cur[1][d] = lbase + c0[q]; cur[2][d] = lbase + c0[q] + c1[q]; cur[3][d] = lbase + c0[q] + c1[q] + c2[q];
        gdelta[d] = gbase + in_digit[q] - lbase;
        lbase += c0[q] + c1[q] + c2[q] + c3[q];
        gbase += dtot[q];
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (k[j] >> shift) & DMASK;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        uint32_t rank = 0, cnt = 0;
        if (valid) {
            rank = (uint32_t)__popcll(same & below);
            cnt = (uint32_t)__popcll(same);
            const uint32_t pos = cur[w][d] + rank;
            if (STAGED) { sk[pos] = k[j]; sv[pos] = v[j]; }
            else { const uint32_t gp = pos + gdelta[d]; kout[gp] = k[j]; vout[gp] = v[j]; }     // small inputs: latency, not bandwidth
        }
        if (valid && rank + 1 == cnt) cur[w][d] += cnt;  // last lane of the group advances the cursor
    }
    if (!STAGED) return;
    __syncthreads();
    const int64_t rem = n - bbase;
    const int cnt_blk = rem < RS_TILE ? (int)rem : RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int lp = j * RS_TPB + tid;
        if (lp < cnt_blk) {
            const uint32_t key = sk[lp];
            const uint32_t gp = lp + gdelta[(key >> shift) & DMASK];
            kout[gp] = key;
            vout[gp] = sv[lp];
        }
    }
}

// ---- the segmented sort of a multi-hot batch (round 4) ---------------------------------------------------------------
// A multi-hot batch's keys are (field, id).  An entry's FIELD follows from its bag, so the partition by field costs no
// radix pass: a column scan of the bag lengths (k_bag_scan) gives every bag its place among its field's entries, the key
// kernel (kernels_emb.hip k_emb_keys_seg) writes (id, bag) there, and two 9-bit passes over the ids, each field sorted on
// its own, finish the job -- two passes over 3.2 M pairs instead of three (configs[4]'s shape: 22-bit keys).
// Layout between the launches: field f's entries start at a multiple of the tile, pb[f] = sum over f' < f of
// round_up(ftotal[f'], RS_TILE), so that no tile holds two fields; the slots behind a field's last entry are never read
// (every kernel knows the field's count).  The last pass writes the compact array the backward expects: all entries in
// (field, id, batch order), keys = rows of the concatenated table again.
// Same result as radix_sort_pairs on the 22-bit keys, entry for entry (stable: equal keys keep their batch order).
constexpr int SG_DB = 9, SG_ND = 1 << SG_DB, SG_DPT = SG_ND / RS_TPB;
struct SegSortArgs {
    int F;
    const uint32_t *ftotal;         // [F] entries of every field (k_bag_scan)
    const int64_t *row_base;        // [F + 1] first row of every field in the concatenated table
    uint32_t *ftot;                 // [F][SG_ND] this pass's digit totals per field (zero at the start of the pass)
    uint32_t *tcounts;              // [tiles][SG_ND] this pass's digit counts per tile
};
// padded bases of all fields into LDS (pb[F] = end); returns nothing: call from every thread, ends with a barrier
__device__ __forceinline__ void seg_bases(const SegSortArgs &s, uint32_t *pb, uint32_t *cb) {
    if (threadIdx.x == 0) {
        uint32_t p = 0, c = 0;
        for (int f = 0; f < s.F; ++f) {
            pb[f] = p; cb[f] = c;
            const uint32_t n = s.ftotal[f];
            p += (n + RS_TILE - 1) / RS_TILE * RS_TILE; c += n;
        }
        pb[s.F] = p; cb[s.F] = c;
    }
    __syncthreads();
}
__device__ __forceinline__ int seg_field_of(const uint32_t *pb, int F, uint32_t pos) {
    int f = 0;
    while (f + 1 < F && pos >= pb[f + 1]) ++f;
    return f;
}

// pre[b * F + f] = entries of field f in samples < b; ftotal[f]; also zeroes zero_words words at zero (the sort's per-field totals).
// One workgroup per field; a thread takes BS_SPT consecutive samples per sweep, all of its loads issued up front
// (the first version -- 256 threads, 16 samples each behind a runtime trip count -- took 47 us beside the gather: a chain of
// strided loads).
constexpr int BS_TPB = 256, BS_SPT = 16;     // (1024-thread workgroups waited ~30 us for a CU with 16 free wave slots beside the gather)
__global__ __launch_bounds__(BS_TPB) void k_bag_scan(const int64_t *__restrict__ offsets, int B, int F, uint32_t *__restrict__ pre,
                                                     uint32_t *__restrict__ ftotal, uint32_t *__restrict__ zero, int zero_words, unsigned long long *ts) {
    StampScope stamp(ts);
    __shared__ uint32_t wsum[BS_TPB / 64];
    __shared__ uint32_t carry_s;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid + f * BS_TPB; i < zero_words; i += BS_TPB * gridDim.x) zero[i] = 0u;
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    for (int base = 0; base < B; base += BS_TPB * BS_SPT) {
        const int b0 = base + tid * BS_SPT;
        int64_t o[BS_SPT + 1];
        // bag (b, f) and the bag behind it: offsets[b F + f], offsets[b F + f + 1] -- both loaded (the next sample's bag of this field
        // is F bags further on)
        uint32_t len[BS_SPT];
#pragma unroll
        for (int j = 0; j < BS_SPT; ++j) {
            const int b = b0 + j < B ? b0 + j : B - 1;
            const int64_t bag = (int64_t)b * F + f;
            o[j] = offsets[bag + 1] - offsets[bag];
        }
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < BS_SPT; ++j) { len[j] = b0 + j < B ? (uint32_t)o[j] : 0u; sum += len[j]; }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t run = carry_s + inc - sum;
        for (int ww = 0; ww < w; ++ww) run += wsum[ww];
#pragma unroll
        for (int j = 0; j < BS_SPT; ++j) {
            if (b0 + j < B) pre[(int64_t)(b0 + j) * F + f] = run;
            run += len[j];
        }
        __syncthreads();
        if (tid == BS_TPB - 1) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) ftotal[f] = carry_s;
}

template <int IPT>
__global__ __launch_bounds__(RS_TPB) void k_seg_hist(const uint32_t *__restrict__ keys, int shift, SegSortArgs s, unsigned long long *ts) {
    constexpr int TILE = RS_TPB * IPT;          // this pass's tile (the padded layout's tile, RS_TILE, is a multiple of it)
    __shared__ uint32_t h[SG_ND];
    __shared__ uint32_t pb[65], cb[65];
    StampScope stamp(ts);
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < SG_DPT; ++q) h[tid + q * RS_TPB] = 0;
    seg_bases(s, pb, cb);
    const uint32_t bbase = blockIdx.x * (uint32_t)TILE;
    if (bbase >= pb[s.F]) return;                          // (the grid is an upper bound)
    const int f = seg_field_of(pb, s.F, bbase);
    const uint32_t in_f = bbase - pb[f], nf = s.ftotal[f];
    const uint32_t nvalid = nf > in_f ? (nf - in_f < (uint32_t)TILE ? nf - in_f : (uint32_t)TILE) : 0u;
    uint32_t k[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint32_t i = j * RS_TPB + tid;
        k[j] = keys[bbase + (i < nvalid ? i : 0)];
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j)
        if ((uint32_t)(j * RS_TPB + tid) < nvalid) atomicAdd(&h[(k[j] >> shift) & (uint32_t)(SG_ND - 1)], 1u);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SG_DPT; ++q) {
        const int d = tid + q * RS_TPB;
        const uint32_t c = h[d];
        s.tcounts[(size_t)blockIdx.x * SG_ND + d] = c;
        if (c) atomicAdd(&s.ftot[(size_t)f * SG_ND + d], c);
    }
}

// (ranking and staging as k_radix_scatter<false, true, .>; LAST: compact output, keys = rows of the concatenated table)
template <bool LAST, int IPT>
__global__ __launch_bounds__(RS_TPB) void k_seg_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                        uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, int shift,
                                                        SegSortArgs s, unsigned long long *ts) {
    StampScope stamp(ts);
    constexpr uint32_t DMASK = SG_ND - 1;
    constexpr int TILE = RS_TPB * IPT, WAVE_SPAN = TILE / 4;
    __shared__ uint32_t cur[4][SG_ND];
    __shared__ uint32_t gdelta[SG_ND];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t sk[TILE], sv[TILE];
    __shared__ uint32_t pb[65], cb[65];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 4 * SG_ND; i += RS_TPB) ((uint32_t *)cur)[i] = 0;
    seg_bases(s, pb, cb);
    const uint32_t bbase = blockIdx.x * (uint32_t)TILE;
    if (bbase >= pb[s.F]) return;
    const int f = seg_field_of(pb, s.F, bbase);
    const uint32_t in_f = bbase - pb[f], nf = s.ftotal[f];
    const uint32_t nvalid = nf > in_f ? (nf - in_f < (uint32_t)TILE ? nf - in_f : (uint32_t)TILE) : 0u;
    const int t0 = (int)(pb[f] / TILE), ti = (int)blockIdx.x - t0;       // this field's first tile, my index among its tiles
    const uint32_t wbase = (uint32_t)w * WAVE_SPAN;                       // (inside the tile)
    uint32_t k[IPT], v[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint32_t i = wbase + j * 64 + lane;
        const uint32_t ci = bbase + (i < nvalid ? i : 0);
        k[j] = kin[ci];
        v[j] = vin[ci];
    }
    // keys of my digits in the field's earlier tiles (all loads independent, clamped), and the field's digit totals
    uint32_t in_digit[SG_DPT], dtot[SG_DPT];
#pragma unroll
    for (int q = 0; q < SG_DPT; ++q) {
        const int d = tid * SG_DPT + q;
        const uint32_t *lo = s.tcounts + (size_t)t0 * SG_ND + d;
        uint32_t pre = 0;
        for (int i0 = 0; i0 < ti; i0 += 16) {
            uint32_t lv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) lv[i] = lo[(size_t)(i0 + i < ti ? i0 + i : 0) * SG_ND];
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i0 + i < ti) pre += lv[i];
        }
        in_digit[q] = pre;
        dtot[q] = s.ftot[(size_t)f * SG_ND + d];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j)
        if (wbase + j * 64 + lane < nvalid) atomicAdd(&cur[w][(k[j] >> shift) & DMASK], 1u);
    // exclusive scan of the field's digit totals (digit bases inside the field) and of this tile's digit counts (local bases)
    uint32_t tsum = 0;
#pragma unroll
    for (int q = 0; q < SG_DPT; ++q) tsum += dtot[q];
    uint32_t ginc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(ginc, off); if (lane >= off) ginc += t; }
    if (lane == 63) wtot[w] = ginc;
    __syncthreads();
    uint32_t gbase = ginc - tsum + (LAST ? cb[f] : pb[f]);
    for (int ww = 0; ww < w; ++ww) gbase += wtot[ww];
    uint32_t c0[SG_DPT], c1[SG_DPT], c2[SG_DPT], c3[SG_DPT];
    uint32_t bsum = 0;
#pragma unroll
    for (int q = 0; q < SG_DPT; ++q) {
        const int d = tid * SG_DPT + q;
        c0[q] = cur[0][d]; c1[q] = cur[1][d]; c2[q] = cur[2][d]; c3[q] = cur[3][d];
        bsum += c0[q] + c1[q] + c2[q] + c3[q];
    }
    uint32_t linc = bsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(linc, off); if (lane >= off) linc += t; }
    __syncthreads();                       // wtot is re-used
    if (lane == 63) wtot[w] = linc;
    __syncthreads();
    uint32_t lbase = linc - bsum;
    for (int ww = 0; ww < w; ++ww) lbase += wtot[ww];
#pragma unroll
    for (int q = 0; q < SG_DPT; ++q) {
        const int d = tid * SG_DPT + q;
        cur[0][d] = lbase; cur[1][d] = lbase + c0[q]; cur[2][d] = lbase + c0[q] + c1[q]; cur[3][d] = lbase + c0[q] + c1[q] + c2[q];
        gdelta[d] = gbase + in_digit[q] - lbase;
        lbase += c0[q] + c1[q] + c2[q] + c3[q];
        gbase += dtot[q];
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const bool valid = wbase + j * 64 + lane < nvalid;
        const uint32_t d = (k[j] >> shift) & DMASK;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < SG_DB; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        uint32_t rank = 0, cnt = 0;
        if (valid) {
            rank = (uint32_t)__popcll(same & below);
            cnt = (uint32_t)__popcll(same);
            const uint32_t pos = cur[w][d] + rank;
            sk[pos] = k[j]; sv[pos] = v[j];
        }
        if (valid && rank + 1 == cnt) cur[w][d] += cnt;  // last lane of the group advances the cursor
    }
    __syncthreads();
    const uint32_t rb = LAST ? (uint32_t)s.row_base[f] : 0u;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint32_t lp = j * RS_TPB + tid;
        if (lp < nvalid) {
            const uint32_t key = sk[lp];
            const uint32_t gp = lp + gdelta[(key >> shift) & DMASK];
            kout[gp] = key + rb;
            vout[gp] = sv[lp];
        }
    }
}

// ---- segments -------------------------------------------------------------
// head(idx) = idx == 0 || keys[idx] != keys[idx-1], from two unconditional loads
__device__ __forceinline__ bool head_of(uint32_t cur, uint32_t prev, int64_t idx, int64_t n) {
    return idx < n && (idx == 0 || cur != prev);
}

__global__ __launch_bounds__(RS_TPB) void k_seg_count(const uint32_t *__restrict__ keys, int64_t n,
                                                      uint32_t *__restrict__ blk_heads) {
    __shared__ uint32_t red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    uint32_t kc[RS_IPT], kp[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = base + j * RS_TPB + tid;
        const int64_t ci = idx < n ? idx : n - 1;
        kc[j] = keys[ci];
        kp[j] = keys[ci > 0 ? ci - 1 : 0];
    }
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) c += head_of(kc[j], kp[j], base + j * RS_TPB + tid, n) ? 1u : 0u;
    for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) red[w] = c;
    __syncthreads();
    if (tid == 0) blk_heads[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// the runs longer than long_min entries, appended in any order (k_emb_reduce_update's long-key role walks the list):
// nseg[1] counts them (zeroed by k_seg_emit)
__global__ __launch_bounds__(256) void k_long_runs(const uint32_t *__restrict__ seg_start, uint32_t *__restrict__ nseg,
                                                   uint32_t *__restrict__ long_list, uint32_t long_min) {
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= nseg[0]) return;
    const uint32_t s0 = seg_start[u], e0 = seg_start[u + 1];
    if (e0 - s0 > long_min) {
        uint32_t *ll = long_list + 3 * (size_t)atomicAdd(&nseg[1], 1u);
        ll[0] = u; ll[1] = s0; ll[2] = e0;
    }
}

__global__ __launch_bounds__(RS_TPB) void k_seg_emit(const uint32_t *__restrict__ keys, int64_t n,
                                                     const uint32_t *__restrict__ blk_heads,
                                                     uint32_t *__restrict__ seg_start,
                                                     uint32_t *__restrict__ seg_id,
                                                     uint32_t *__restrict__ nseg_dev, unsigned long long *ts) {
    __shared__ uint32_t red[4];
    __shared__ uint32_t wave_heads[4];
    StampScope stamp(ts);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * RS_WAVE_SPAN;
    uint32_t kc[RS_IPT], kp[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const int64_t ci = idx < n ? idx : n - 1;
        kc[j] = keys[ci];
        kp[j] = keys[ci > 0 ? ci - 1 : 0];
    }
    // heads in all earlier blocks
    uint32_t acc = 0;
    for (int i = tid; i < (int)blockIdx.x; i += RS_TPB) acc += blk_heads[i];
    for (int off = 32; off; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) red[w] = acc;
    uint32_t hbits = 0, wcount = 0;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const bool h = head_of(kc[j], kp[j], wbase + j * 64 + lane, n);
        hbits |= (h ? 1u : 0u) << j;
        wcount += (uint32_t)__popcll(__ballot(h));
    }
    if (lane == 0) wave_heads[w] = wcount;
    __syncthreads();
    uint32_t run = red[0] + red[1] + red[2] + red[3];
    for (int ww = 0; ww < w; ++ww) run += wave_heads[ww];
    const uint64_t le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const bool h = (hbits >> j) & 1u;
        const uint64_t hm = __ballot(h);
        if (idx < n) {
            const uint32_t incl = run + (uint32_t)__popcll(hm & le);  // heads up to and incl. idx
            const uint32_t sid = incl - 1;
            seg_id[idx] = sid;
            if (h) seg_start[sid] = (uint32_t)idx;
            if (idx == n - 1) {
                seg_start[sid + 1] = (uint32_t)n;
                nseg_dev[0] = sid + 1;
                nseg_dev[1] = 0;                        // long-run counter (k_long_runs, when asked for)
            }
        }
        run += (uint32_t)__popcll(hm);
    }
}


// Round 5: count + emit in ONE launch (the two segment launches and their boundaries were a 35 us hole
// behind the multi-hot step's sort).  Every workgroup counts the run heads of its tile, publishes the count tagged with the
// launch's sequence number and adds up the counts of the workgroups in front of it -- they were dispatched earlier, so
// spinning on their words cannot deadlock (the look-back of k_field_sort_segments and the sharded plan) -- then writes
// seg_id / seg_start exactly as k_seg_emit does.  One pass over the sorted keys instead of two.
__global__ __launch_bounds__(RS_TPB) void k_seg_fused(const uint32_t *__restrict__ keys, int64_t n, unsigned long long *__restrict__ pub, uint32_t seq,
                                                      uint32_t *__restrict__ seg_start, uint32_t *__restrict__ seg_id,
                                                      uint32_t *__restrict__ nseg_dev, unsigned long long *ts) {
    __shared__ uint32_t red[4];
    __shared__ uint32_t wave_heads[4];
    StampScope stamp(ts);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = (int)blockIdx.x;
    const int64_t wbase = (int64_t)b * RS_TILE + (int64_t)w * RS_WAVE_SPAN;
    uint32_t kc[RS_IPT], kp[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const int64_t ci = idx < n ? idx : n - 1;
        kc[j] = keys[ci];
        kp[j] = keys[ci > 0 ? ci - 1 : 0];
    }
    uint32_t hbits = 0, wcount = 0;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const bool h = head_of(kc[j], kp[j], wbase + j * 64 + lane, n);
        hbits |= (h ? 1u : 0u) << j;
        wcount += (uint32_t)__popcll(__ballot(h));
    }
    if (lane == 0) wave_heads[w] = wcount;
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(&pub[b], ((unsigned long long)seq << 32) | (unsigned long long)(wave_heads[0] + wave_heads[1] + wave_heads[2] + wave_heads[3]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // heads in all earlier workgroups
    uint32_t acc = 0;
    for (int i = tid; i < b; i += RS_TPB) {
        unsigned long long x;
        do { x = __hip_atomic_load(&pub[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(x >> 32) != seq);
        acc += (uint32_t)x;
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) red[w] = acc;
    __syncthreads();
    uint32_t run = red[0] + red[1] + red[2] + red[3];
    for (int ww = 0; ww < w; ++ww) run += wave_heads[ww];
    const uint64_t le = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const bool h = (hbits >> j) & 1u;
        const uint64_t hm = __ballot(h);
        if (idx < n) {
            const uint32_t incl = run + (uint32_t)__popcll(hm & le);  // heads up to and incl. idx
            const uint32_t sid = incl - 1;
            seg_id[idx] = sid;
            if (h) seg_start[sid] = (uint32_t)idx;
            if (idx == n - 1) {
                seg_start[sid + 1] = (uint32_t)n;
                nseg_dev[0] = sid + 1;
                nseg_dev[1] = 0;                        // long-run counter (k_long_runs, when asked for)
            }
        }
        run += (uint32_t)__popcll(hm);
    }
}

// ---------------------------------------------------------------------------
// Single-hot batches: the whole "sort the (row key, entry) pairs, cut them into per-key runs" chain in ONE launch.
// Field f's B keys occupy their own interval of the key space (row = row_base[f] + id, kernels_emb.hip), so the
// global stable sort is F independent sorts of B pairs -- each small enough (B <= 8192) to live in one workgroup's
// LDS.  Workgroup f:
//   1. loads its column of the [B][F] key matrix as 64-bit composites (key << 32 | sample): ascending composites
//      = ascending keys, ties in batch order, i.e. exactly the stable radix sort's result;
//   2. bitonic-sorts them in LDS (1024 threads, log2(NP)(log2(NP)+1)/2 stages);
//   3. marks the run heads, scans them, counts the runs longer than `long_min` entries;
//   4. publishes (epoch, #long runs, #runs) in pub[f] and adds up the words of the fields before it (they were
//      dispatched earlier and wait only on earlier ones still: no circular wait, however many fields are resident);
//   5. writes sorted_keys / sorted_ents / seg_id / seg_start (+ the list of long runs for k_emb_reduce_update's
//      long-key role); the last field also writes nseg, the sentinel seg_start[nseg] and the long-run count.
// 11 launches of the radix chain (3 x hist/scan/scatter + 2 segment kernels, ~63 us of a side stream and most of the
// chip's CUs touched by each) become one that occupies F CUs.
// ---------------------------------------------------------------------------
constexpr int FS_TPB = 1024;
constexpr int FS_MAX = 8192;

struct FieldSortArgs {
    const uint32_t *keys;               // [B][F] row keys in entry order (entry = b * F + f)
    const int64_t *keys_base;           // [F] first row key of every field (subtracted before the sort) or nullptr
    int B, F, NP, long_min, npass, digit_bits;
    uint32_t *sorted_keys, *sorted_ents, *seg_start, *seg_id, *nseg;   // nseg[0] = runs, nseg[1] = long runs
    uint32_t *long_list;                // run ids with more than long_min entries (any order)
    unsigned long long *pub;            // [F] look-back words
    uint32_t epoch;
    unsigned long long *ts;
    unsigned int *start_flag; unsigned int start_val;   // "this launch has started" = everything in front of it on its stream is done (or NULL)
};

#ifdef PS_FS_TIMING
__device__ unsigned long long g_fs_t[64 * 8];
#define FS_T(k) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_fs_t[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define FS_T(k) do { } while (0)
#endif

// The sort itself: stable LSD radix passes entirely in LDS, ballot-ranked like k_radix_scatter (a bitonic network
// on the same workgroup was VALU-bound on its one CU: 78 stages x ~20 instructions per element, 52 us at B = 4096;
// a radix pass costs ~1 wave instruction per element).  Wave w owns the contiguous span [w * SPAN, (w + 1) * SPAN)
// of positions, 64 consecutive ones per chunk, so (wave, chunk, lane) order IS position order and the ranking
// below is stable.
template <int EPT>
__global__ __launch_bounds__(FS_TPB) void k_field_sort_segments(FieldSortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_lds[];
    constexpr int NP = FS_TPB * EPT, NW = FS_TPB / 64, SPAN = NP / NW, CH = SPAN / 64;      // CH == EPT
    uint32_t *kA = reinterpret_cast<uint32_t *>(fs_lds), *vA = kA + NP, *kB = vA + NP, *vB = kB + NP;
    __shared__ uint32_t dcnt[NW][256];
    __shared__ uint32_t wtot[16], wlong[16], base_s[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int f = blockIdx.x, B = a.B;
    StampScope stamp(a.ts);
    if (a.start_flag && f == 0 && tid == 0) __hip_atomic_store(a.start_flag, a.start_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 2) base_s[tid] = 0;
    FS_T(0);
    const uint32_t fbase = a.keys_base ? (uint32_t)a.keys_base[f] : 0u;
    {
        uint32_t kk[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int p = w * SPAN + c * 64 + lane;
            kk[c] = a.keys[(size_t)(p < B ? p : B - 1) * a.F + f];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int p = w * SPAN + c * 64 + lane;
            kA[p] = p < B ? kk[c] - fbase : 0xffffffffu;      // pads: the largest digit in every pass, behind everything
            vA[p] = (uint32_t)p;
        }
    }
    FS_T(1);
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t *kin = kA, *vin = vA, *kout = kB, *vout = vB;
    for (int pass = 0, shift = 0; pass < a.npass; ++pass, shift += a.digit_bits) {
        const uint32_t dmask = (1u << a.digit_bits) - 1u;
        for (int q = tid; q < NW * 256; q += FS_TPB) (&dcnt[0][0])[q] = 0;
        __syncthreads();
        uint32_t key[CH], val[CH], dig[CH], rank[CH], gcnt[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int p = w * SPAN + c * 64 + lane;
            key[c] = kin[p]; val[c] = vin[p];
            dig[c] = (key[c] >> shift) & dmask;
            uint64_t same = ~0ull;
            for (int b = 0; b < a.digit_bits; ++b) {
                const bool bit = (dig[c] >> b) & 1u;
                const uint64_t m = __ballot(bit);
                same &= bit ? m : ~m;
            }
            rank[c] = (uint32_t)__popcll(same & below);
            gcnt[c] = (uint32_t)__popcll(same);
            if (rank[c] == 0) dcnt[w][dig[c]] += gcnt[c];      // one lane per digit group; chunks of a wave in order
        }
        __syncthreads();
        {
            // exclusive scan of the NW x 256 counters in (digit, wave) order: thread t owns digit t / 4, waves 4 (t % 4) ..
            const int d = tid >> 2, w0 = (tid & 3) * 4;
            const uint32_t c0 = dcnt[w0][d], c1 = dcnt[w0 + 1][d], c2 = dcnt[w0 + 2][d], c3 = dcnt[w0 + 3][d];
            const uint32_t tot = c0 + c1 + c2 + c3;
            uint32_t inc = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(inc, off);
                if (lane >= off) inc += t;
            }
            if (lane == 63) wtot[w] = inc;
            __syncthreads();
            uint32_t ex = inc - tot;
            for (int q = 0; q < w; ++q) ex += wtot[q];
            dcnt[w0][d] = ex; dcnt[w0 + 1][d] = ex + c0; dcnt[w0 + 2][d] = ex + c0 + c1; dcnt[w0 + 3][d] = ex + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t pos = dcnt[w][dig[c]] + rank[c];
            kout[pos] = key[c]; vout[pos] = val[c];
            if (rank[c] + 1 == gcnt[c]) dcnt[w][dig[c]] += gcnt[c];      // last lane of the group advances the cursor
        }
        __syncthreads();
        uint32_t *t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    const uint32_t *ks = kin, *vs = vin;                     // sorted (local id, sample); kout / vout are free now
    uint32_t *starts = kout;                                 // [NP + 1] <= 2 * NP words (kout and vout are adjacent)
    const int r0 = tid * EPT;
    FS_T(2);
    // run heads: each thread owns EPT consecutive sorted positions
    uint32_t heads = 0;                                      // bit e: position r0 + e starts a run
    for (int e = 0; e < EPT; ++e) {
        const int r = r0 + e;
        if (r < B && (r == 0 || ks[r] != ks[r - 1])) heads |= 1u << e;
    }
    uint32_t cnt = __popc(heads), inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t before = inc - cnt, nrun = 0;
    for (int q = 0; q < FS_TPB / 64; ++q) { if (q < w) before += wtot[q]; nrun += wtot[q]; }
    // starts[idx] = first sorted position of run idx; starts[nrun] = B
    {
        uint32_t idx = before;
        for (int e = 0; e < EPT; ++e) if (heads >> e & 1u) starts[idx++] = (uint32_t)(r0 + e);
        if (tid == 0) starts[nrun] = (uint32_t)B;
    }
    __syncthreads();
    // long runs of this field: count, then slots by the same scan
    uint32_t lcnt = 0;
    {
        uint32_t idx = before;
        for (int e = 0; e < EPT; ++e) if (heads >> e & 1u) { if (starts[idx + 1] - starts[idx] > (uint32_t)a.long_min) ++lcnt; ++idx; }
    }
    uint32_t linc = lcnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(linc, off);
        if (lane >= off) linc += t;
    }
    if (lane == 63) wlong[w] = linc;
    __syncthreads();
    uint32_t lbefore = linc - lcnt, nlong = 0;
    for (int q = 0; q < FS_TPB / 64; ++q) { if (q < w) lbefore += wlong[q]; nlong += wlong[q]; }
    FS_T(3);
    // look-back over the fields before this one
    if (tid == 0)
        __hip_atomic_store(&a.pub[f], ((unsigned long long)a.epoch << 32) | ((unsigned long long)nlong << 16) | nrun,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t sr = 0, sl = 0;
    for (int g = tid; g < f; g += FS_TPB) {
        unsigned long long v;
        do { v = __hip_atomic_load(&a.pub[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(v >> 32) != a.epoch);
        sr += (uint32_t)v & 0xffffu; sl += (uint32_t)(v >> 16) & 0xffffu;
    }
    if (sr | sl) { atomicAdd(&base_s[0], sr); atomicAdd(&base_s[1], sl); }
    __syncthreads();
    FS_T(4);
    const uint32_t rbase = base_s[0], lbase = base_s[1];
    if (tid == 0) {     // this field's runs and long runs: [rbase, rbase + nrun) of the runs, [lbase, lbase + nlong) of the long list (k_emb_reduce_update deals by field pair)
        uint32_t *ft = reinterpret_cast<uint32_t *>(a.pub + PS_FS_TAB_OFF(a.F)) + 4 * (size_t)f;
        ft[0] = rbase; ft[1] = nrun; ft[2] = lbase; ft[3] = nlong;
    }
    const uint32_t pos0 = (uint32_t)f * (uint32_t)B;
    {
        uint32_t idx = before, li = lbefore;                 // idx = runs started before position r0
        for (int e = 0; e < EPT; ++e) {
            const int r = r0 + e;
            if (r >= B) break;
            if (heads >> e & 1u) {
                a.seg_start[rbase + idx] = pos0 + (uint32_t)r;
                if (starts[idx + 1] - starts[idx] > (uint32_t)a.long_min) {          // (run id, first entry, end) in one line
                    uint32_t *ll = a.long_list + 3 * (size_t)(lbase + li++);
                    ll[0] = rbase + idx; ll[1] = pos0 + starts[idx]; ll[2] = pos0 + starts[idx + 1];
                }
                ++idx;
            }
            a.sorted_keys[pos0 + r] = ks[r] + fbase;
            a.sorted_ents[pos0 + r] = vs[r] * (uint32_t)a.F + (uint32_t)f;
            a.seg_id[pos0 + r] = rbase + idx - 1;
        }
    }
    FS_T(5);
    if (f == a.F - 1 && tid == 0) {
        a.nseg[0] = rbase + nrun;
        a.nseg[1] = lbase + nlong;
        a.seg_start[rbase + nrun] = (uint32_t)a.F * (uint32_t)B;
    }
}

}  // namespace

int g_radix11 = 0;          // ps_tune_set("radix11", 1): 11-bit digits for large sorts (2 passes instead of 3 at 22 bits): measured SLOWER
int sort_ws_alloc(SortWorkspace &ws, int64_t cap) {
    sort_ws_free(ws);
    ws.cap = cap;
    ws.nblk = cdiv(cap > 0 ? cap : 1, RS_TILE);
    HIPCHK(hipMalloc(&ws.keys_alt, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc(&ws.vals_alt, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc(&ws.counts, sizeof(uint32_t) * 2048 * (size_t)ws.nblk));       // up to 11-bit digits
    HIPCHK(hipMalloc(&ws.blk_heads, sizeof(uint32_t) * (size_t)ws.nblk));
    HIPCHK(hipMalloc(&ws.totals, sizeof(uint32_t) * 2048));
    HIPCHK(hipMalloc(&ws.seg_pub, sizeof(unsigned long long) * (size_t)(ws.nblk + 1)));
    ws.seg_seq = 0;         // (the words are zeroed by the first build_segments, on ITS stream: nothing here may touch the null stream --
                            //  a hipMemsetAsync / hipStreamSynchronize on stream 0 brought the default stream's hardware queue into play,
                            //  and every store + model created after that ran its multi-stream steps 1.6-2.2x slower: 0.60 ms for the
                            //  multi-hot step, 0.33 for the sharded one, bench.py's legs of round 5's first evidence run)
    HIPCHK(hipMalloc(&ws.hi, sizeof(uint32_t) * 4 * 2048 * (size_t)cdiv(ws.nblk, RS_SB)));     // <= 4 passes x 2048 digits x superblocks
    return PS_OK;
}

void sort_ws_free(SortWorkspace &ws) {
    if (ws.keys_alt) (void)hipFree(ws.keys_alt);
    if (ws.vals_alt) (void)hipFree(ws.vals_alt);
    if (ws.counts) (void)hipFree(ws.counts);
    if (ws.blk_heads) (void)hipFree(ws.blk_heads);
    if (ws.totals) (void)hipFree(ws.totals);
    if (ws.seg_pub) (void)hipFree(ws.seg_pub);
    if (ws.hi) (void)hipFree(ws.hi);
    ws = SortWorkspace();
}

int radix_sort_pairs(SortWorkspace &ws, uint32_t *keys, uint32_t *vals, int64_t n, int key_bits,
                     bool iota_vals, uint32_t **keys_res, uint32_t **vals_res, hipStream_t st) {
    if (n > ws.cap) return ps_set_err(PS_E_BAD_ARG, "radix_sort_pairs: n=%lld > cap=%lld", (long long)n, (long long)ws.cap);
    *keys_res = keys; *vals_res = vals;
    if (n <= 0) return PS_OK;
    // 11-bit digits would save a pass over a large array (3.2 M pairs x 22 bits: 2 passes instead of 3) -- measured at
    // configs[4]'s shape: 0.436 ms/step against 0.397 with three 8-bit passes (72 KB of LDS per workgroup, 2048-bucket
    // LDS atomics): off unless ps_tune_set("radix11", 1)
    const bool wide = g_radix11 && n >= (1 << 20) && (key_bits + 10) / 11 < (key_bits + 7) / 8;
    const int db = wide ? 11 : 8;
    int passes = (key_bits + db - 1) / db;
    if (passes < 1) passes = 1;
    const int nblk = cdiv(n, RS_TILE);
    // A pass is TWO launches: the per-tile digit counts (+ per-superblock sums by integer atomics), then the scatter, which
    // adds up "keys of my digit in earlier tiles" itself from the two levels.  (The scan launch in between cost the
    // chain 10 us + a boundary per pass at configs[4]'s shape: tools/gpu_timeline.py, MULTI_HOT=1.)
    const bool scan_free = g_radix_scan_free && passes <= 4;
    const int nsb = cdiv(nblk, RS_SB), nd = 1 << db;
    if (scan_free) HIPCHK(hipMemsetAsync(ws.hi, 0, sizeof(uint32_t) * (size_t)passes * nd * nsb, st));
    uint32_t *kin = keys, *vin = vals, *kout = ws.keys_alt, *vout = ws.vals_alt;
    for (int p = 0; p < passes; ++p) {
        const int shift = db * p;
        uint32_t *hi = scan_free ? ws.hi + (size_t)p * nd * nsb : nullptr;
        // LDS staging pays once the scatter is bandwidth-bound (measured: 3.2 M pairs 3x faster, 1e5 pairs 25 % slower)
        const bool staged = n >= (1 << 19);
        const bool iota = p == 0 && iota_vals;
#define RS_SCATTER(I, S, DBITS) hipLaunchKernelGGL((k_radix_scatter<I, S, DBITS>), dim3(nblk), dim3(RS_TPB), 0, st, kin, (const uint32_t *)vin, kout, vout, n, shift, ws.counts, ws.totals, nblk, hi, nsb, stamp_next("radix_scatter"))
        if (wide) {
            hipLaunchKernelGGL(k_radix_hist<11>, dim3(nblk), dim3(RS_TPB), 0, st, kin, n, shift, ws.counts, nblk, hi, nsb, stamp_next("radix_hist"));
            if (!scan_free) hipLaunchKernelGGL(k_scan_rows, dim3(2048), dim3(RS_TPB), 0, st, ws.counts, nblk, ws.totals, stamp_next("radix_scan"));
            if (iota) RS_SCATTER(true, true, 11); else RS_SCATTER(false, true, 11);        // wide implies staged
        } else {
            hipLaunchKernelGGL(k_radix_hist<8>, dim3(nblk), dim3(RS_TPB), 0, st, kin, n, shift, ws.counts, nblk, hi, nsb, stamp_next("radix_hist"));
            if (!scan_free) hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(RS_TPB), 0, st, ws.counts, nblk, ws.totals, stamp_next("radix_scan"));
            if (iota) { if (staged) RS_SCATTER(true, true, 8); else RS_SCATTER(true, false, 8); }
            else { if (staged) RS_SCATTER(false, true, 8); else RS_SCATTER(false, false, 8); }
        }
#undef RS_SCATTER
        uint32_t *t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    *keys_res = kin; *vals_res = vin;  // where the sorted pairs ended up (keys/vals or the alt buffers)
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// The segmented sort's launches (see k_bag_scan): offsets -> pre / ftotal (and the sort's totals zeroed) is enqueued by
// seg_sort_scan; the key kernel then fills (kp, vp); seg_sort_pairs sorts them into (keys_out, vals_out) [n], compact.
// ps_tune_set("mh_presort", v): where the multi-hot step's scan / key kernel / first sort pass run (ps_model.hip enqueue_forward).
//   0  behind the join with the training stream, beside the gather (round 4)                                   0.387-0.389 ms / step
//   1  on side chain 0 at once, no join: they land beside the PREVIOUS step's FC chain and slow its GEMMs       0.394-0.402
//   2  ... held until the previous step's embedding backward has started                                        0.380-0.384
//   3  ... and the sort's second half released by the first forward GEMM's start (default)                      0.376-0.378
// (tools/r05_mh_check.sh, interleaved on one box: profiles/r05_mh_presort_ab.txt)
int g_mh_presort = 3;
int g_seg_fused = 1;        // ps_tune_set("seg_fused", 0): build_segments as two launches (count, emit) again
int g_mh_seg_sort = 1;      // ps_tune_set("mh_seg_sort", 0): multi-hot batches through the three-pass radix sort on 22-bit keys again (round 3)
int seg_sort_alloc(SegSortWs &ws, int64_t nnz_cap, int64_t nbags_cap, int F) {
    seg_sort_free(ws);
    ws.cap = nnz_cap + (int64_t)F * RS_TILE;               // every field padded to whole tiles
    ws.ntile = (int)(ws.cap / (RS_TILE / 2)) + 2;          // (tile counts of either pass: the second pass may use half-size tiles)
    ws.F = F;
    // (all or nothing: a partial workspace would make the caller skip the allocation next time and sort through null buffers)
    struct { uint32_t **p; size_t n; } want[] = {
        {&ws.pre, (size_t)(nbags_cap + 1)}, {&ws.ftotal, 64}, {&ws.ftot, 2 * (size_t)F * SG_ND}, {&ws.tcounts, (size_t)ws.ntile * SG_ND},
        {&ws.kp, (size_t)ws.cap}, {&ws.vp, (size_t)ws.cap}, {&ws.kq, (size_t)ws.cap}, {&ws.vq, (size_t)ws.cap}};
    for (auto &w : want)
        if (hipMalloc((void **)w.p, sizeof(uint32_t) * w.n) != hipSuccess) {
            (void)hipGetLastError();
            seg_sort_free(ws);
            return ps_set_err(PS_E_HIP, "segmented sort workspace: hipMalloc of %zu bytes failed", sizeof(uint32_t) * w.n);
        }
    // (nothing to initialise: k_bag_scan writes ftotal[0 .. F) and zeroes ftot in front of every sort.  A hipMemset here is
    //  NOT ordered with the store's non-blocking stream: it zeroed ftotal after the first step's scan had filled it -- one
    //  run in ten of a test whose model trains once)
    return PS_OK;
}
void seg_sort_free(SegSortWs &ws) {
    uint32_t *ps[] = {ws.pre, ws.ftotal, ws.ftot, ws.tcounts, ws.kp, ws.vp, ws.kq, ws.vq};
    for (uint32_t *p : ps) if (p) (void)hipFree(p);
    ws = SegSortWs();
}
int seg_sort_tile() { return RS_TILE; }
int64_t seg_sort_bytes(const SegSortWs &ws) { return (int64_t)sizeof(uint32_t) * (4 * ws.cap + (int64_t)ws.ntile * SG_ND + 2 * (int64_t)ws.F * SG_ND); }
bool seg_sort_fits(const int64_t *rows_per_field, int F) {
    if (F > 64) return false;
    for (int f = 0; f < F; ++f) if (rows_per_field[f] > (1 << (2 * SG_DB))) return false;      // two 9-bit passes cover the field's ids
    return true;
}
int seg_sort_scan(SegSortWs &ws, const int64_t *offsets_dev, int B, int F, hipStream_t st) {
    if (F != ws.F) return ps_set_err(PS_E_BAD_ARG, "seg_sort_scan: %d fields, workspace made for %d", F, ws.F);
    hipLaunchKernelGGL(k_bag_scan, dim3(F), dim3(BS_TPB), 0, st, offsets_dev, B, F, ws.pre, ws.ftotal, ws.ftot, 2 * F * SG_ND, stamp_next("bag_scan"));
    HIPCHK(hipGetLastError());
    return PS_OK;
}
// which: 1 = the first pass only ((kp, vp) -> (kq, vq): touches the workspace alone), 2 = the second pass only ((kq, vq) ->
// (keys_out, vals_out)), 3 = both
int seg_sort_pairs(SegSortWs &ws, int64_t n, const int64_t *row_base_dev, uint32_t *keys_out, uint32_t *vals_out, hipStream_t st, int which) {
    if (n + (int64_t)ws.F * RS_TILE > ws.cap) return ps_set_err(PS_E_BAD_ARG, "seg_sort_pairs: n=%lld > cap", (long long)n);
    if (n <= 0) return PS_OK;
    // The two passes use different tiles over the same padded layout (fields padded to whole RS_TILE tiles).  The FIRST pass runs
    // beside the previous step's backward (mh_presort): large tiles -- half as many resident workgroups cost that backward less
    // (per-key reduce + Ftrl 98 -> 82 us).  The SECOND pass is on the step's own critical path beside the forward GEMMs: with
    // 8192-pair tiles its scatter took 69 us instead of 39 and the sort chain, not the FC chain, ended the forward phase.
    constexpr int IPT2 = RS_IPT >= 32 ? RS_IPT / 2 : RS_IPT;
    const int grid1 = (int)cdiv(n, RS_TILE) + ws.F;         // upper bounds on the tiles of the padded layout
    const int grid2 = (int)cdiv(n, RS_TPB * IPT2) + ws.F * (RS_IPT / IPT2);
    SegSortArgs a{ws.F, ws.ftotal, row_base_dev, ws.ftot, ws.tcounts};
    if (which & 1) {
    hipLaunchKernelGGL((k_seg_hist<RS_IPT>), dim3(grid1), dim3(RS_TPB), 0, st, ws.kp, 0, a, stamp_next("radix_hist"));
    hipLaunchKernelGGL((k_seg_scatter<false, RS_IPT>), dim3(grid1), dim3(RS_TPB), 0, st, ws.kp, ws.vp, ws.kq, ws.vq, 0, a, stamp_next("radix_scatter"));
    }
    a.ftot = ws.ftot + (size_t)ws.F * SG_ND;
    if (which & 2) {
    hipLaunchKernelGGL((k_seg_hist<IPT2>), dim3(grid2), dim3(RS_TPB), 0, st, ws.kq, SG_DB, a, stamp_next("radix_hist"));
    hipLaunchKernelGGL((k_seg_scatter<true, IPT2>), dim3(grid2), dim3(RS_TPB), 0, st, ws.kq, ws.vq, keys_out, vals_out, SG_DB, a, stamp_next("radix_scatter"));
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int build_segments(SortWorkspace &ws, const uint32_t *keys_sorted, int64_t n, uint32_t *seg_start,
                   uint32_t *seg_id, uint32_t *nseg_dev, hipStream_t st, uint32_t *long_list, int long_min) {
    if (n > ws.cap) return ps_set_err(PS_E_BAD_ARG, "build_segments: n > cap");
    if (n <= 0) {
        HIPCHK(hipMemsetAsync(nseg_dev, 0, 2 * sizeof(uint32_t), st));
        return PS_OK;
    }
    const int nblk = cdiv(n, RS_TILE);
    if (g_seg_fused && ws.seg_pub) {
        // Under stream capture (cfg.use_graph) the launch's sequence number is frozen into the graph: a replay would find
        // its own tags of the previous replay in pub[] and a look-back could read the previous batch's head counts
        // (ADVICE r5).  A captured launch therefore zeroes the words itself -- the memset is a node of the same graph, in
        // front of the kernel, and every replay starts from "no launch yet" whatever ran on the workspace before.
        if (stream_is_capturing(st)) HIPCHK(hipMemsetAsync(ws.seg_pub, 0, sizeof(unsigned long long) * (size_t)(ws.nblk + 1), st));
        if (ws.seg_seq == 0) HIPCHK(hipMemsetAsync(ws.seg_pub, 0, sizeof(unsigned long long) * (size_t)(ws.nblk + 1), st));     // (tag 0 = no launch yet)
        if (++ws.seg_seq == 0) { HIPCHK(hipMemsetAsync(ws.seg_pub, 0, sizeof(unsigned long long) * (size_t)(ws.nblk + 1), st)); ++ws.seg_seq; }
        hipLaunchKernelGGL(k_seg_fused, dim3(nblk), dim3(RS_TPB), 0, st, keys_sorted, n, ws.seg_pub, ws.seg_seq, seg_start, seg_id, nseg_dev, stamp_next("seg_emit"));
    } else {
    hipLaunchKernelGGL(k_seg_count, dim3(nblk), dim3(RS_TPB), 0, st, keys_sorted, n, ws.blk_heads);
    hipLaunchKernelGGL(k_seg_emit, dim3(nblk), dim3(RS_TPB), 0, st, keys_sorted, n, ws.blk_heads,
                       seg_start, seg_id, nseg_dev, stamp_next("seg_emit"));
    }
    if (long_list)      // at most n / (long_min + 1) runs can be that long; one thread per run, any order
        hipLaunchKernelGGL(k_long_runs, dim3(cdiv(n, 256)), dim3(256), 0, st, seg_start, nseg_dev, long_list, (uint32_t)long_min);
    HIPCHK(hipGetLastError());
    return PS_OK;
}


// ---------------------------------------------------------------------------
// field_sort_segments: see k_field_sort_segments.  keys = [B][F] row keys of a single-hot batch.
// ---------------------------------------------------------------------------
bool field_sort_fits(int B, int F) { return B >= 1 && B <= FS_MAX && F >= 1 && F < 65536; }

int field_sort_segments(const uint32_t *keys, const int64_t *keys_base, int key_bits, int B, int F, int long_min,
                        uint32_t *sorted_keys, uint32_t *sorted_ents, uint32_t *seg_start, uint32_t *seg_id,
                        uint32_t *nseg_dev, uint32_t *long_list, unsigned long long *pub, uint32_t epoch, hipStream_t st,
                        unsigned int *start_flag, unsigned int start_val) {
    if (!field_sort_fits(B, F) || key_bits < 1 || key_bits > 32)
        return ps_set_err(PS_E_BAD_ARG, "field_sort_segments: B = %d, F = %d, key_bits = %d", B, F, key_bits);
    int ept = 1;
    while (FS_TPB * ept < B) ept <<= 1;
    const int NP = FS_TPB * ept;
    const int npass = (key_bits + 7) / 8, digit_bits = (key_bits + npass - 1) / npass;
    FieldSortArgs a{keys, keys_base, B, F, NP, long_min, npass, digit_bits, sorted_keys, sorted_ents, seg_start, seg_id, nseg_dev,
                    long_list, pub, epoch, stamp_next("field_sort"), start_flag, start_val};
    // (stream capture: the epoch is frozen into the graph -- every replay zeroes the look-back words first; build_segments' comment)
    if (stream_is_capturing(st)) HIPCHK(hipMemsetAsync(pub, 0, sizeof(unsigned long long) * (size_t)F, st));
    const size_t lds = (size_t)NP * 16;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_field_sort_segments<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   FS_MAX * 16));
        attr_set = true;
    }
    switch (ept) {
    case 1: hipLaunchKernelGGL(k_field_sort_segments<1>, dim3(F), dim3(FS_TPB), lds, st, a); break;
    case 2: hipLaunchKernelGGL(k_field_sort_segments<2>, dim3(F), dim3(FS_TPB), lds, st, a); break;
    case 4: hipLaunchKernelGGL(k_field_sort_segments<4>, dim3(F), dim3(FS_TPB), lds, st, a); break;
    default: hipLaunchKernelGGL(k_field_sort_segments<8>, dim3(F), dim3(FS_TPB), lds, st, a); break;
    }
    HIPCHK(hipGetLastError());
    return PS_OK;
}

#ifdef PS_FS_TIMING
extern "C" int ps_dbg_fs_timing(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fs_t), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif
