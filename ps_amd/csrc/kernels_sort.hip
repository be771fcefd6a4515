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
constexpr int RS_IPT = 16;
constexpr int RS_TILE = RS_TPB * RS_IPT;  // 4096 keys per workgroup
constexpr int RS_WAVE_SPAN = RS_TILE / 4; // 1024 consecutive keys per wave

// per-block digit histogram; counts is digit-major: counts[digit * nblk + blk]
__global__ __launch_bounds__(RS_TPB) void k_radix_hist(const uint32_t *__restrict__ keys, int64_t n,
                                                       int shift, uint32_t *__restrict__ counts,
                                                       int nblk) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
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
        if (idx < n) atomicAdd(&h[(k[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)tid * nblk + blockIdx.x] = h[tid];
}

// counts is digit-major [256][nblk].  Workgroup d turns row d into its exclusive prefix (positions of digit d's
// keys of block b among all keys with digit d) and leaves the row total in totals[d].  One workgroup per digit:
// the scan scales with the key count (a single-workgroup scan of 256*nblk counters was THE cost at 3e6 keys).
__global__ __launch_bounds__(RS_TPB) void k_scan_rows(uint32_t *__restrict__ counts, int nblk, uint32_t *__restrict__ totals) {
    __shared__ uint32_t wsum[4];
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
template <bool IOTA, bool STAGED>
__global__ __launch_bounds__(RS_TPB) void k_radix_scatter(const uint32_t *__restrict__ kin,
                                                          const uint32_t *__restrict__ vin,
                                                          uint32_t *__restrict__ kout,
                                                          uint32_t *__restrict__ vout, int64_t n,
                                                          int shift,
                                                          const uint32_t *__restrict__ offs,
                                                          const uint32_t *__restrict__ totals,
                                                          int nblk) {
    __shared__ uint32_t cur[4][256];
    __shared__ uint32_t gdelta[256];       // global position - block-local position, per digit
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t sk[STAGED ? RS_TILE : 1], sv[STAGED ? RS_TILE : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 1024; i += RS_TPB) ((uint32_t *)cur)[i] = 0;
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
    const uint32_t in_digit = offs[(size_t)tid * nblk + blockIdx.x];   // keys of digit tid in earlier blocks
    const uint32_t dtot = totals[tid];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        if (idx < n) atomicAdd(&cur[w][(k[j] >> shift) & 255u], 1u);
    }
    // exclusive scan of the 256 digit totals (global digit bases) and of this block's digit counts (local bases)
    uint32_t ginc = dtot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(ginc, off);
        if (lane >= off) ginc += t;
    }
    if (lane == 63) wtot[w] = ginc;
    __syncthreads();
    uint32_t gbase = ginc - dtot;
    for (int ww = 0; ww < w; ++ww) gbase += wtot[ww];
    const uint32_t c0 = cur[0][tid], c1 = cur[1][tid], c2 = cur[2][tid], c3 = cur[3][tid];
    const uint32_t btot = c0 + c1 + c2 + c3;
    uint32_t linc = btot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(linc, off);
        if (lane >= off) linc += t;
    }
    __syncthreads();                       // wtot is re-used
    if (lane == 63) wtot[w] = linc;
    __syncthreads();
    uint32_t lbase = linc - btot;
    for (int ww = 0; ww < w; ++ww) lbase += wtot[ww];
    cur[0][tid] = lbase; cur[1][tid] = lbase + c0; cur[2][tid] = lbase + c0 + c1; cur[3][tid] = lbase + c0 + c1 + c2;
    gdelta[tid] = gbase + in_digit - lbase;
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (k[j] >> shift) & 255u;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
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
            const uint32_t gp = lp + gdelta[(key >> shift) & 255u];
            kout[gp] = key;
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

__global__ __launch_bounds__(RS_TPB) void k_seg_emit(const uint32_t *__restrict__ keys, int64_t n,
                                                     const uint32_t *__restrict__ blk_heads,
                                                     uint32_t *__restrict__ seg_start,
                                                     uint32_t *__restrict__ seg_id,
                                                     uint32_t *__restrict__ nseg_dev) {
    __shared__ uint32_t red[4];
    __shared__ uint32_t wave_heads[4];
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
                *nseg_dev = sid + 1;
            }
        }
        run += (uint32_t)__popcll(hm);
    }
}

}  // namespace

int sort_ws_alloc(SortWorkspace &ws, int64_t cap) {
    sort_ws_free(ws);
    ws.cap = cap;
    ws.nblk = cdiv(cap > 0 ? cap : 1, RS_TILE);
    HIPCHK(hipMalloc(&ws.keys_alt, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc(&ws.vals_alt, sizeof(uint32_t) * (size_t)(cap + 1)));
    HIPCHK(hipMalloc(&ws.counts, sizeof(uint32_t) * 256 * (size_t)ws.nblk));
    HIPCHK(hipMalloc(&ws.blk_heads, sizeof(uint32_t) * (size_t)ws.nblk));
    HIPCHK(hipMalloc(&ws.totals, sizeof(uint32_t) * 256));
    return PS_OK;
}

void sort_ws_free(SortWorkspace &ws) {
    if (ws.keys_alt) (void)hipFree(ws.keys_alt);
    if (ws.vals_alt) (void)hipFree(ws.vals_alt);
    if (ws.counts) (void)hipFree(ws.counts);
    if (ws.blk_heads) (void)hipFree(ws.blk_heads);
    if (ws.totals) (void)hipFree(ws.totals);
    ws = SortWorkspace();
}

int radix_sort_pairs(SortWorkspace &ws, uint32_t *keys, uint32_t *vals, int64_t n, int key_bits,
                     bool iota_vals, uint32_t **keys_res, uint32_t **vals_res, hipStream_t st) {
    if (n > ws.cap) return ps_set_err(PS_E_BAD_ARG, "radix_sort_pairs: n=%lld > cap=%lld", (long long)n, (long long)ws.cap);
    *keys_res = keys; *vals_res = vals;
    if (n <= 0) return PS_OK;
    int passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    const int nblk = cdiv(n, RS_TILE);
    uint32_t *kin = keys, *vin = vals, *kout = ws.keys_alt, *vout = ws.vals_alt;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(RS_TPB), 0, st, kin, n, shift, ws.counts, nblk);
        hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(RS_TPB), 0, st, ws.counts, nblk, ws.totals);
        // LDS staging pays once the scatter is bandwidth-bound (measured: 3.2 M pairs 3x faster, 1e5 pairs 25 % slower)
        const bool staged = n >= (1 << 19);
        const bool iota = p == 0 && iota_vals;
#define RS_SCATTER(I, S) hipLaunchKernelGGL((k_radix_scatter<I, S>), dim3(nblk), dim3(RS_TPB), 0, st, kin, (const uint32_t *)vin, kout, vout, n, shift, ws.counts, ws.totals, nblk)
        if (iota) { if (staged) RS_SCATTER(true, true); else RS_SCATTER(true, false); }
        else { if (staged) RS_SCATTER(false, true); else RS_SCATTER(false, false); }
#undef RS_SCATTER
        uint32_t *t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    *keys_res = kin; *vals_res = vin;  // where the sorted pairs ended up (keys/vals or the alt buffers)
    HIPCHK(hipGetLastError());
    return PS_OK;
}

int build_segments(SortWorkspace &ws, const uint32_t *keys_sorted, int64_t n, uint32_t *seg_start,
                   uint32_t *seg_id, uint32_t *nseg_dev, hipStream_t st) {
    if (n > ws.cap) return ps_set_err(PS_E_BAD_ARG, "build_segments: n > cap");
    if (n <= 0) {
        HIPCHK(hipMemsetAsync(nseg_dev, 0, sizeof(uint32_t), st));
        return PS_OK;
    }
    const int nblk = cdiv(n, RS_TILE);
    hipLaunchKernelGGL(k_seg_count, dim3(nblk), dim3(RS_TPB), 0, st, keys_sorted, n, ws.blk_heads);
    hipLaunchKernelGGL(k_seg_emit, dim3(nblk), dim3(RS_TPB), 0, st, keys_sorted, n, ws.blk_heads,
                       seg_start, seg_id, nseg_dev);
    HIPCHK(hipGetLastError());
    return PS_OK;
}
