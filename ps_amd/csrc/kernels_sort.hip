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

// in-place exclusive scan of `total` u32 by ONE workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_scan_exclusive(uint32_t *__restrict__ a, int total) {
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (total + 1023) / 1024;
    const int lo = tid * per;
    const int hi = lo + per < total ? lo + per : total;
    uint32_t s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    // wave-level inclusive scan with shuffles, then across the 16 waves
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int i = 0; i < w; ++i) wbase += wsum[i];
    uint32_t run = wbase + inc - s;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const uint32_t v = a[i];
        a[i] = run;
        run += v;
    }
}

// stable scatter: wave w of the block owns keys [blk*4096 + w*1024, +1024) and
// walks them in 16 batches of 64 in index order, ranking equal digits inside
// a batch with ballots (lanes in increasing order) and across batches with a
// running per-(wave,digit) cursor in LDS.
template <bool IOTA>
__global__ __launch_bounds__(RS_TPB) void k_radix_scatter(const uint32_t *__restrict__ kin,
                                                          const uint32_t *__restrict__ vin,
                                                          uint32_t *__restrict__ kout,
                                                          uint32_t *__restrict__ vout, int64_t n,
                                                          int shift,
                                                          const uint32_t *__restrict__ offs,
                                                          int nblk) {
    __shared__ uint32_t cur[4][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 1024; i += RS_TPB) ((uint32_t *)cur)[i] = 0;
    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * RS_WAVE_SPAN;
    uint32_t k[RS_IPT], v[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        const int64_t ci = idx < n ? idx : n - 1;
        k[j] = kin[ci];
        v[j] = IOTA ? (uint32_t)idx : vin[ci];
    }
    const uint32_t goff = offs[(size_t)tid * nblk + blockIdx.x];   // digit tid's global cursor for this block
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; ++j) {
        const int64_t idx = wbase + j * 64 + lane;
        if (idx < n) atomicAdd(&cur[w][(k[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {   // thread tid owns digit tid: turn per-wave counts into per-wave global cursors
        uint32_t g = goff;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const uint32_t c = cur[ww][tid];
            cur[ww][tid] = g;
            g += c;
        }
    }
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
            kout[pos] = k[j];
            vout[pos] = v[j];
        }
        if (valid && rank + 1 == cnt) cur[w][d] += cnt;  // last lane of the group advances the cursor
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
    return PS_OK;
}

void sort_ws_free(SortWorkspace &ws) {
    if (ws.keys_alt) (void)hipFree(ws.keys_alt);
    if (ws.vals_alt) (void)hipFree(ws.vals_alt);
    if (ws.counts) (void)hipFree(ws.counts);
    if (ws.blk_heads) (void)hipFree(ws.blk_heads);
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
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, ws.counts, 256 * nblk);
        if (p == 0 && iota_vals)
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nblk), dim3(RS_TPB), 0, st, kin, (const uint32_t *)vin, kout, vout, n, shift, ws.counts, nblk);
        else
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nblk), dim3(RS_TPB), 0, st, kin, (const uint32_t *)vin, kout, vout, n, shift, ws.counts, nblk);
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
