// ps_eval.hip -- evaluate/AUC.java:32-82 on the device (the predict / AUC path).
//
// The reference sorts the (p, y) pairs ascending by p (Arrays.sort on objects = stable), walks them from
// the highest p down, and adds (x - prev) * y at every NEGATIVE sample, where x = #negatives so far / posNum
// and y = #positives so far / negNum (its tp/fp names are swapped, evaluate/AUC.java:52-62).  Each negative
// moves x by 1/posNum, so the sum is
//
//     AUC = ( sum over negatives of  #positives ranked above it ) / (posNum * negNum)
//
// = the fraction of (positive, negative) pairs ranked correctly, ties broken by the stable sort (a tie
// counts as correct iff the positive came later in the input).  The numerator is an integer: computed here
// exactly (stable radix sort of the float bits + a blocked prefix count), then ONE double division.  The
// reference's own double accumulation differs from that by rounding only (~1e-16 relative).
#include <string.h>

#include "ps_store.h"

namespace {

// order-preserving map float -> u32 = the order of Double.compareTo on the widened values
// (evaluate/AUC.java:36: a total order, -0.0 < 0.0, every NaN equal and above +Infinity)
__global__ __launch_bounds__(256) void k_auc_keys(const float *__restrict__ p, int64_t n, uint32_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = __float_as_uint(p[i]);
    keys[i] = (p[i] != p[i]) ? 0xFFFFFFFFu : (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

#define AUC_TILE 2048   // sorted positions per workgroup (8 per thread)

// positives per tile of the sorted order
__global__ __launch_bounds__(256) void k_auc_tile_pos(const uint32_t *__restrict__ sorted_ent, const float *__restrict__ y, int64_t n,
                                                      uint32_t *__restrict__ tile_pos) {
    __shared__ uint32_t s[4];
    const int64_t base = (int64_t)blockIdx.x * AUC_TILE;
    uint32_t c = 0;
    for (int j = 0; j < AUC_TILE / 256; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const int64_t ic = i < n ? i : n - 1;
        const float lab = y[sorted_ent[ic]];
        c += (i < n && lab > 0.f) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_pos[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of the tile counts (one workgroup; ntiles is small: n / 2048)
__global__ __launch_bounds__(1024) void k_auc_scan(uint32_t *__restrict__ tile_pos, int64_t ntiles, unsigned long long *__restrict__ totals) {
    __shared__ unsigned long long s[1024];
    unsigned long long carry = 0;
    for (int64_t base = 0; base < ntiles; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const unsigned long long v = i < ntiles ? tile_pos[i] : 0ull;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned long long t = threadIdx.x >= off ? s[threadIdx.x - off] : 0ull;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const unsigned long long incl = s[threadIdx.x];
        if (i < ntiles) tile_pos[i] = (uint32_t)(carry + incl - v);     // positives before this tile (fits: n < 2^32)
        const unsigned long long tot = s[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) totals[0] = carry;      // posNum
}

// per negative: positives ranked ABOVE it = posNum - positives at or below it; summed exactly in u64
__global__ __launch_bounds__(256) void k_auc_pairs(const uint32_t *__restrict__ sorted_ent, const float *__restrict__ y, int64_t n,
                                                   const uint32_t *__restrict__ tile_before, unsigned long long *__restrict__ totals) {
    __shared__ uint32_t wave_pos[4];
    __shared__ unsigned long long red[4];
    const int64_t base = (int64_t)blockIdx.x * AUC_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long P = totals[0];
    uint32_t before = tile_before[blockIdx.x];     // positives at sorted positions < the current row of 256
    unsigned long long acc = 0;
    for (int j = 0; j < AUC_TILE / 256; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const int64_t ic = i < n ? i : n - 1;
        const float lab = y[sorted_ent[ic]];
        const bool valid = i < n, pos = valid && lab > 0.f;
        const unsigned long long bal = __ballot(pos);
        const uint32_t in_wave_before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_pos[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t wbefore = 0;
        for (int w = 0; w < wave; ++w) wbefore += wave_pos[w];
        const uint32_t row_total = wave_pos[0] + wave_pos[1] + wave_pos[2] + wave_pos[3];
        if (valid && !pos) acc += P - (unsigned long long)(before + wbefore + in_wave_before);
        before += row_total;
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&totals[1], red[0] + red[1] + red[2] + red[3]);     // integer: order-free
}

}  // namespace

extern "C" int ps_auc_compute(ps_store_t *s, const float *p, const float *y, int64_t n, int on_device,
                              double *auc, int64_t *pos_num, int64_t *neg_num) {
    if (!s || !auc || n < 0 || (n > 0 && (!p || !y))) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (n >= (1ll << 32)) return ps_set_err(PS_E_UNSUPPORTED, "AUC over %lld samples (limit 2^32 - 1)", (long long)n);
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    if (n == 0) {       // sampleCount leaves 0/0: the reference's loop adds nothing -> 0.0
        *auc = 0.0;
        if (pos_num) *pos_num = 0;
        if (neg_num) *neg_num = 0;
        return PS_OK;
    }
    float *pd = nullptr, *yd = nullptr;
    uint32_t *keys = nullptr, *ents = nullptr, *tile = nullptr;
    unsigned long long *totals = nullptr;
    SortWorkspace ws;
    int rc = PS_OK;
    auto cleanup = [&]() {
        if (!on_device) { if (pd) (void)hipFree(pd); if (yd) (void)hipFree(yd); }
        if (keys) (void)hipFree(keys);
        if (ents) (void)hipFree(ents);
        if (tile) (void)hipFree(tile);
        if (totals) (void)hipFree(totals);
        sort_ws_free(ws);
    };
#define EV(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { rc = ps_set_err(PS_E_HIP, "%s -> %s", #x, hipGetErrorString(e__)); cleanup(); return rc; } } while (0)
    if (on_device) { pd = const_cast<float *>(p); yd = const_cast<float *>(y); }
    else {
        EV(hipMalloc((void **)&pd, sizeof(float) * n));
        EV(hipMalloc((void **)&yd, sizeof(float) * n));
        EV(hipMemcpyAsync(pd, p, sizeof(float) * n, hipMemcpyHostToDevice, st));
        EV(hipMemcpyAsync(yd, y, sizeof(float) * n, hipMemcpyHostToDevice, st));
    }
    const int64_t ntiles = (n + AUC_TILE - 1) / AUC_TILE;
    EV(hipMalloc((void **)&keys, sizeof(uint32_t) * (n + 1)));
    EV(hipMalloc((void **)&ents, sizeof(uint32_t) * (n + 1)));
    EV(hipMalloc((void **)&tile, sizeof(uint32_t) * (ntiles + 1)));
    EV(hipMalloc((void **)&totals, sizeof(unsigned long long) * 2));
    EV(hipMemsetAsync(totals, 0, sizeof(unsigned long long) * 2, st));
    rc = sort_ws_alloc(ws, n);
    if (rc != PS_OK) { cleanup(); return rc; }
    hipLaunchKernelGGL(k_auc_keys, dim3(cdiv(n, 256)), dim3(256), 0, st, pd, n, keys);
    uint32_t *sk = nullptr, *se = nullptr;
    rc = radix_sort_pairs(ws, keys, ents, n, 32, true, &sk, &se, st);       // stable: ties keep input order (Arrays.sort)
    if (rc != PS_OK) { cleanup(); return rc; }
    hipLaunchKernelGGL(k_auc_tile_pos, dim3((unsigned)ntiles), dim3(256), 0, st, se, yd, n, tile);
    hipLaunchKernelGGL(k_auc_scan, dim3(1), dim3(1024), 0, st, tile, ntiles, totals);
    hipLaunchKernelGGL(k_auc_pairs, dim3((unsigned)ntiles), dim3(256), 0, st, se, yd, n, tile, totals);
    EV(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    EV(hipMemcpyAsync(h, totals, sizeof h, hipMemcpyDeviceToHost, st));
    EV(hipStreamSynchronize(st));
#undef EV
    cleanup();
    const double P = (double)h[0], Nn = (double)(n - (int64_t)h[0]);
    if (pos_num) *pos_num = (int64_t)h[0];
    if (neg_num) *neg_num = n - (int64_t)h[0];
    // degenerate label sets, as the reference's arithmetic leaves them (evaluate/AUC.java:52-80): no negatives ->
    // x stays 0/posNum = 0 and nothing is ever added (0.0); no positives -> x = tp/0 = Infinity, times y = 0 -> NaN
    if (h[0] == 0) *auc = (double)NAN;
    else if (Nn == 0) *auc = 0.0;
    else *auc = (double)h[1] / (P * Nn);
    return PS_OK;
}
