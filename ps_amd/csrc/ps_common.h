// ps_common.h -- internal declarations shared by the HIP translation units of
// libps_amd.so (gfx950 only).  The public boundary is include/ps_native.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/ps_native.h"

int ps_set_err(int code, const char *fmt, ...);

// Creation / destruction of HIP runtime objects (streams, events, device and pinned allocations) is serialised
// process-wide: several host threads may each drive their own store (tests run N ranks as N threads on one GPU),
// and concurrent hipStreamCreate / hipMalloc / hipFree from many threads is the one place where the threads meet
// inside the runtime.  Never held across a kernel launch chain, a collective or a callback.
#include <mutex>
std::recursive_mutex &ps_rt_mutex();
struct RtGuard {
    std::lock_guard<std::recursive_mutex> g;
    RtGuard() : g(ps_rt_mutex()) {}
};

#define HIPCHK(x)                                                                       \
    do {                                                                                \
        hipError_t e__ = (x);                                                           \
        if (e__ != hipSuccess)                                                          \
            return ps_set_err(PS_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #x,       \
                              hipGetErrorString(e__));                                  \
    } while (0)
#define PSCHK(x)                     \
    do {                             \
        int r__ = (x);               \
        if (r__ != PS_OK) return r__; \
    } while (0)

// roctx ranges around the C-ABI entry points (visible in rocprofv3 --marker-trace timelines).  Off unless
// PS_AMD_ROCTX=1: libroctx64 is dlopen'ed on first use, a range costs one predictable branch otherwise.
void ps_roctx_push(const char *name);
void ps_roctx_pop();
struct RoctxRange {
    RoctxRange(const char *name) { ps_roctx_push(name); }
    ~RoctxRange() { ps_roctx_pop(); }
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// the stream is recording a hipGraph (ps_model.hip train_graph): kernel arguments are frozen, device memory is not
static inline bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs == hipStreamCaptureStatusActive;
}

// ---------------------------------------------------------------------------
// counter-based row init: +-U(0, scale) as a pure function of
// (seed, table, row, col).  The test oracle restates this definition on the
// CPU; nothing is shared with it at include or link level.
// Replaces util/MatrixUtil.java:62-74 (unseeded RandomUtils).
// ---------------------------------------------------------------------------
#define PS_TABLE_WIDE (1ull << 20)
#define PS_TABLE_WIDE_B ((1ull << 20) + 1)
#define PS_TABLE_FC(i) ((2ull << 20) + 2ull * (uint64_t)(i))

__host__ __device__ inline uint64_t ps_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline float ps_init_value(uint64_t seed, uint64_t table, uint64_t row,
                                               uint64_t col, float scale) {
    uint64_t h = ps_splitmix64(seed + 0x9E3779B97F4A7C15ull * (table + 1));
    h = ps_splitmix64(h ^ row);
    h = ps_splitmix64(h ^ (col * 0xD6E8FEB86659FD93ull));
    float u = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
    float m = u * scale;
    return ((h >> 39) & 1) ? -m : m;
}
float ps_xavier_scale(int in_dims, int out_dims);

// ---------------------------------------------------------------------------
// device-side updater parameters
// ---------------------------------------------------------------------------
struct UpdParams {
    int kind;  // PS_UPD_*
    float alfa, beta1, beta2, eps, c1, c2, neg_alfa;  // adam (c1 = 1-beta1, c2 = 1-beta2 in f32)
    float beta, l1, l2;                               // ftrl
    float eta;
};
UpdParams make_upd_params(const ps_updater_t &u);

// ---------------------------------------------------------------------------
// sort / segment primitives (kernels_sort.hip)
// ---------------------------------------------------------------------------
struct SortWorkspace {
    uint32_t *keys_alt = nullptr, *vals_alt = nullptr;  // ping-pong buffers [cap]
    uint32_t *counts = nullptr;                         // [2048 * nblk] (8- or 11-bit digits)
    uint32_t *blk_heads = nullptr;                      // [nblk]
    uint32_t *totals = nullptr;                         // [2048] keys per digit of the current pass
    uint32_t *hi = nullptr;                             // [passes][digits][superblocks] second level of counts (scan-free passes)
    unsigned long long *seg_pub = nullptr; uint32_t seg_seq = 0;     // build_segments in one launch: per-workgroup head counts tagged with the launch's number (look-back)
    int64_t cap = 0;
    int nblk = 0;
};
// the segmented (per-field) two-pass sort of a multi-hot batch (kernels_sort.hip k_bag_scan / k_seg_hist / k_seg_scatter)
struct SegSortWs {
    uint32_t *pre = nullptr;        // [bags] entries of the bag's field in earlier samples
    uint32_t *ftotal = nullptr;     // [64] entries per field
    uint32_t *ftot = nullptr;       // [2 passes][F][512] digit totals per field
    uint32_t *tcounts = nullptr;    // [tiles][512] digit counts per tile
    uint32_t *kp = nullptr, *vp = nullptr, *kq = nullptr, *vq = nullptr;      // (id, bag) pairs, every field padded to whole tiles
    int64_t cap = 0; int ntile = 0, F = 0;
};
int seg_sort_alloc(SegSortWs &ws, int64_t nnz_cap, int64_t nbags_cap, int F);
void seg_sort_free(SegSortWs &ws);
int64_t seg_sort_bytes(const SegSortWs &ws);
int seg_sort_tile();
bool seg_sort_fits(const int64_t *rows_per_field, int F);
int seg_sort_scan(SegSortWs &ws, const int64_t *offsets_dev, int B, int F, hipStream_t st);
int seg_sort_pairs(SegSortWs &ws, int64_t n, const int64_t *row_base_dev, uint32_t *keys_out, uint32_t *vals_out, hipStream_t st, int which = 3);
extern int g_wide_in_gather, g_emb_list_min, g_emb_list_grid;
extern int g_mh_seg_sort, g_mh_presort, g_mh_prio, g_slots_in_gather, g_keys_grid, g_emb_xcd, g_emb_lxcd, g_super_in_update, g_fwd_order, g_super_list;
int sort_ws_alloc(SortWorkspace &ws, int64_t cap);
void sort_ws_free(SortWorkspace &ws);
// Stable LSD radix sort of (key, val) pairs on `key_bits` low bits.
// iota_vals: vals start as 0..n-1 (not read).  The sorted pairs end up in
// *keys_res / *vals_res (either keys/vals or the workspace's alt buffers,
// depending on the pass count) -- no copy back.
int radix_sort_pairs(SortWorkspace &ws, uint32_t *keys, uint32_t *vals, int64_t n, int key_bits,
                     bool iota_vals, uint32_t **keys_res, uint32_t **vals_res, hipStream_t st);
// Segments of equal keys in a sorted key array: seg_start[nseg+1], seg_id[n],
// *nseg_dev.
// long_list (optional): (run id, first entry, end) triples of the runs longer than long_min entries, nseg_dev[1] of them, any order.
int build_segments(SortWorkspace &ws, const uint32_t *keys_sorted, int64_t n, uint32_t *seg_start,
                   uint32_t *seg_id, uint32_t *nseg_dev, hipStream_t st, uint32_t *long_list = nullptr, int long_min = 0);
// Single-hot batch (keys = [B][F], field f's keys in their own interval): stable sort + segments + the list of runs
// longer than long_min entries, ONE launch of F workgroups.  Same sorted_keys / sorted_ents / seg_* as
// radix_sort_pairs(iota) + build_segments; nseg_dev[1] = number of long runs.  keys_base[f] (device) = first key of
// field f's interval, key_bits = bits of the largest (key - keys_base[f]).  pub: F look-back words (zeroed once),
// epoch: a value that differs from every earlier launch on the same pub (nonzero).
#define PS_FS_TAB_OFF(F) (((F) + 1) & ~1)      // the field table starts this many 8-byte words behind pub (16-byte aligned)
bool field_sort_fits(int B, int F);
int field_sort_segments(const uint32_t *keys, const int64_t *keys_base, int key_bits, int B, int F, int long_min,
                        uint32_t *sorted_keys, uint32_t *sorted_ents, uint32_t *seg_start, uint32_t *seg_id,
                        uint32_t *nseg_dev, uint32_t *long_list, unsigned long long *pub, uint32_t epoch, hipStream_t st,
                        unsigned int *start_flag = nullptr, unsigned int start_val = 0);     // start_flag: raised by the launch's first workgroup

// What one launch carries beyond its kernel arguments.  Explicit, per launch: the caller fills a LaunchOpts and hands
// it to the launcher (rounds 1-2 "armed" thread-local globals that the next launcher of the same host thread consumed).
//   stop_event  the event another stream will wait on rides on the dispatch packet's own completion signal
//               (hipExtLaunchKernelGGL) instead of a hipEventRecord behind it (a separate barrier packet: ~3-4 us of the
//               recording stream before its next kernel starts; tools/gpu_timeline.py)
//   flag        the launch's first workgroup stores flag_val there when it starts ("everything in front of me on my
//               stream has finished") for a device-side waiter (launch_spin_until)
//   wait        the launch's first workgroup ends only once *wait has reached wait_val (a join that costs no launch)
//   prio        the launch's waves run at raised priority (s_setprio)
// launched (out): a kernel was launched and took all of the above; false (an empty problem): the caller settles them.
struct LaunchOpts {
    hipEvent_t stop_event = nullptr;
    unsigned int *flag = nullptr; unsigned int flag_val = 0;
    const unsigned int *wait = nullptr; unsigned int wait_val = 0;
    int prio = 0;
    bool launched = false;
};
extern int g_main_prio;
extern int g_sort_late;
extern int g_tn_prio, g_gemm_pipe, g_gemm_ks, g_dw_late, g_keys_early;
extern int g_tn_start_wait, g_tail_fused, g_shard_overlap, g_dw_split, g_tail_defer;
extern int g_dev_wait, g_tail_dev, g_gemm_8w, g_radix11, g_end_wait, g_radix_scan_free, g_plan_early;
// Every device-side wait is BOUNDED: a waiter that has not seen its flag after g_spin_timeout_ticks (10 ns ticks of the
// device's wall clock; ps_tune_set("spin_timeout_ms")) adds 1 to *werr, stores which wait it was beside it and gives up.
// The host finds the count at its next wait on the store's stream (store_check_device), returns PS_E_STATE and switches
// the store to the event form of every join; the step that timed out has run without one of its dependencies.
extern unsigned long long g_spin_timeout_ticks;
struct WaitBound { unsigned int *err; unsigned long long ticks; unsigned int code; };
WaitBound wait_bound(unsigned int *werr, unsigned int code);
int launch_spin_until(const unsigned int *flag, unsigned int val, hipStream_t st, unsigned int *werr, unsigned int code = 0);   // a one-wave kernel that ends when *flag reached val
int launch_flag_set(unsigned int *flag, unsigned int val, hipStream_t st);             // *flag = val, in stream order
// both in ONE launch: *set = set_val when the kernel starts (what is in front of it on the stream is done), then the wait
int launch_set_then_spin(unsigned int *set, unsigned int set_val, const unsigned int *flag, unsigned int val, hipStream_t st, unsigned int *werr, unsigned int code);
// ... with an optional set (NULL: none) and TWO waits (flag2 NULL: one)
int launch_set_then_spin2(unsigned int *set, unsigned int set_val, const unsigned int *flag, unsigned int val, const unsigned int *flag2, unsigned int val2,
                          hipStream_t st, unsigned int *werr, unsigned int code);
#define PS_LAUNCH_EV(kernel, grid, block, shmem, st, ev, ...)                                                  \
    do {                                                                                                       \
        hipEvent_t se_ = (ev);                                                                                 \
        if (se_) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, nullptr, se_, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                  \
    } while (0)
// ---------------------------------------------------------------------------
// GPU-side time stamps (measurement only: ps_tune_set("stamps", 1), tools/gpu_timeline.py).  A stamped kernel takes
// a slot pointer in its arguments (nullptr when off = one scalar compare per workgroup): slot[0] = start of
// workgroup 0, slot[1] = latest sampled workgroup end, 10 ns ticks of the device's wall clock.  Unlike a profiler's
// trace this costs the HOST nothing, so the stream stays as far ahead of the GPU as in a normal run.
// ---------------------------------------------------------------------------
unsigned long long *stamp_next(const char *name);      // nullptr when stamps are off or the buffer is full
int stamps_enable(int on);
#ifdef __HIPCC__
// ---- device-side waits, all bounded (WaitBound above) ----
// After bound.ticks of the device's wall clock the waiter counts itself in *bound.err, notes which wait it was, and goes on.
// Returns whether it had to wait at all.
__device__ __forceinline__ bool spin_bounded(const unsigned int *f, unsigned int v, const WaitBound &b) {
    if ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) >= 0) return false;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (b.ticks && (unsigned long long)wall_clock64() - t0 > b.ticks) {
            if (b.err) {
                atomicAdd(b.err, 1u);
                __hip_atomic_store(b.err + 1, b.code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the last one to give up
                atomicCAS(b.err + 2, 0u, b.code + 1000u);                                               // the first one (+1000: code 0 is a wait too)
            }
            break;
        }
    }
    return true;
}
// the same wait on a word a PEER DEVICE stores to (mapped peer memory, ps_comm.hip): system-scope loads
__device__ __forceinline__ bool spin_bounded_sys(const unsigned int *f, unsigned int v, const WaitBound &b) {
    if ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - v) >= 0) return false;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - v) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (b.ticks && (unsigned long long)wall_clock64() - t0 > b.ticks) {
            if (b.err) {
                atomicAdd(b.err, 1u);
                __hip_atomic_store(b.err + 1, b.code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicCAS(b.err + 2, 0u, b.code + 1000u);
            }
            break;
        }
    }
    return true;
}
// "This launch does not END before another chain has reached X": the first workgroup, done with its work, holds its
// slot until the flag is there (normally long since) -- the join costs the waiting chain no launch of its own.  What
// the other chain wrote is read by the NEXT launch of this stream (its start acquires).
struct EndWait {
    const unsigned int *f; unsigned int v; WaitBound b;
    __device__ __forceinline__ EndWait(const unsigned int *flag, unsigned int val, const WaitBound &bound) : f(flag), v(val), b(bound) {}
    __device__ __forceinline__ ~EndWait() {
        if (f && blockIdx.x == 0 && threadIdx.x == 0) (void)spin_bounded(f, v, b);
    }
};
// "This launch does not START its work before X": every workgroup checks the flag when it starts.  The intended case is
// one load: the launch sits in order behind a kernel that outlasts X, so the flag is up -- and then this kernel's own
// start (which acquires: an XCD's L2 drops its non-coherent lines) came AFTER the data X stands for was released, and
// nothing else is needed.  A workgroup that really had to wait acquires at agent scope afterwards: X's data was written by
// a kernel of another stream, and this XCD's L2 may hold older lines of those addresses.  (The fence is a buffer_inv
// sc1 = an L2 invalidate: paid unconditionally by every workgroup of a 224-workgroup GEMM it cost the step 4 us and
// slowed every kernel running beside it -- measured, round 3.)  Data nobody reads between this kernel's start and X is
// all this relies on.  Call from every thread of the workgroup (a barrier inside).
__device__ __forceinline__ void start_wait(const unsigned int *f, unsigned int v, const WaitBound &b) {
    if (!f) return;
    int waited = 0;
    if (threadIdx.x == 0) waited = spin_bounded(f, v, b) ? 1 : 0;
    if (__syncthreads_or(waited)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
struct StampScope {
    unsigned long long *p;
    unsigned int always;            // the first `always` workgroups all report their end (the long-key role of the embedding update)
    // start: workgroup 0 (dispatched first); end: the maximum over every 33rd workgroup (33, not 32: workgroup b runs on XCD b % 8, and the
    // kernels that deal their work XCD by XCD would be sampled on XCD 7 only) and the last one -- one atomic
    // per workgroup on one address made a 7 000-workgroup launch 2.4x slower
    __device__ __forceinline__ explicit StampScope(unsigned long long *q, unsigned int all_below = 0) : p(q), always(all_below) {
        if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] = (unsigned long long)wall_clock64();
    }
    __device__ __forceinline__ ~StampScope() {
        // (a large `always` range is sampled too, every 8th: 2048 long-key workgroups with an atomic each on this one word made the
        //  stamped embedding update 31 us instead of 18 -- the measurement, not the kernel)
        if (p && threadIdx.x == 0 && ((blockIdx.x % 33u) == 32u || blockIdx.x == gridDim.x - 1 || (blockIdx.x < always && (always <= 256u || (blockIdx.x & 7u) == 7u))))
            atomicMax(p + 1, (unsigned long long)wall_clock64());
    }
};
#endif

// ---------------------------------------------------------------------------
// GEMM (kernels_gemm.hip): f32 MFMA 32x32x2, exact f32
// ---------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_SIGMOID = 2, EPI_MASK_POS = 3 };
// C[M][N] = epi( A[M][K] * Bt[N][K]^T )   (both operands K-contiguous)
//   EPI_MASK_POS: C = acc * (mask[row][col] > 0 ? 1 : 0) for col < mask_cols, acc otherwise
int gemm_nt(const float *A, int lda, int a_rows, const float *Bt, int ldb, int b_rows, float *C,
            int ldc, int M, int N, int K, int epi, const float *mask, int ldmask, int mask_cols,
            const int *skip_flag, hipStream_t st, LaunchOpts *lo = nullptr, unsigned int *werr = nullptr);
// Cpart[z][Kout][ldc] = sum over m in split z of A[m][kout] * D[m][n]  (split-K over M)
int gemm_tn_splitk(const float *A, int lda, int a_cols, const float *D, int ldd, int d_cols,
                   float *Cpart, int ldc, int64_t part_stride, int Kout, int N, int M, int nsplit,
                   const int *skip_flag, hipStream_t st, const LaunchOpts *lo = nullptr, unsigned int *werr = nullptr);   // lo: prio, and
                   // wait = a START wait: no workgroup reads its operands before *wait reached wait_val
int gemm_tn_choose_split(int Kout, int N, int M);
// two consecutive relu FcLayer.forward GEMMs in one launch (kernels_gemm.hip k_fc_fwd_pair); _ok: do the shapes fit
int gemm_nt_fwd_pair_ok(int M, int N1, int N2, int K1, int K2);
int gemm_nt_fwd_pair(const float *A, int lda, int a_rows, const float *W1t, int ldb1, int N1, float *Y1, int ldy1, int K1,
                     const float *W2t, int ldb2, int N2, float *Y2, int ldy2, int K2, int M, unsigned int *ctr, unsigned int *epoch,
                     unsigned int *xcc_err, hipStream_t st, LaunchOpts *lo = nullptr, unsigned int *werr = nullptr);
extern int g_fwd_pair;
extern int g_last_rows, g_sort_ablate, g_field_sort, g_ext_events;
extern int g_gemm_nt_cfg, g_gemm_tn_cfg, g_gemm_xcd, g_gemm_ablate, g_gemm_tn_target;
extern int g_seq_ablate, g_emb_short_grid, g_seq_long_grid;
extern int g_gather_nt, g_gather_lds, g_plan_sort;
extern int g_fwd_panel, g_wide_slots;
extern int g_plan_fused, g_shard_sort_defer, g_sort_layer, g_plan_mid, g_seg_fused;
extern int g_rccl_force, g_blk_factor, g_blk_cap, g_push_grouped_max_mb, g_comm_timing;
extern int g_mh_ilp16;   // multi-hot gather: row loads in flight per 16-lane group (D = 64); 0 = default
