// ps_ops.hip -- stand-alone operators and measurement hooks of the C ABI:
// device buffers, EmbeddingLayer.forward / FcLayer.forward as single calls
// (what a JNI GpuEmbeddingLayer / GpuFcLayer binds), and the large-table
// gather benchmark of BASELINE config 4.
#include <string.h>

#include "ps_store.h"

extern "C" int ps_dev_alloc(ps_store_t *s, size_t bytes, void **out_dev) {
    RtGuard rt_guard;
    if (!s || !out_dev) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipMalloc(out_dev, bytes ? bytes : 16));
    HIPCHK(hipMemsetAsync(*out_dev, 0, bytes ? bytes : 16, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}
extern "C" int ps_dev_free(ps_store_t *s, void *p) {
    RtGuard rt_guard;
    if (!s) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (p) HIPCHK(hipFree(p));
    return PS_OK;
}
extern "C" int ps_dev_upload(ps_store_t *s, void *dst, const void *src, size_t bytes) {
    if (!s || !dst || !src) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}
extern "C" int ps_dev_download(ps_store_t *s, void *dst, const void *src, size_t bytes) {
    if (!s || !dst || !src) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}
extern "C" int ps_store_sync(ps_store_t *s) {
    if (!s) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(s->stream));
    return store_check_bad_ids(s);
}

extern "C" int ps_emb_forward(ps_store_t *s, const int64_t *ids_dev, const int64_t *offsets_dev, int B,
                              int act, float *out_dev, int ld) {
    RoctxRange roctx_range("ps_emb_forward");
    if (!s || !ids_dev || !out_dev || B <= 0) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (!s->emb.W) return ps_set_err(PS_MISSING, "no embedding tables");
    if (ld < s->emb.F * s->emb.D || (s->emb.D % 4 == 0 && (ld & 3))) return ps_set_err(PS_E_BAD_ARG, "bad ld %d", ld);
    PSCHK(store_enter(s));
    EmbFwdArgs e;
    memset(&e, 0, sizeof e);
    e.W = s->emb.W; e.row_base = s->emb.row_base_dev; e.ids = ids_dev; e.offsets = offsets_dev;
    e.B = B; e.F = s->emb.F; e.D = s->emb.D; e.X = 0; e.act = act;
    e.out = out_dev; e.ld = ld; e.err = s->err_dev;
    e.table_bytes = sizeof(float) * (size_t)s->emb.total_rows * s->emb.D;
    return launch_emb_fwd(e, s->stream);
}

extern "C" int ps_fc_forward(ps_store_t *s, int layer, int act, const float *x_dev, int ldx, int B,
                             float *y_dev, int ldy) {
    RoctxRange roctx_range("ps_fc_forward");
    if (!s || !x_dev || !y_dev || B <= 0) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (layer < 0 || layer >= (int)s->fc.size() || !s->fc[layer].present) return ps_set_err(PS_MISSING, "fc%d absent", layer);
    FcParams &p = s->fc[layer];
    if (ldx != p.Kpad) return ps_set_err(PS_E_BAD_ARG, "fc%d wants ldx = %d (in+1 rounded up to 16, ones column at %d)", layer, p.Kpad, p.K);
    if (ldy < p.N) return ps_set_err(PS_E_BAD_ARG, "ldy %d < out %d", ldy, p.N);
    PSCHK(store_enter(s));
    const int epi = act == PS_ACT_RELU ? EPI_RELU : act == PS_ACT_SIGMOID ? EPI_SIGMOID : EPI_NONE;
    return gemm_nt(x_dev, ldx, B, p.Wt, p.Kpad, p.N, y_dev, ldy, B, p.N, p.Kpad, epi, nullptr, 0, 0, nullptr, s->stream);
}

// ---------------------------------------------------------------------------
// BASELINE config 4: one huge table, random gather, HBM roofline
// ---------------------------------------------------------------------------
namespace {
__global__ void k_fill_table(float *W, int64_t n4, uint64_t seed) {
    // cheap device hash fill, float4 per thread (no host copy of a 256 GB table)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const uint64_t h = ps_splitmix64(seed ^ (uint64_t)i);
        float4 v;
        v.x = (float)(uint32_t)(h & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
        v.y = (float)(uint32_t)((h >> 16) & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
        v.z = (float)(uint32_t)((h >> 32) & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
        v.w = (float)(uint32_t)((h >> 48) & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
        reinterpret_cast<float4 *>(W)[i] = v;
    }
}
__global__ void k_rand_ids(int64_t *ids, int64_t n, int64_t rows, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = (int64_t)(ps_splitmix64(seed + (uint64_t)i * 0x9E3779B97F4A7C15ull) % (uint64_t)rows);
}
__global__ void k_iota_offsets(int64_t *off, int64_t nbags, int bag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nbags) off[i] = i * bag;
}
}  // namespace

namespace {
// the synthetic table + lookups of BASELINE configs[3]: everything a pure function of (rows, D, n, bag, seed)
struct GatherRun {
    float *W = nullptr, *out = nullptr;
    int64_t *ids = nullptr, *off = nullptr, *rb = nullptr;
    int *err = nullptr;
    hipStream_t st = nullptr;
    EmbFwdArgs a;
    ~GatherRun() {
        RtGuard rt_guard;
        (void)hipStreamSynchronize(st);
        if (W) (void)hipFree(W); if (out) (void)hipFree(out); if (ids) (void)hipFree(ids);
        if (off) (void)hipFree(off); if (rb) (void)hipFree(rb); if (err) (void)hipFree(err);
    }
    int setup(ps_store *s, int64_t rows, int D, int64_t n, int bag, uint64_t seed) {
        RtGuard rt_guard;
        st = s->stream;
        const int64_t nnz = n * bag;
        const size_t wbytes = sizeof(float) * (size_t)rows * D;
        hipError_t e = hipMalloc((void **)&W, wbytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();      // clear: the caller may retry with a smaller table
            W = nullptr;
            return ps_set_err(PS_E_HIP, "hipMalloc of the %.1f GB table failed: %s", wbytes / 1e9, hipGetErrorString(e));
        }
        HIPCHK(hipMalloc((void **)&out, sizeof(float) * (size_t)n * D));
        HIPCHK(hipMalloc((void **)&ids, sizeof(int64_t) * (size_t)nnz));
        HIPCHK(hipMalloc((void **)&off, sizeof(int64_t) * (size_t)(n + 1)));
        HIPCHK(hipMalloc((void **)&rb, sizeof(int64_t) * 2));
        HIPCHK(hipMalloc((void **)&err, sizeof(int)));
        HIPCHK(hipMemsetAsync(err, 0, sizeof(int), st));
        const int64_t base[2] = {0, rows};
        HIPCHK(hipMemcpyAsync(rb, base, sizeof base, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));          // `base` is a stack array
        hipLaunchKernelGGL(k_fill_table, dim3(256 * 32), dim3(256), 0, st, W, (int64_t)(rows * (int64_t)D / 4), seed);
        hipLaunchKernelGGL(k_rand_ids, dim3(cdiv(nnz, 256)), dim3(256), 0, st, ids, nnz, rows, seed ^ 0xABCDEFull);
        hipLaunchKernelGGL(k_iota_offsets, dim3(cdiv(n + 1, 256)), dim3(256), 0, st, off, n, bag);
        HIPCHK(hipGetLastError());
        memset(&a, 0, sizeof a);
        a.W = W; a.row_base = rb; a.ids = ids; a.offsets = bag > 1 ? off : nullptr;
        a.B = (int)n; a.F = 1; a.D = D; a.X = 0; a.act = PS_ACT_RELU; a.out = out; a.ld = D; a.err = err;
        a.table_bytes = wbytes;
        return PS_OK;
    }
};
}  // namespace

extern "C" int ps_bench_gather(ps_store_t *s, int64_t rows, int D, int64_t n, int bag, int iters, uint64_t seed,
                               double *avg_ms_out, double *bytes_read_out, double *bytes_written_out) {
    if (!s || rows <= 0 || D <= 0 || (D & 3) || n <= 0 || n > 0x7fffffff || bag <= 0 || iters <= 0 || !avg_ms_out)
        return ps_set_err(PS_E_BAD_ARG, "bad argument (D must be a multiple of 4)");
    PSCHK(store_enter(s));
    GatherRun g;
    PSCHK(g.setup(s, rows, D, n, bag, seed));
    hipStream_t st = g.st;
    PSCHK(launch_emb_fwd(g.a, st));   // warm-up
    hipEvent_t ea, eb;
    HIPCHK(hipEventCreate(&ea)); HIPCHK(hipEventCreate(&eb));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventRecord(ea, st));
    for (int i = 0; i < iters; ++i) PSCHK(launch_emb_fwd(g.a, st));
    HIPCHK(hipEventRecord(eb, st));
    HIPCHK(hipEventSynchronize(eb));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ea, eb));
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    const int64_t nnz = n * bag;
    *avg_ms_out = (double)ms / iters;
    if (bytes_read_out) *bytes_read_out = (double)nnz * (4.0 * D + 8.0) + (bag > 1 ? 8.0 * (double)(n + 1) : 0.0);
    if (bytes_written_out) *bytes_written_out = 4.0 * (double)n * D;
    return PS_OK;
}

extern "C" int ps_bench_gather_check(ps_store_t *s, int64_t rows, int D, int64_t n, int bag, uint64_t seed,
                                     int64_t n_sample, int64_t *bag_index_out, int64_t *ids_out, float *out_rows) {
    if (!s || rows <= 0 || D <= 0 || (D & 3) || n <= 0 || n > 0x7fffffff || bag <= 0 || n_sample <= 0 || n_sample > n ||
        !bag_index_out || !ids_out || !out_rows)
        return ps_set_err(PS_E_BAD_ARG, "bad argument");
    PSCHK(store_enter(s));
    GatherRun g;
    PSCHK(g.setup(s, rows, D, n, bag, seed));
    hipStream_t st = g.st;
    PSCHK(launch_emb_fwd(g.a, st));
    const int64_t stride = n / n_sample;
    for (int64_t i = 0; i < n_sample; ++i) {
        const int64_t b = i * stride + (i * 7919) % stride;       // spread over the launch, not only tile starts
        bag_index_out[i] = b;
        HIPCHK(hipMemcpyAsync(ids_out + i * bag, g.ids + b * bag, sizeof(int64_t) * (size_t)bag, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(out_rows + i * D, g.out + b * D, sizeof(float) * (size_t)D, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, g.err, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (err) return ps_set_err(PS_E_STATE, "%d generated ids were out of range", err);
    return PS_OK;
}

// ---------------------------------------------------------------------------
// tuning knobs and the GEMM micro-benchmark (measurement only)
// ---------------------------------------------------------------------------
// Host wait for one stream (NULL: the store's) -- for hosts that plug their own collectives into ps_comm_ops_t
// and stage through the host: the callbacks' `stream` argument is the stream their inputs were produced on.
extern "C" int ps_stream_sync(ps_store_t *s, void *hip_stream) {
    if (!s) return ps_set_err(PS_E_BAD_ARG, "store is NULL");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : s->stream));
    return PS_OK;
}

// How this store's models join their streams right now: 1 = device-side flags (bounded waits), 0 = events; why: the reason
// for the event form ("" when flags are in use).
extern const char *g_dev_wait_off_reason;
extern "C" int ps_store_join_mode(const ps_store_t *s, char *why, int why_cap) {
    if (!s) return -1;
    const char *w = "";
    const bool ok = dev_waits_ok(s);
    if (!ok) {
        if (s->dev_wait_off) w = "a device-side wait timed out on this store";
        else if (!g_dev_wait) w = g_dev_wait_off_reason ? g_dev_wait_off_reason : "dev_wait = 0";
        else w = "more than one live model on this device in this process";
    }
    if (why && why_cap > 0) snprintf(why, (size_t)why_cap, "%s", w);
    return ok ? 1 : 0;
}
extern "C" int64_t ps_store_wait_timeouts(const ps_store_t *s) { return s ? s->wait_timeouts : -1; }

// test hook: a bounded wait on a flag that nobody raises, on the store's stream, then the host-side check -- must come
// back with PS_E_STATE after the timeout, not hang (tests/test_gpu_schedule.py)
extern "C" int ps_dbg_stuck_wait(ps_store_t *s, int in_gemm) {
    if (!s) return ps_set_err(PS_E_BAD_ARG, "store is NULL");
    PSCHK(store_enter(s));
    unsigned int *flag = nullptr;
    float *buf = nullptr;
    { RtGuard g; HIPCHK(hipMalloc((void **)&flag, 64)); HIPCHK(hipMalloc((void **)&buf, sizeof(float) * 3 * 64 * 64)); }
    HIPCHK(hipMemsetAsync(flag, 0, 64, s->stream));
    HIPCHK(hipMemsetAsync(buf, 0, sizeof(float) * 3 * 64 * 64, s->stream));
    int rc = PS_OK;
    if (in_gemm) {          // the end wait of a GEMM launch (kernels_gemm.hip EndWait)
        LaunchOpts lo;
        lo.wait = flag; lo.wait_val = 1;
        rc = gemm_nt(buf, 64, 64, buf + 64 * 64, 64, 64, buf + 2 * 64 * 64, 64, 64, 64, 64, EPI_NONE, nullptr, 0, 0, nullptr, s->stream, &lo, s->werr());
    } else rc = launch_spin_until(flag, 1, s->stream, s->werr(), 99);
    if (rc == PS_OK) rc = store_check_bad_ids(s);
    (void)hipStreamSynchronize(s->stream);
    { RtGuard g; (void)hipFree(flag); (void)hipFree(buf); }
    return rc;
}

extern "C" int ps_tune_set(const char *knob, int value) {
    if (!knob) return ps_set_err(PS_E_BAD_ARG, "null knob");
    if (strcmp(knob, "gemm_nt_cfg") == 0) { g_gemm_nt_cfg = value; return PS_OK; }
    if (strcmp(knob, "gemm_tn_cfg") == 0) { g_gemm_tn_cfg = value; return PS_OK; }
    if (strcmp(knob, "gemm_xcd") == 0) { g_gemm_xcd = value; return PS_OK; }
    if (strcmp(knob, "gemm_tn_target") == 0) { g_gemm_tn_target = value; return PS_OK; }
    if (strcmp(knob, "last_rows") == 0) { g_last_rows = value; return PS_OK; }
    if (strcmp(knob, "sort_ablate") == 0) { g_sort_ablate = value; return PS_OK; }
    if (strcmp(knob, "field_sort") == 0) { g_field_sort = value; return PS_OK; }
    if (strcmp(knob, "ext_events") == 0) { g_ext_events = value; return PS_OK; }
    if (strcmp(knob, "dev_wait") == 0) { g_dev_wait = value; return PS_OK; }
    if (strcmp(knob, "spin_timeout_ms") == 0) { g_spin_timeout_ticks = value > 0 ? (unsigned long long)value * 100000ull : 0ull; return PS_OK; }
    if (strcmp(knob, "tail_dev") == 0) { g_tail_dev = value; return PS_OK; }
    if (strcmp(knob, "tail_fused") == 0) { g_tail_fused = value; return PS_OK; }
    if (strcmp(knob, "tn_start_wait") == 0) { g_tn_start_wait = value; return PS_OK; }
    if (strcmp(knob, "dw_split") == 0) { g_dw_split = value; return PS_OK; }
    if (strcmp(knob, "fwd_pair") == 0) { g_fwd_pair = value; return PS_OK; }
    if (strcmp(knob, "tail_defer") == 0) { g_tail_defer = value; return PS_OK; }
    if (strcmp(knob, "end_wait") == 0) { g_end_wait = value; return PS_OK; }
    if (strcmp(knob, "main_prio") == 0) { g_main_prio = value; return PS_OK; }
    if (strcmp(knob, "gemm_pipe") == 0) { g_gemm_pipe = value; return PS_OK; }
    if (strcmp(knob, "gemm_ks") == 0) { g_gemm_ks = value; return PS_OK; }
    if (strcmp(knob, "tn_prio") == 0) { g_tn_prio = value; return PS_OK; }
    if (strcmp(knob, "dw_late") == 0) { g_dw_late = value; return PS_OK; }
    if (strcmp(knob, "keys_early") == 0) { g_keys_early = value; return PS_OK; }
    if (strcmp(knob, "sort_late") == 0) { g_sort_late = value; return PS_OK; }
    if (strcmp(knob, "plan_early") == 0) { g_plan_early = value; return PS_OK; }
    if (strcmp(knob, "plan_mid") == 0) { g_plan_mid = value; return PS_OK; }
    if (strcmp(knob, "seg_fused") == 0) { g_seg_fused = value; return PS_OK; }
    if (strcmp(knob, "mh_presort") == 0) { g_mh_presort = value; return PS_OK; }
    if (strcmp(knob, "emb_list_min") == 0) { g_emb_list_min = value; return PS_OK; }
    if (strcmp(knob, "emb_list_grid") == 0) { g_emb_list_grid = value; return PS_OK; }
    if (strcmp(knob, "wide_in_gather") == 0) { g_wide_in_gather = value; return PS_OK; }
    if (strcmp(knob, "mh_prio") == 0) { g_mh_prio = value; return PS_OK; }
    if (strcmp(knob, "keys_grid") == 0) { g_keys_grid = value; return PS_OK; }
    if (strcmp(knob, "emb_xcd") == 0) { g_emb_xcd = value; return PS_OK; }
    if (strcmp(knob, "emb_lxcd") == 0) { g_emb_lxcd = value; return PS_OK; }
    if (strcmp(knob, "super_in_update") == 0) { g_super_in_update = value; return PS_OK; }
    if (strcmp(knob, "mapped_peer") == 0) { g_mapped_peer = value; return PS_OK; }
    if (strcmp(knob, "mapped_ablate") == 0) { g_mapped_ablate = value; return PS_OK; }
    if (strcmp(knob, "mapped_fuse") == 0) { g_mapped_fuse = value; return PS_OK; }
    if (strcmp(knob, "mapped_lists") == 0) { g_mapped_lists = value; return PS_OK; }
    if (strcmp(knob, "super_list") == 0) { g_super_list = value; return PS_OK; }
    if (strcmp(knob, "fwd_order") == 0) { g_fwd_order = value; return PS_OK; }
    if (strcmp(knob, "slots_in_gather") == 0) { g_slots_in_gather = value; return PS_OK; }
    if (strcmp(knob, "shard_overlap") == 0) { g_shard_overlap = value; return PS_OK; }
    if (strcmp(knob, "radix_scan_free") == 0) { g_radix_scan_free = value; return PS_OK; }
    if (strcmp(knob, "gemm_8w") == 0) { g_gemm_8w = value; return PS_OK; }
    if (strcmp(knob, "radix11") == 0) { g_radix11 = value; return PS_OK; }
    if (strcmp(knob, "stamps") == 0) return stamps_enable(value);
    if (strcmp(knob, "gemm_ablate") == 0) { g_gemm_ablate = value; return PS_OK; }
    if (strcmp(knob, "mh_ilp16") == 0) { g_mh_ilp16 = value; return PS_OK; }
    if (strcmp(knob, "seq_ablate") == 0) { g_seq_ablate = value; return PS_OK; }
    if (strcmp(knob, "seq_long_grid") == 0) { g_seq_long_grid = value; return PS_OK; }
    if (strcmp(knob, "emb_short_grid") == 0) { g_emb_short_grid = value > 0 ? value : 2048; return PS_OK; }
    if (strcmp(knob, "gather_nt") == 0) { g_gather_nt = value; return PS_OK; }
    if (strcmp(knob, "gather_lds") == 0) {
#if defined(PS_GEMM_LAB) && PS_GEMM_LAB
        g_gather_lds = value; return PS_OK;
#else
        if (value) return ps_set_err(PS_E_UNSUPPORTED, "the LDS-staged gather is a rejected variant: it lives in the lab build (tools/gemm_lab_build.sh)");
        return PS_OK;
#endif
    }
    if (strcmp(knob, "plan_sort") == 0) { g_plan_sort = value; return PS_OK; }
    if (strcmp(knob, "plan_fused") == 0) { g_plan_fused = value; return PS_OK; }
    if (strcmp(knob, "mh_seg_sort") == 0) { g_mh_seg_sort = value; return PS_OK; }
    if (strcmp(knob, "sort_layer") == 0) { g_sort_layer = value; return PS_OK; }
    if (strcmp(knob, "fwd_panel") == 0) { g_fwd_panel = value; return PS_OK; }
    if (strcmp(knob, "wide_slots") == 0) { g_wide_slots = value; return PS_OK; }
    if (strcmp(knob, "shard_sort_defer") == 0) { g_shard_sort_defer = value; return PS_OK; }
    if (strcmp(knob, "rccl_force") == 0) { g_rccl_force = value; return PS_OK; }
    if (strcmp(knob, "comm_timing") == 0) { g_comm_timing = value; return PS_OK; }
    if (strcmp(knob, "blk_factor") == 0) { g_blk_factor = value; return PS_OK; }
    if (strcmp(knob, "blk_cap") == 0) { g_blk_cap = value; return PS_OK; }
    if (strcmp(knob, "push_grouped_max_mb") == 0) { g_push_grouped_max_mb = value; return PS_OK; }
    return ps_set_err(PS_E_BAD_ARG, "unknown knob %s", knob);
}

extern "C" int ps_bench_gemm(ps_store_t *s, int kind, int M, int N, int K, int nsplit, int iters, double *avg_ms_out) {
    if (!s || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || !avg_ms_out) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    const int Kp = (int)round_up(K, 16), Np = (int)round_up(N, 16);
    float *A = nullptr, *B = nullptr, *Cc = nullptr;
    const size_t ea = (size_t)M * (kind == 0 ? Kp : (size_t)round_up(K, 16));
    HIPCHK(hipMalloc((void **)&A, sizeof(float) * (size_t)M * Kp));
    HIPCHK(hipMalloc((void **)&B, sizeof(float) * (size_t)(kind == 0 ? (size_t)N * Kp : (size_t)M * Np)));
    HIPCHK(hipMalloc((void **)&Cc, sizeof(float) * (kind == 0 ? (size_t)M * Np : (size_t)nsplit * Kp * Np)));
    (void)ea;
    PSCHK(launch_fill(A, (int64_t)M * Kp, 0.5f, st));
    PSCHK(launch_fill(B, kind == 0 ? (int64_t)N * Kp : (int64_t)M * Np, 0.25f, st));
    // PS_GEMM_GATHER_EMU=<rows> (with a PS_GEMM_ABLATE & 512 build): a table of that many 64-byte rows for the emulated gather
    float *emu = nullptr;
    int emu_rows = 0;
    if (const char *e = getenv("PS_GEMM_GATHER_EMU")) {
        emu_rows = atoi(e);
        if (emu_rows > 0) { HIPCHK(hipMalloc((void **)&emu, sizeof(float) * 16 * (size_t)emu_rows)); PSCHK(launch_fill(emu, (int64_t)emu_rows * 16, 0.5f, st)); }
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    int rc = PS_OK;
    for (int it = -2; it < iters && rc == PS_OK; ++it) {
        if (it == 0) HIPCHK(hipEventRecord(e0, st));
        if (kind == 0) rc = gemm_nt(A, Kp, M, B, Kp, N, Cc, Np, M, N, Kp, EPI_RELU, emu, emu_rows, 0, nullptr, st);
        else rc = gemm_tn_splitk(A, Kp, Kp, B, Np, Np, Cc, Np, (int64_t)Kp * Np, K, N, M, nsplit, nullptr, st);
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms_out = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(Cc);
    if (emu) (void)hipFree(emu);
    return rc;
}


// ---------------------------------------------------------------------------
// GPU-side time stamps (ps_common.h): STAMP_SLOTS launches, then stamp_next returns nullptr
// ---------------------------------------------------------------------------
namespace {
constexpr int STAMP_SLOTS = 8192;
unsigned long long *g_stamp_buf = nullptr;      // [STAMP_SLOTS][2]
std::vector<const char *> g_stamp_names;
bool g_stamps_on = false;
}
int stamps_enable(int on) {
    g_stamps_on = false;
    g_stamp_names.clear();
    if (!on) return PS_OK;
    // (a stamped kernel still in flight on some store's stream would race with the re-initialisation below -- every enable gets a
    //  buffer of its own; the previous one is left to whatever still writes to it: 128 KB per enable, a measurement facility)
    g_stamp_buf = nullptr;
    HIPCHK(hipMalloc((void **)&g_stamp_buf, sizeof(unsigned long long) * 2 * STAMP_SLOTS));
    std::vector<unsigned long long> init(2 * STAMP_SLOTS);
    for (int i = 0; i < STAMP_SLOTS; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
    // (on a stream of its own, never the null stream -- ps_store.h: the default stream's hardware queue slows every multi-stream step
    //  enqueued after it, i.e. exactly the steps about to be stamped.  The caller has waited for its stores' streams.)
    static hipStream_t stamp_stream = nullptr;
    if (!stamp_stream) HIPCHK(hipStreamCreateWithFlags(&stamp_stream, hipStreamNonBlocking));
    HIPCHK(hipMemcpyAsync(g_stamp_buf, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice, stamp_stream));
    HIPCHK(hipStreamSynchronize(stamp_stream));
    g_stamps_on = true;
    return PS_OK;
}
unsigned long long *stamp_next(const char *name) {
    if (!g_stamps_on || (int)g_stamp_names.size() >= STAMP_SLOTS) return nullptr;
    g_stamp_names.push_back(name);
    return g_stamp_buf + 2 * (g_stamp_names.size() - 1);
}
// names: '\n'-separated, vals: [n][2]; returns the number of stamped launches (after a device synchronize)
extern "C" int ps_dbg_stamps(char *names, int names_cap, unsigned long long *vals, int vals_cap) {
    if (!g_stamp_buf) return 0;
    // (the measurement is over when this is called: a device-wide wait and a synchronous copy may touch the null stream here)
    if (hipDeviceSynchronize() != hipSuccess) return -1;                                                                              // null-stream-ok
    const int n = (int)std::min<size_t>(g_stamp_names.size(), (size_t)vals_cap);
    if (hipMemcpy(vals, g_stamp_buf, sizeof(unsigned long long) * 2 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;     // null-stream-ok
    std::string s;
    for (int i = 0; i < n; ++i) { s += g_stamp_names[i]; s += '\n'; }
    snprintf(names, (size_t)names_cap, "%s", s.c_str());
    return n;
}
