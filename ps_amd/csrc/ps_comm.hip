// ps_comm.hip -- the parameter-server exchange driven from inside the library: ONE C call per step.
//
// ps_amd/sharded.py drives the same device-side halves through torch.distributed; measured on MI355X that
// wire is host-bound (five Python-level collectives, two host round trips for split sizes, ~45 ctypes
// launches: ~0.5 ms per step for ~0.3 ms of GPU work).  Here the whole step of net/PSRouterClient.java:60-151
// + net/PServer.java:102-283 is enqueued by ps_shard_step with ONE host wait (the N x N matrix of key counts):
//
//   plan (sort/unique by owner)            ps_shard_plan_launch
//   all-gather of the per-owner counts     -> every rank learns what it sends AND what it receives
//   all-to-all-v row ids                   PSRouterClient.getList fan-out
//   owner gather + all-to-all-v rows back  PServer.getList -> worker cache
//   forward / backward on the cache        Model.train
//   all-reduce of [fc | wide G | wide C | wide.bias]
//   all-to-all-v per-key gradients         PSClient.push
//   owner: mean over pushing workers (or async) + updater; replicated tensors: identical update
//
// The collectives go through a small table of callbacks (ps_comm_ops_t): the product implementation is RCCL
// (ncclSend/ncclRecv groups, ncclAllGather, ncclAllReduce over xGMI), loaded with dlopen so that single-GPU
// use never maps the 570 MB library; tests plug in an in-process implementation to run N ranks on one GPU.
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include "ps_store.h"

// ---------------------------------------------------------------------------
// RCCL, bound at run time (the soname torch also ships: one copy per process)
// ---------------------------------------------------------------------------
namespace {

typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
enum { RCCL_CHAR = 0, RCCL_FLOAT = 7, RCCL_SUM = 0 };     // ncclInt8 / ncclFloat32 / ncclSum (rccl.h)

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(rcclUniqueId *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} g_rccl;

int rccl_load() {
    if (g_rccl.h) return PS_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return ps_set_err(PS_MISSING, "librccl.so.1 not found: %s", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return ps_set_err(PS_MISSING, "RCCL symbol %s missing", name); } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return PS_OK;
}

#define RCCLCHK(x) do { int r__ = (x); if (r__ != 0) return ps_set_err(PS_E_HIP, "%s -> %s", #x, g_rccl.GetErrorString(r__)); } while (0)

// `side` carries the counts all-gather of the NEXT step's prefetch (its own communicator: operations on one
// communicator are ordered, so sharing it would chain the running step behind the prefetch)
struct RcclCtx { rcclComm_t comm = nullptr, side = nullptr; int nranks = 1, rank = 0; bool use_side = false; };

int rccl_all_gather(void *ctx, const void *send, void *recv, size_t bytes, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    hipStream_t st = (hipStream_t)stream;
    if (c->nranks == 1) {
        HIPCHK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st));
        return PS_OK;
    }
    RCCLCHK(g_rccl.AllGather(send, recv, bytes, RCCL_CHAR, (c->use_side && c->side) ? c->side : c->comm, st));
    return PS_OK;
}

int rccl_all_to_all_v(void *ctx, const void *send, const int64_t *sc, void *recv, const int64_t *rc, size_t eb, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    hipStream_t st = (hipStream_t)stream;
    size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
    const bool wire = c->nranks > 1;
    if (wire) RCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (p == c->rank) { self_so = so; self_ro = ro; }
        else {
            if (sc[p] > 0) RCCLCHK(g_rccl.Send((const char *)send + so * eb, (size_t)sc[p] * eb, RCCL_CHAR, p, c->comm, st));
            if (rc[p] > 0) RCCLCHK(g_rccl.Recv((char *)recv + ro * eb, (size_t)rc[p] * eb, RCCL_CHAR, p, c->comm, st));
        }
        so += (size_t)sc[p]; ro += (size_t)rc[p];
    }
    if (wire) RCCLCHK(g_rccl.GroupEnd());
    if (sc[c->rank] != rc[c->rank]) return ps_set_err(PS_E_STATE, "all-to-all-v: self counts differ");
    if (sc[c->rank] > 0)       // this rank's own keys never touch the wire
        HIPCHK(hipMemcpyAsync((char *)recv + self_ro * eb, (const char *)send + self_so * eb, (size_t)sc[c->rank] * eb, hipMemcpyDeviceToDevice, st));
    return PS_OK;
}

int rccl_all_reduce(void *ctx, float *buf, int64_t n, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    if (c->nranks == 1 || n <= 0) return PS_OK;
    // on the MAIN communicator: ps_shard_step_finish issues it on the main stream behind the gradient all-to-all-v, so
    // every rank enqueues the collectives of one communicator in one order on one stream.  (Two communicators driven
    // from two streams at once is a documented NCCL/RCCL deadlock hazard: device launch order across ranks is then not
    // deterministic.  The side communicator carries only the counts all-gather of an explicitly requested prefetch.)
    RCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT, RCCL_SUM, c->comm, (hipStream_t)stream));
    return PS_OK;
}

// The exchange buffers are sized ONCE, for the worst case (every worker requests all of its keys from this owner):
// no hipMalloc / hipFree -- both synchronise the device -- ever runs inside a step or between two collectives.
template <typename T>
int size_once(ps_store *s, T **p, int64_t *cap, int64_t need, size_t elem) {
    if (need <= *cap) return PS_OK;
    RtGuard rt_guard;
    if (*p) { HIPCHK(hipStreamSynchronize(s->stream)); (void)hipFree(*p); *p = nullptr; }
    HIPCHK(hipMalloc((void **)p, elem * (size_t)need));
    s->bytes += (int64_t)(elem * (size_t)need);
    *cap = need;
    return PS_OK;
}

}  // namespace

extern "C" int ps_comm_rccl_unique_id(char *out256) {
    if (!out256) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(rccl_load());
    for (int k = 0; k < 2; ++k) {          // two ids: the main communicator and the prefetch one
        rcclUniqueId id;
        RCCLCHK(g_rccl.GetUniqueId(&id));
        memcpy(out256 + 128 * k, id.internal, 128);
    }
    return PS_OK;
}

extern "C" int ps_comm_rccl_create(ps_store_t *s, int nranks, int rank, const char *id256, ps_comm_ops_t *out) {
    if (!s || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id256)) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(s->device));
    RcclCtx *c = new RcclCtx();
    c->nranks = nranks; c->rank = rank;
    if (nranks > 1) {
        int rc = rccl_load();
        if (rc != PS_OK) { delete c; return rc; }
        rcclUniqueId id;
        memcpy(id.internal, id256, 128);
        int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
        if (r == 0) {
            memcpy(id.internal, id256 + 128, 128);
            r = g_rccl.CommInitRank(&c->side, nranks, id, rank);
        }
        if (r != 0) {
            if (c->comm) (void)g_rccl.CommDestroy(c->comm);
            delete c;
            return ps_set_err(PS_E_HIP, "ncclCommInitRank -> %s", g_rccl.GetErrorString(r));
        }
    }
    memset(out, 0, sizeof *out);
    out->ctx = c; out->nranks = nranks; out->rank = rank;
    out->all_gather = rccl_all_gather; out->all_to_all_v = rccl_all_to_all_v; out->all_reduce_sum_f32 = rccl_all_reduce;
    return PS_OK;
}

extern "C" int ps_comm_rccl_destroy(ps_comm_ops_t *ops) {
    if (!ops || !ops->ctx) return PS_OK;
    RcclCtx *c = (RcclCtx *)ops->ctx;
    if (c->side) (void)g_rccl.CommDestroy(c->side);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    ops->ctx = nullptr;
    return PS_OK;
}

// ---------------------------------------------------------------------------
// one-shot wire check: every collective of the table once, on known patterns, verified on the host.  bench.py runs it
// before the first timed step, so that the first execution of the RCCL path on real multi-GPU hardware cannot
// silently exchange the wrong bytes (uneven all-to-all-v counts, peer order, all-gather order, float all-reduce).
// ---------------------------------------------------------------------------
extern "C" int ps_comm_selfcheck(ps_store_t *s, const ps_comm_ops_t *comm) {
    if (!s || !comm || !comm->all_gather || !comm->all_to_all_v || !comm->all_reduce_sum_f32) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    const int n = comm->nranks, me = comm->rank;
    if (n < 1 || me < 0 || me >= n) return ps_set_err(PS_E_BAD_ARG, "bad communicator");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    // rank r sends (p + 1 + r % 3) words to peer p: word i = r << 20 | p << 10 | i
    std::vector<int64_t> sc((size_t)n), rc((size_t)n);
    int64_t ns = 0, nr = 0;
    for (int p = 0; p < n; ++p) { sc[p] = p + 1 + me % 3; rc[p] = me + 1 + p % 3; ns += sc[p]; nr += rc[p]; }
    std::vector<uint32_t> send((size_t)ns), recv((size_t)nr, 0xFFFFFFFFu), gath((size_t)n, 0u);
    int64_t o = 0;
    for (int p = 0; p < n; ++p) for (int64_t i = 0; i < sc[p]; ++i) send[(size_t)o++] = ((uint32_t)me << 20) | ((uint32_t)p << 10) | (uint32_t)i;
    const int nf = 257;
    std::vector<float> red((size_t)nf);
    for (int i = 0; i < nf; ++i) red[(size_t)i] = (float)(me + 1) + 0.25f * (float)i;      // exact in f32 for every rank count in use
    uint32_t *dsend = nullptr, *drecv = nullptr, *dg_in = nullptr, *dg_out = nullptr;
    float *dred = nullptr;
    auto fr = [&]() { (void)hipStreamSynchronize(st); (void)hipFree(dsend); (void)hipFree(drecv); (void)hipFree(dg_in); (void)hipFree(dg_out); (void)hipFree(dred); };
    {
        RtGuard rt_guard;
        HIPCHK(hipMalloc((void **)&dsend, 4 * (size_t)ns)); HIPCHK(hipMalloc((void **)&drecv, 4 * (size_t)nr));
        HIPCHK(hipMalloc((void **)&dg_in, 4)); HIPCHK(hipMalloc((void **)&dg_out, 4 * (size_t)n)); HIPCHK(hipMalloc((void **)&dred, 4 * (size_t)nf));
    }
    const uint32_t tag = 0xC0DE0000u + (uint32_t)me;
    int rcode = PS_OK;
    do {
        if (hipMemcpyAsync(dsend, send.data(), 4 * (size_t)ns, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemsetAsync(drecv, 0xFF, 4 * (size_t)nr, st) != hipSuccess ||
            hipMemcpyAsync(dg_in, &tag, 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(dred, red.data(), 4 * (size_t)nf, hipMemcpyHostToDevice, st) != hipSuccess) { rcode = ps_set_err(PS_E_HIP, "selfcheck: upload failed"); break; }
        if ((rcode = comm->all_gather(comm->ctx, dg_in, dg_out, 4, st)) != PS_OK) break;
        if ((rcode = comm->all_to_all_v(comm->ctx, dsend, sc.data(), drecv, rc.data(), 4, st)) != PS_OK) break;
        if ((rcode = comm->all_reduce_sum_f32(comm->ctx, dred, nf, st)) != PS_OK) break;
        if (hipMemcpyAsync(recv.data(), drecv, 4 * (size_t)nr, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(gath.data(), dg_out, 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(red.data(), dred, 4 * (size_t)nf, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rcode = ps_set_err(PS_E_HIP, "selfcheck: readback failed"); break; }
        for (int p = 0; p < n && rcode == PS_OK; ++p)
            if (gath[(size_t)p] != 0xC0DE0000u + (uint32_t)p) rcode = ps_set_err(PS_E_STATE, "selfcheck: all-gather slot %d holds %08x", p, gath[(size_t)p]);
        o = 0;
        for (int p = 0; p < n && rcode == PS_OK; ++p)
            for (int64_t i = 0; i < rc[p] && rcode == PS_OK; ++i, ++o) {
                const uint32_t want = ((uint32_t)p << 20) | ((uint32_t)me << 10) | (uint32_t)i;
                if (recv[(size_t)o] != want) rcode = ps_set_err(PS_E_STATE, "selfcheck: all-to-all-v word %lld from rank %d is %08x, expected %08x", (long long)i, p, recv[(size_t)o], want);
            }
        for (int i = 0; i < nf && rcode == PS_OK; ++i) {
            const float want = (float)n * (float)(n + 1) * 0.5f + 0.25f * (float)i * (float)n;
            if (n > 1 && red[(size_t)i] != want) rcode = ps_set_err(PS_E_STATE, "selfcheck: all-reduce element %d is %g, expected %g", i, red[(size_t)i], want);
        }
    } while (0);
    { RtGuard rt_guard; fr(); }
    return rcode;
}

// ---------------------------------------------------------------------------
// one BSP (or async) step of worker + owner
// ---------------------------------------------------------------------------
// Measured on MI355X at N = 1: beginning step t+1 on the prefetch stream while step t trains is SLOWER (0.40 vs
// 0.33 ms per step) -- the plan is a chain of ~15 tiny kernels, and interleaving them with the training chain
// delays every launch of the critical path more than the overlap saves; bench.py therefore runs the halves back
// to back (--overlap 0).  The split stays: it is what a host with longer steps (bigger batches) would use.
namespace {
__global__ void k_publish_counts(const uint32_t *__restrict__ src, uint32_t *dst_host, int n, uint32_t *flag_host, uint32_t epoch,
                                 unsigned int *started, unsigned int started_val, unsigned long long *ts) {
    StampScope stamp(ts);
    // "everything in front of this launch on its stream has finished" for a device-side waiter (an early plan's second half)
    if (started && threadIdx.x == 0) __hip_atomic_store(started, started_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst_host[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

// begin: everything of a step that reads no weight -- plan, per-owner counts, the all-gather of the counts and
// their copy to pinned memory -- enqueued without a host wait.  use_side != 0 runs it on the store's prefetch
// stream (and the prefetch communicator) so it can run beside the previous step's training: call begin for
// step t+1 (on ANOTHER model of the same store: its own key lists and activations) before finish of step t.
extern "C" int ps_shard_step_begin(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int use_side) {
    RoctxRange roctx_range("ps_shard_step_begin");
    if (!m || !batch || !comm || !comm->all_gather || !comm->all_to_all_v || !comm->all_reduce_sum_f32)
        return ps_set_err(PS_E_BAD_ARG, "bad argument");
    ps_store *s = m->s;
    const int nsh = comm->nranks, rank = comm->rank;
    if (nsh < 1 || rank < 0 || rank >= nsh) return ps_set_err(PS_E_BAD_ARG, "bad communicator");
    HIPCHK(hipSetDevice(s->device));
    if (use_side && !s->prefetch_stream) {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&s->prefetch_stream, hipStreamNonBlocking, hi));
    }
    hipStream_t st = use_side ? s->prefetch_stream : s->stream;
    ps_model::Shard &sh = m->sh;
    if (!sh.x_ev) {
        HIPCHK(hipEventCreateWithFlags(&sh.x_ev, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sh.done_ev, hipEventDisableTiming));
    }
    // this model's previous step still reads its key lists until its finish has run on the training stream
    if (use_side && sh.done_recorded) HIPCHK(hipStreamWaitEvent(st, sh.done_ev, 0));
    {   // worst case: all nnz_cap keys of every worker live on this owner
        const int64_t rmax = m->nnz_cap * (int64_t)nsh, D = m->cfg.D;
        PSCHK(size_once(s, &sh.x_recv_rows, &sh.x_recv_cap, rmax, sizeof(uint32_t)));
        PSCHK(size_once(s, &sh.x_rows_out, &sh.x_rows_cap, rmax * D, sizeof(float)));
        PSCHK(size_once(s, &sh.x_recv_grads, &sh.x_grads_cap, rmax * D, sizeof(float)));
        PSCHK(size_once(s, &sh.x_cache, &sh.x_cache_cap, m->nnz_cap * D, sizeof(float)));
        PSCHK(shard_push_reserve(s, nsh));
    }
    if (comm->ctx && comm->all_gather == rccl_all_gather) ((RcclCtx *)comm->ctx)->use_side = use_side != 0;
    PSCHK(shard_plan_enqueue(m, batch, nsh, st, false, !use_side));
    const size_t row = sizeof(uint32_t) * (size_t)(nsh + 1);        // every rank's owner_start[0..nranks]: the host takes the differences
    if (!sh.matrix_dev) {
        HIPCHK(hipMalloc((void **)&sh.matrix_dev, row * (size_t)nsh));
        HIPCHK(hipHostMalloc((void **)&sh.matrix_host, row * (size_t)nsh + 64, hipHostMallocDefault));      // + the epoch word
        memset(sh.matrix_host, 0, row * (size_t)nsh + 64);
    }
    PSCHK(comm->all_gather(comm->ctx, sh.owner_start, sh.matrix_dev, row, st));
    // The counts go to the host by a KERNEL that writes them into the pinned (host-coherent) matrix and then raises an
    // epoch word beside it; the host spins on that word.  (hipMemcpyAsync + hipEventRecord + hipEventSynchronize cost a
    // copy packet and a record packet on the stream, and the host woke up 20-40 us late in some processes: a bimodal
    // sharded step, 0.217 / 0.24 ms.)
    if (++sh.x_epoch == 0) ++sh.x_epoch;
    unsigned int *started = nullptr;
    if (sh.tail_due) {          // an early plan: its second half waits for this launch's start on side chain 0
        if (++m->start_epoch == 0) ++m->start_epoch;
        sh.pub_epoch = m->start_epoch;
        started = m->start_flag + 6;
    }
    hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(256), 0, st, sh.matrix_dev, sh.matrix_host, (int)(nsh * (nsh + 1)),
                       sh.matrix_host + (size_t)nsh * (nsh + 1), sh.x_epoch, started, sh.pub_epoch, stamp_next("publish_counts"));
    HIPCHK(hipGetLastError());
    PSCHK(shard_plan_enqueue_tail(m, nsh, st));
    if (use_side) HIPCHK(hipEventRecord(sh.x_ev, st));       // (the training stream orders itself behind the prefetch stream)
    sh.x_begun = true; sh.x_side = use_side != 0;
    return PS_OK;
}

// finish, with the NEXT step's begin slipped in between "gradients ready" and "push": the plan of batch t+1 and its
// counts all-gather read no weight, so they may run before step t's push and updates -- and when the host then waits
// for those counts (the step's one host wait) the GPU still has the push, the owner update and the replicated update
// of step t queued: the host enqueues the start of step t+1 under them instead of in front of an idle GPU.
// (Measured at N = 1: the host wait + the starved first launches were ~60 us of a 0.285 ms step.)
extern "C" int ps_shard_step_finish_begin(ps_model_t *m, const ps_comm_ops_t *comm, int is_async, const ps_batch_t *next_batch, float *loss) {
    RoctxRange roctx_range("ps_shard_step_finish");
    if (!m || !comm) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    ps_store *s = m->s;
    ps_model::Shard &sh = m->sh;
    if (!sh.x_begun) return ps_set_err(PS_E_STATE, "ps_shard_step_begin first");
    const int nsh = comm->nranks, rank = comm->rank;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    // PS_HOST_TIMING=1 (measurement): every 1000 calls, the host time of this call outside / inside the wait for the counts
    static const bool host_timing = getenv("PS_HOST_TIMING") != nullptr;
    struct HostTimer {
        bool on; double t0, wait = 0;
        static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
        explicit HostTimer(bool o) : on(o), t0(o ? now() : 0) {}
        ~HostTimer() {
            if (!on) return;
            static thread_local double sum_all = 0, sum_wait = 0; static thread_local long calls = 0;
            sum_all += now() - t0; sum_wait += wait;
            if (++calls % 1000 == 0) { fprintf(stderr, "[ps_shard_step] host: %.1f us per step enqueueing, %.1f us waiting for the counts\n", (sum_all - sum_wait) / 1000, sum_wait / 1000); sum_all = sum_wait = 0; }
        }
    } host_timer(host_timing);
    const double wait_t0 = host_timing ? HostTimer::now() : 0;
    {   // the step's one host wait: split sizes of every exchange (spin on the epoch word k_publish_counts raises)
        volatile uint32_t *flag = sh.matrix_host + (size_t)nsh * (nsh + 1);
        int64_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != sh.x_epoch) {
            if (++spins > (1ll << 22)) {                 // ~seconds: the kernel never ran -- surface the stream's error instead of hanging
                HIPCHK(hipStreamSynchronize(sh.x_side ? s->prefetch_stream : st));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != sh.x_epoch) return ps_set_err(PS_E_STATE, "the counts of the exchange never arrived");
                break;
            }
            __builtin_ia32_pause();
        }
    }
    if (host_timing) host_timer.wait = HostTimer::now() - wait_t0;
    if (sh.x_side) HIPCHK(hipStreamWaitEvent(st, sh.x_ev, 0));
    sh.x_begun = false; sh.plan_pending = false;
    const int D = m->cfg.D;
    std::vector<int64_t> sc((size_t)nsh), rc((size_t)nsh);
    int64_t U = 0, nrecv = 0;
    for (int o = 0; o < nsh; ++o) {
        const uint32_t *mine = sh.matrix_host + (size_t)rank * (nsh + 1), *theirs = sh.matrix_host + (size_t)o * (nsh + 1);
        sc[o] = (int64_t)mine[o + 1] - (int64_t)mine[o];              // what I request from / push to owner o
        rc[o] = (int64_t)theirs[rank + 1] - (int64_t)theirs[rank];    // what worker o requests from / pushes to me
        if (sc[o] < 0 || rc[o] < 0) return ps_set_err(PS_E_STATE, "negative key count in the exchange matrix");
        U += sc[o]; nrecv += rc[o];
    }
    sh.U = U;
    if (U > m->nnz_cap || nrecv > sh.x_recv_cap)
        return ps_set_err(PS_E_STATE, "exchange counts (%lld requested, %lld received) exceed the buffers sized at ps_shard_step_begin", (long long)U, (long long)nrecv);
    // getList: ids out, rows back
    PSCHK(comm->all_to_all_v(comm->ctx, sh.send_rows, sc.data(), sh.x_recv_rows, rc.data(), sizeof(uint32_t), st));
    PSCHK(ps_shard_serve_pull(s, sh.x_recv_rows, nrecv, sh.x_rows_out));
    PSCHK(comm->all_to_all_v(comm->ctx, sh.x_rows_out, rc.data(), sh.x_cache, sc.data(), sizeof(float) * (size_t)D, st));
    // train on the cache
    PSCHK(ps_shard_forward_backward(m, sh.x_cache, nullptr));
    // the next step's key lists + counts (same stream, same communicator: every rank issues the same order)
    if (next_batch) PSCHK(ps_shard_step_begin(m, next_batch, comm, 0));
    // push: the per-key gradients to their owners, then the dense + wide reduction -- same communicator, same stream,
    // same order on every rank.  The owner's row update is enqueued between the two: it only needs the all-to-all-v.
    PSCHK(comm->all_to_all_v(comm->ctx, m->grads_out, sc.data(), sh.x_recv_grads, rc.data(), sizeof(float) * (size_t)D, st));
    if (nsh > 1) PSCHK(comm->all_reduce_sum_f32(comm->ctx, sh.flat, sh.flat_elems, st));
    PSCHK(ps_shard_apply_push(s, sh.x_recv_rows, sh.x_recv_grads, nrecv, rc.data(), nsh, is_async));
    PSCHK(ps_shard_apply_flat(m, nsh));
    HIPCHK(hipEventRecord(sh.done_ev, st));
    sh.done_recorded = true;
    if (loss) {
        HIPCHK(hipMemcpyAsync(loss, m->loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return PS_OK;
}

extern "C" int ps_shard_step_finish(ps_model_t *m, const ps_comm_ops_t *comm, int is_async, float *loss) {
    return ps_shard_step_finish_begin(m, comm, is_async, nullptr, loss);
}

extern "C" int ps_shard_step(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int is_async, float *loss) {
    PSCHK(ps_shard_step_begin(m, batch, comm, 0));
    return ps_shard_step_finish(m, comm, is_async, loss);
}
