// ps_comm.hip -- the parameter-server exchange driven from inside the library: ONE C call per step.
//
// ps_amd/sharded.py drives the same device-side halves through torch.distributed; measured on MI355X that
// wire is host-bound (five Python-level collectives, two host round trips for split sizes, ~45 ctypes
// launches: ~0.4 ms per step for ~0.16 ms of GPU work).  Here the whole step of net/PSRouterClient.java:60-151
// + net/PServer.java:102-283 is enqueued by ps_shard_step with ONE host wait (the counts of the two
// weight-dependent exchanges, which arrive with the NEXT step's key lists long before they are needed):
//
//   plan (unique keys by owner)            ps_shard_plan_launch, packed into one fixed-size block per owner
//   all-to-all of the id blocks            PSRouterClient.getList fan-out; no split sizes; carries the counts
//   owner gather + all-to-all-v rows back  PServer.getList -> worker cache
//   forward / backward on the cache        Model.train
//   all-to-all-v per-key gradients         PSClient.push
//   owner: mean over pushing workers (or async) + updater
//   all-reduce of [fc | wide G | wide C | wide.bias]; replicated tensors: identical update
//
// The collectives go through a small table of callbacks (ps_comm_ops_t): the product implementation is RCCL
// (ncclSend/ncclRecv groups, ncclAllGather, ncclAllReduce over xGMI; three communicators, see RcclCtx), loaded
// with dlopen so that single-GPU use never maps the 570 MB library; tests plug in their own (N rank threads in
// one process, N rank processes over gloo) to run N ranks on one GPU.
#include <dlfcn.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>

#include <vector>

#include "ps_store.h"
#include "ps_put.h"

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
    int (*CommCount)(const rcclComm_t, int *) = nullptr;
    int (*CommUserRank)(const rcclComm_t, int *) = nullptr;
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
    SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank");
#undef SYM
    g_rccl.h = h;
    return PS_OK;
}

#define RCCLCHK(x) do { int r__ = (x); if (r__ != 0) return ps_set_err(PS_E_HIP, "%s -> %s", #x, g_rccl.GetErrorString(r__)); } while (0)

// Three communicators.  `comm` carries what sits on the step's critical chain, on the training stream: the rows of the
// pull and the gradients of the push.  `side` carries what does not -- the next step's key lists (they read no weight)
// and the all-reduce of the replicated tensors' gradients -- on a side stream of the model, so that neither queues in
// front of the push (operations on ONE communicator are ordered).  Every rank issues the operations of a communicator in
// one fixed order (per step: side = [key lists of t+1, all-reduce of t], comm = [rows of t, gradients of t]); the step code
// selects the communicator (use_side) before a call.  skip_self: the caller reads this rank's own part where it was
// produced -- the all-to-all-v then moves nothing for it.
// `ar` carries only the all-reduce, on side chain 1 (the id exchange runs on side chain 0 right behind the plan's kernels:
// one communicator per stream, so that no two streams ever drive one communicator).
// wire: the collectives go through RCCL (nranks > 1, or a 1-rank table made under rccl_force).  self_wire (forced
// tables only): this rank's OWN part of an all-to-all-v travels too, as a grouped ncclSend + ncclRecv to itself --
// 1: and is read from the receive buffer like any peer's; 2: travels AND the step reads it in place (skip_self), the
// deployment's data path with the wire running underneath.  calls: RCCL operations issued so far (ps_comm_rccl_info).
struct RcclCtx {
    rcclComm_t comm = nullptr, side = nullptr, ar = nullptr; int nranks = 1, rank = 0; int which = 0; bool skip_self = false;
    bool wire = false; int self_wire = 0; int64_t calls[4] = {0, 0, 0, 0};      // all-gather | send/recv groups | all-reduce | sends + recvs
};
rcclComm_t pick_comm(RcclCtx *c) { return (c->which == 1 && c->side) ? c->side : (c->which == 2 && c->ar) ? c->ar : c->comm; }

int rccl_all_gather(void *ctx, const void *send, void *recv, size_t bytes, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    hipStream_t st = (hipStream_t)stream;
    if (!c->wire) {
        HIPCHK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st));
        return PS_OK;
    }
    RCCLCHK(g_rccl.AllGather(send, recv, bytes, RCCL_CHAR, pick_comm(c), st));
    ++c->calls[0];
    return PS_OK;
}

int rccl_all_to_all_v(void *ctx, const void *send, const int64_t *sc, void *recv, const int64_t *rc, size_t eb, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    hipStream_t st = (hipStream_t)stream;
    size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
    const bool wire = c->wire;
    rcclComm_t cm = pick_comm(c);
    if (sc[c->rank] != rc[c->rank]) return ps_set_err(PS_E_STATE, "all-to-all-v: self counts differ");
    if (wire) RCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < c->nranks; ++p) {
        if (p == c->rank) { self_so = so; self_ro = ro; }
        if (wire && (p != c->rank || c->self_wire)) {
            if (sc[p] > 0) { RCCLCHK(g_rccl.Send((const char *)send + so * eb, (size_t)sc[p] * eb, RCCL_CHAR, p, cm, st)); ++c->calls[3]; }
            if (rc[p] > 0) { RCCLCHK(g_rccl.Recv((char *)recv + ro * eb, (size_t)rc[p] * eb, RCCL_CHAR, p, cm, st)); ++c->calls[3]; }
        }
        so += (size_t)sc[p]; ro += (size_t)rc[p];
    }
    if (wire) { RCCLCHK(g_rccl.GroupEnd()); ++c->calls[1]; }
    if (sc[c->rank] > 0 && !c->skip_self && !(wire && c->self_wire))       // this rank's own keys never touch the wire
        HIPCHK(hipMemcpyAsync((char *)recv + self_ro * eb, (const char *)send + self_so * eb, (size_t)sc[c->rank] * eb, hipMemcpyDeviceToDevice, st));
    return PS_OK;
}

int rccl_all_reduce(void *ctx, float *buf, int64_t n, void *stream) {
    RcclCtx *c = (RcclCtx *)ctx;
    if (!c->wire || n <= 0) return PS_OK;
    RCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT, RCCL_SUM, pick_comm(c), (hipStream_t)stream));
    ++c->calls[2];
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

// which: 0 the main communicator, 1 the key-list one, 2 the all-reduce one (RcclCtx)
void comm_select(const ps_comm_ops_t *comm, int which, bool skip_self) {
    if (comm->ctx && comm->all_to_all_v == rccl_all_to_all_v) { RcclCtx *c = (RcclCtx *)comm->ctx; c->which = which; c->skip_self = skip_self; }
}
bool comm_is_rccl(const ps_comm_ops_t *comm) { return comm->ctx && comm->all_to_all_v == rccl_all_to_all_v; }
// does a collective of this table reach a wire (an N-rank table, or a 1-rank RCCL table made under rccl_force)?
bool comm_wired(const ps_comm_ops_t *comm) { return comm->nranks > 1 || (comm_is_rccl(comm) && ((RcclCtx *)comm->ctx)->wire); }
// is this rank's own part of the rows / gradient exchanges read where it was produced?  RCCL: always (unless the forced
// wire carries it: rccl_force = 1); a plugged-in table: when it says so (PS_COMM_OWN_IN_PLACE)
bool comm_own_in_place(const ps_comm_ops_t *comm) {
    if (comm_is_rccl(comm)) { const RcclCtx *c = (const RcclCtx *)comm->ctx; return !(c->wire && c->self_wire == 1); }
    return (comm->flags & PS_COMM_OWN_IN_PLACE) != 0;
}
}  // namespace
// ps_tune_set("rccl_force", 1 | 2) / PS_RCCL_FORCE: a 1-rank table still goes through RCCL (ps_native.h)
int g_rccl_force = getenv("PS_RCCL_FORCE") ? atoi(getenv("PS_RCCL_FORCE")) : 0;

extern "C" int ps_comm_rccl_unique_id(char *out384) {
    if (!out384) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(rccl_load());
    for (int k = 0; k < 3; ++k) {          // three ids: the main communicator, the key-list one, the all-reduce one
        rcclUniqueId id;
        RCCLCHK(g_rccl.GetUniqueId(&id));
        memcpy(out384 + 128 * k, id.internal, 128);
    }
    return PS_OK;
}

extern "C" int ps_comm_rccl_create(ps_store_t *s, int nranks, int rank, const char *id384, ps_comm_ops_t *out) {
    const char *id256 = id384;
    if (!s || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id256)) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    PSCHK(store_enter(s));
    RcclCtx *c = new RcclCtx();
    c->nranks = nranks; c->rank = rank;
    const int force = nranks == 1 ? g_rccl_force : 0;
    if (nranks > 1 || force) {
        int rc = rccl_load();
        if (rc != PS_OK) { delete c; return rc; }
        char own_ids[384];
        if (!id256) {               // (a forced 1-rank table: nobody to share the ids with)
            rc = ps_comm_rccl_unique_id(own_ids);
            if (rc != PS_OK) { delete c; return rc; }
            id256 = own_ids;
        }
        rcclUniqueId id;
        memcpy(id.internal, id256, 128);
        int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
        if (r == 0) {
            memcpy(id.internal, id256 + 128, 128);
            r = g_rccl.CommInitRank(&c->side, nranks, id, rank);
        }
        if (r == 0) {
            memcpy(id.internal, id256 + 256, 128);
            r = g_rccl.CommInitRank(&c->ar, nranks, id, rank);
        }
        if (r != 0) {
            if (c->side) (void)g_rccl.CommDestroy(c->side);
            if (c->comm) (void)g_rccl.CommDestroy(c->comm);
            delete c;
            return ps_set_err(PS_E_HIP, "ncclCommInitRank -> %s", g_rccl.GetErrorString(r));
        }
        c->wire = true;
        c->self_wire = force;
    }
    memset(out, 0, sizeof *out);
    out->ctx = c; out->nranks = nranks; out->rank = rank;
    out->all_gather = rccl_all_gather; out->all_to_all_v = rccl_all_to_all_v; out->all_reduce_sum_f32 = rccl_all_reduce;
    out->flags = PS_COMM_OWN_IN_PLACE;
    return PS_OK;
}

extern "C" int ps_comm_rccl_destroy(ps_comm_ops_t *ops) {
    if (!ops || !ops->ctx) return PS_OK;
    RcclCtx *c = (RcclCtx *)ops->ctx;
    if (c->ar) (void)g_rccl.CommDestroy(c->ar);
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
    PSCHK(store_enter(s));
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
    const std::vector<float> red0 = red;
    // an RCCL table has three communicators (rows + gradients | key lists | all-reduce): every one of them is checked
    const int ncomm = (comm_is_rccl(comm) && ((RcclCtx *)comm->ctx)->wire) ? 3 : 1;
    for (int which = 0; which < ncomm && rcode == PS_OK; ++which) {
    red = red0;
    comm_select(comm, which, false);
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
            if (red[(size_t)i] != want) rcode = ps_set_err(PS_E_STATE, "selfcheck: all-reduce element %d is %g, expected %g", i, red[(size_t)i], want);
        }
    } while (0);
    comm_select(comm, 0, false);
    if (rcode != PS_OK && ncomm > 1) {
        const std::string msg = ps_last_error();
        rcode = ps_set_err(rcode, "communicator %d of 3: %s", which, msg.c_str());
    }
    }
    { RtGuard rt_guard; fr(); }
    return rcode;
}

// RCCL's own view of a communicator made by ps_comm_rccl_create (bench.py prints it: evidence that the wire the step used
// really spans the ranks): ranks in the main communicator, this rank's index in it, whether the side communicator exists.
extern "C" int ps_comm_rccl_info(const ps_comm_ops_t *ops, int *comm_count, int *user_rank, int *has_side) {
    if (!ops || !ops->ctx || ops->all_to_all_v != rccl_all_to_all_v) return ps_set_err(PS_E_BAD_ARG, "not an RCCL communicator table");
    RcclCtx *c = (RcclCtx *)ops->ctx;
    int n = c->nranks, r = c->rank;
    if (c->comm) { RCCLCHK(g_rccl.CommCount(c->comm, &n)); RCCLCHK(g_rccl.CommUserRank(c->comm, &r)); }
    if (comm_count) *comm_count = n;
    if (user_rank) *user_rank = r;
    if (has_side) *has_side = (c->side ? 1 : 0) + (c->ar ? 1 : 0);
    return PS_OK;
}

// RCCL operations this table has issued so far: out[0] ncclAllGather, [1] ncclGroupStart/End pairs (all-to-all-v), [2]
// ncclAllReduce, [3] ncclSend + ncclRecv calls; out[4] = 1 when the table reaches a wire at all (N > 1, or rccl_force).
extern "C" int ps_comm_rccl_calls(const ps_comm_ops_t *ops, int64_t *out5) {
    if (!ops || !ops->ctx || ops->all_to_all_v != rccl_all_to_all_v || !out5) return ps_set_err(PS_E_BAD_ARG, "not an RCCL communicator table");
    const RcclCtx *c = (const RcclCtx *)ops->ctx;
    for (int i = 0; i < 4; ++i) out5[i] = c->calls[i];
    out5[4] = c->wire ? 1 : 0;
    return PS_OK;
}

// ---------------------------------------------------------------------------
// one BSP (or async) step of worker + owner
// ---------------------------------------------------------------------------
// Round 3.  What one step enqueues, and where (N ranks; `side chain 1` = the model's dW / dense-update stream):
//
//   training stream (communicator `comm`)                     side chain 1 (communicator `side`)
//   -------------------------------------                     ---------------------------------
//   [host: counts of step t, published during step t-1]
//   owner gather of the requested rows (PServer.getList)
//   all-to-all-v rows -> worker cache
//   forward / backward of step t  .........................   dW GEMMs, flat gradient [fc | wide G | wide C | bias]
//        (the plan of step t+1 runs on side chain 0 meanwhile)  pack t+1's key lists into per-owner blocks
//   all-to-all-v per-key gradients (PSClient.push)            all-to-all of the FIXED-SIZE id blocks of step t+1
//   owner: mean over pushing workers (or async) + updater       + publication of their counts to the host
//                                                             all-reduce of the flat gradient, replicated update
//
// The key lists travel as fixed-size blocks [count | rows | padding] (ps_store.h): no split sizes, so no host wait in
// front of that exchange, and it carries the counts of the two exchanges that do need them.  Rounds 1-2 all-gathered an
// N x N count matrix, waited for it on the host and then sent the ids with exact sizes in front of the pull, on the
// critical chain.  On the critical chain now: two collectives (rows, gradients) and three small kernels per step.
// A rank's own keys are never copied: the owner-side gather's output, the gradient buffer and the packed id block are
// read where they are (EmbFwdArgs.W_alt, PushApplyArgs.rows_p / grads_p).
// Without device-side joins (events only, ps_store_join_mode = 0) or with ps_tune_set("shard_overlap", 0) everything is
// enqueued on the training stream with the one communicator, in the order of the left column then the right.
int g_shard_overlap = 1;
int g_comm_timing = 0;      // ps_tune_set("comm_timing", 1): HIP events around every collective of the step (measurement pass)
namespace {
int coll_collect(ps_model *m) {
    ps_model::Shard &sh = m->sh;
    for (auto &e : sh.coll_ev) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { sh.coll_acc[2 * e.kind] += 1; sh.coll_acc[2 * e.kind + 1] += ms; }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    sh.coll_ev.clear();
    return PS_OK;
}
// one collective of the step, bracketed by events on the stream it is enqueued on when comm_timing is set
template <class Fn>
int timed_coll(ps_model *m, int kind, hipStream_t st, Fn &&fn) {
    if (!g_comm_timing) return fn();
    ps_model::Shard::CollEv e; e.kind = kind;
    HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b));
    HIPCHK(hipEventRecord(e.a, st));
    const int rc = fn();
    HIPCHK(hipEventRecord(e.b, st));
    m->sh.coll_ev.push_back(e);
    if (m->sh.coll_ev.size() > 2048) {
        // (events of steps long finished; the newest few may still be in flight: collect all but the last 64)
        std::vector<ps_model::Shard::CollEv> keep(m->sh.coll_ev.end() - 64, m->sh.coll_ev.end());
        m->sh.coll_ev.resize(m->sh.coll_ev.size() - 64);
        coll_collect(m);
        m->sh.coll_ev = keep;
    }
    return rc;
}
}  // namespace
// out[2 k] = calls, out[2 k + 1] = total ms of collective kind k (0 id blocks, 1 rows, 2 gradients, 3 all-reduce) since the
// last call, measured under ps_tune_set("comm_timing", 1); resets the sums.
extern "C" int ps_shard_collective_times(ps_model_t *m, double *out8) {
    if (!m || !out8) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(m->s));
    HIPCHK(hipStreamSynchronize(m->s->stream));
    for (int i = 0; i < 3; ++i) if (m->side[i]) HIPCHK(hipStreamSynchronize(m->side[i]));
    coll_collect(m);
    for (int i = 0; i < 8; ++i) { out8[i] = m->sh.coll_acc[i]; m->sh.coll_acc[i] = 0; }
    return PS_OK;
}

// ---------------------------------------------------------------------------
// MAPPED PEER (round 6): the two exchanges on the step's critical chain -- rows back (PServer.getList's reply,
// net/PServer.java:102-117) and gradients out (PSClient.push, net/PSClient.java:154-174: fire and forget) -- as stores into
// the peers' memory instead of grouped ncclSend / ncclRecv.  xGMI is point to point and every peer's HBM can be mapped
// (hipIpcGetMemHandle / hipIpcOpenMemHandle between the rank processes): ONE launch of this rank's own per exchange copies
// every peer's part of the send buffer into that peer's receive buffer with 16-byte write-through stores (global_store_dwordx4
// sc0 sc1: the bytes leave for the fabric, nothing stays dirty in an XCD's L2), every wave drains its stores (s_waitcnt
// vmcnt(0): the writes are acknowledged), the launch's last workgroup raises this rank's flag in every peer's flag words and
// then waits -- bounded, like every wait here -- until every peer's flag for this exchange is up in its own.  The next launch of
// the stream (the forward's gather, the owner-side push) starts behind that kernel's end, so its start acquires what the peers
// stored.  What RCCL costs here is not bandwidth (0.37 MB per peer) but 16-18 us of launch and handshake per grouped
// send / recv (profiles/r05_rccl_env_sweep.txt); the id blocks and the all-reduce, both off the chain, stay RCCL's.
//   * where a peer's rows land: the worker wrote its first cache slot of this owner's rows into the id block's header
//     (PS_BLK_HDR word 2); where a worker's gradients land: region `worker` of the owner's receive buffer (per_peer rows each).
//   * flags are epochs (the number of exchanges of that kind so far, the same on every rank): never reset, compared as
//     (int)(flag - epoch) >= 0.  A buffer is not overwritten early: a peer stores the rows of step t+1 only after its own push
//     of step t has run, which waited for this rank's gradients of step t, sent behind this rank's last read of the cache.
//   * a wait that runs into its bound (a peer that died, a flag that never arrives) counts itself in the store's error words
//     like every other bounded wait: the next host-side check returns PS_E_STATE, and bench.py's staged watchdog re-executes
//     the ranks on the RCCL stage (ps_amd/sharded.py).
// ps_tune_set("mapped_peer", 1): wanted wherever every rank can (N > 1, D % 4 == 0, the sort-free owner push, IPC works);
// 2: a 1-rank table too, moving its own part through the same launch (bench.py's sharded_n1 `mapped_peer` mode).
// ---------------------------------------------------------------------------
int g_mapped_peer = getenv("PS_MAPPED_PEER") ? atoi(getenv("PS_MAPPED_PEER")) : 0;
int g_mapped_ablate = 0;    // measurement only (results wrong): PeerPutArgs.ablate
int g_mapped_lists = 1;     // ps_tune_set("mapped_lists", 0): the id blocks and the flat reduction stay on the table (RCCL) under mapped_peer
int g_mapped_fuse = 1;      // ps_tune_set("mapped_fuse", 0): the rows exchange as a put launch behind the gather again (first form of round 6)
namespace {
__global__ __launch_bounds__(256) void k_peer_put(PeerPutArgs a) {
    StampScope stamp(a.ts);
    peer_put_body(a, blockIdx.x);
}

int mapped_put(ps_model *m, int kind, const float *src, const int64_t *pre, int rowD, int win, const int64_t *dst_row, bool bcast, bool self, hipStream_t st);

// The set-up's wire check (every rank, collectively, before the first step may trust the path -- it has never run between two
// DEVICES on the development box): three rounds of both kinds of put with a pattern that names (round, sender, receiver, row,
// column), each round verified by a KERNEL of the receiving rank (the next launch on the stream, like the step's consumers: a
// host copy would not read through the caches the step reads through).  A round that changes every word also shows a stale
// line of the round before.
__device__ __forceinline__ uint32_t mp_pattern(uint32_t round, uint32_t from, uint32_t to, uint32_t row, uint32_t col) {
    uint32_t x = round * 0x9E3779B1u ^ (from * 0x85EBCA77u + to * 0xC2B2AE3Du) ^ (row * 0x27D4EB2Fu + col * 0x165667B1u);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x | 1u;
}
__global__ __launch_bounds__(256) void k_mapped_fill(uint32_t *src, int rows_per_peer, int D, int npeers, uint32_t rank, uint32_t round) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)npeers * rows_per_peer * D) return;
    const uint32_t col = (uint32_t)(t % D), row = (uint32_t)((t / D) % rows_per_peer), to = (uint32_t)(t / ((int64_t)D * rows_per_peer));
    src[t] = mp_pattern(round, rank, to, row, col);
}
__global__ __launch_bounds__(256) void k_mapped_check(const uint32_t *recv, int64_t region_rows, int rows_per_peer, int D, int npeers, uint32_t rank, uint32_t round, int skip, unsigned int *bad) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)npeers * rows_per_peer * D) return;
    const uint32_t col = (uint32_t)(t % D), row = (uint32_t)((t / D) % rows_per_peer), from = (uint32_t)(t / ((int64_t)D * rows_per_peer));
    if ((int)from == skip) return;
    if (recv[((size_t)from * region_rows + row) * D + col] != mp_pattern(round, from, rank, row, col)) atomicAdd(bad, 1u);
}

void mapped_close(ps_model *m) {
    ps_model::Shard::Mapped &mp = m->sh.mp;
    typedef ps_model::Shard::Mapped MP;
    for (int p = 0; p < PS_MAX_MAPPED; ++p) {
        if (mp.opened[p])
            for (int w = 0; w < MP::NWIN; ++w) {
                void *q = mp.win[w][p];
                if (!q) continue;
                bool dup = false;                       // (the full-size blocks ARE the wire blocks when a model has no separate ones)
                for (int w2 = 0; w2 < w; ++w2) dup = dup || mp.win[w2][p] == q;
                if (!dup) (void)hipIpcCloseMemHandle(q);
            }
        for (int w = 0; w < MP::NWIN; ++w) mp.win[w][p] = nullptr;
        mp.opened[p] = false;
    }
    mp.on = false;
}

// one record per rank in the set-up's all-gather
struct MappedRec {
    uint32_t ok, pid;
    uint64_t per_peer, flat_rows;
    uint64_t addr[ps_model::Shard::Mapped::NWIN];         // the addresses as this rank sees them (a rank THREAD of the same process uses them as they are)
    hipIpcMemHandle_t h[ps_model::Shard::Mapped::NWIN];
    char pad[640 - 8 - 16 - 8 * ps_model::Shard::Mapped::NWIN - ps_model::Shard::Mapped::NWIN * sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(MappedRec) == 640, "one all-gather slot");

// host-side helper of the set-up: all-gather `bytes` per rank through the table (device staging buffers of the caller)
int mapped_gather(ps_store *s, const ps_comm_ops_t *comm, char *dev, const void *mine, void *all, size_t bytes) {
    const int n = comm->nranks;
    HIPCHK(hipMemcpyAsync(dev, mine, bytes, hipMemcpyHostToDevice, s->stream));
    comm_select(comm, 0, false);
    PSCHK(comm->all_gather(comm->ctx, dev, dev + bytes, bytes, s->stream));
    HIPCHK(hipMemcpyAsync(all, dev + bytes, bytes * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

// Once per model, at its first begin, when EVERY rank asked for it (want_all: the agreement words).  Collective: every rank
// makes the same calls whatever happens to it locally -- a rank that cannot (no IPC, an open that fails) says so in the
// second all-gather and all ranks go on with the table's all-to-all-v.
int mapped_setup(ps_model *m, const ps_comm_ops_t *comm, bool want_all) {
    ps_model::Shard &sh = m->sh;
    ps_model::Shard::Mapped &mp = sh.mp;
    ps_store *s = m->s;
    if (mp.tried) return PS_OK;
    mp.tried = true;
    if (!want_all) return PS_OK;
    const int n = comm->nranks, rank = comm->rank;
    mp.nranks = n; mp.rank = rank;
    mp.per_peer = std::min<int64_t>(m->nnz_cap, std::max<int64_t>(s->emb.total_rows, 1));
    RtGuard rt_guard;
    char *stage = nullptr;
    std::vector<MappedRec> all((size_t)n);
    MappedRec mine;
    memset(&mine, 0, sizeof mine);
    bool ok = true;
    typedef ps_model::Shard::Mapped MP;
    const size_t flag_bytes = sizeof(unsigned int) * (size_t)MP::NKIND * PS_MAX_MAPPED * PS_PUT_WGS;
    // this rank's flag words: fine-grained when the runtime gives that (a peer's store must be seen by a kernel that is running)
    if (hipExtMallocWithFlags((void **)&mp.flags_local, flag_bytes, hipDeviceMallocFinegrained) == hipSuccess) mp.flags_fine = true;
    else { (void)hipGetLastError(); if (hipMalloc((void **)&mp.flags_local, flag_bytes) != hipSuccess) { (void)hipGetLastError(); mp.flags_local = nullptr; ok = false; } }
    if (hipMalloc((void **)&mp.arrive, sizeof(unsigned int) * 4) != hipSuccess) { (void)hipGetLastError(); mp.arrive = nullptr; ok = false; }
    else HIPCHK(hipMemsetAsync(mp.arrive, 0, sizeof(unsigned int) * 4, s->stream));
    // every rank's flat gradient [fc | wide ...] lands here, one slab per (step parity, rank): the sum is then taken HERE, in rank order
    mp.flat_rows = (sh.flat_elems + 3) / 4;
    if (hipMalloc((void **)&mp.flat_recv, sizeof(float) * 4 * (size_t)mp.flat_rows * 2 * (size_t)n) != hipSuccess) { (void)hipGetLastError(); mp.flat_recv = nullptr; ok = false; }
    if (hipMalloc((void **)&stage, sizeof(MappedRec) * (size_t)(n + 1)) != hipSuccess) { (void)hipGetLastError(); return ps_set_err(PS_E_HIP, "mapped peer: staging buffer"); }
    if (mp.flags_local) HIPCHK(hipMemsetAsync(mp.flags_local, 0, flag_bytes, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    void *local[MP::NWIN] = {sh.x_cache, sh.x_recv_grads, sh.x_recv_blk[0], sh.x_recv_blk[1], sh.x_recv_full[0], sh.x_recv_full[1], mp.flat_recv, mp.flags_local};
    mine.pid = (uint32_t)getpid();
    mine.per_peer = (uint64_t)mp.per_peer; mine.flat_rows = (uint64_t)mp.flat_rows;
    for (int w = 0; w < MP::NWIN; ++w) {
        mine.addr[w] = (uint64_t)(uintptr_t)local[w];
        if (!local[w]) ok = false;
        else if (ok && n > 1 && hipIpcGetMemHandle(&mine.h[w], local[w]) != hipSuccess) { (void)hipGetLastError(); ok = false; }
    }
    mine.ok = ok ? 1u : 0u;
    int rc = mapped_gather(s, comm, stage, &mine, all.data(), sizeof(MappedRec));
    if (rc != PS_OK) { (void)hipFree(stage); return rc; }
    bool all_ok = true;
    for (int p = 0; p < n; ++p) all_ok = all_ok && all[(size_t)p].ok != 0 && all[(size_t)p].flat_rows == (uint64_t)mp.flat_rows;
    uint32_t opened_ok = 1;
    if (all_ok) {
        for (int p = 0; p < n && opened_ok; ++p) {
            const MappedRec &r = all[(size_t)p];
            mp.peer_per_peer[p] = (int64_t)r.per_peer;
            if (p == rank || r.pid == mine.pid) {           // this rank itself, or a rank thread of this process: the addresses as they are
                for (int w = 0; w < MP::NWIN; ++w) mp.win[w][p] = (void *)(uintptr_t)r.addr[w];
                continue;
            }
            mp.opened[p] = true;
            for (int w = 0; w < MP::NWIN && opened_ok; ++w) {
                int same = -1;                              // (a buffer that serves as two windows -- no separate full-size blocks -- is opened once)
                for (int w2 = 0; w2 < w; ++w2) if (r.addr[w2] == r.addr[w]) same = w2;
                if (same >= 0) { mp.win[w][p] = mp.win[same][p]; continue; }
                void *q = nullptr;
                if (hipIpcOpenMemHandle(&q, r.h[w], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened_ok = 0; q = nullptr; }
                mp.win[w][p] = q;
            }
        }
    } else opened_ok = 0;
    // second round: did every rank get every mapping?
    std::vector<uint32_t> oks((size_t)n * 64);
    uint32_t mine_ok[64];
    memset(mine_ok, 0, sizeof mine_ok);
    mine_ok[0] = opened_ok;
    rc = mapped_gather(s, comm, stage, mine_ok, oks.data(), sizeof mine_ok);
    if (rc != PS_OK) { (void)hipFree(stage); mapped_close(m); return rc; }
    bool every = true;
    for (int p = 0; p < n; ++p) every = every && oks[(size_t)p * 64] != 0;
    if (!every) { (void)hipFree(stage); mapped_close(m); return PS_OK; }
    mp.on = true;
    mp.self = n == 1;
    mp.with_lists = g_mapped_lists != 0;
    // ---- the wire check (see k_mapped_fill): R rows per peer, region p of the receive buffers = what peer p stored ----
    {
        const int D = m->cfg.D;
        const int64_t R = std::max<int64_t>(1, std::min<int64_t>(512, std::min<int64_t>(m->nnz_cap / n, mp.per_peer)));
        uint32_t *src = nullptr; unsigned int *bad = nullptr;
        bool alloc_ok = hipMalloc((void **)&src, sizeof(uint32_t) * (size_t)n * R * D) == hipSuccess &&
                        hipMalloc((void **)&bad, sizeof(unsigned int)) == hipSuccess;
        unsigned int nbad = 0;
        if (alloc_ok) {
            std::vector<int64_t> pre((size_t)n + 1), slot0((size_t)n, (int64_t)rank * R), grow((size_t)n);       // "my rows go to your slots rank * R ..."
            for (int p = 0; p < n; ++p) grow[(size_t)p] = (int64_t)rank * mp.peer_per_peer[p];
            for (int p = 0; p <= n; ++p) pre[(size_t)p] = (int64_t)p * R;
            hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned int), s->stream);
            const unsigned int grid = (unsigned int)cdiv((int64_t)n * R * D, 256);
            for (uint32_t round = 1; round <= 3 && e == hipSuccess; ++round)
                for (int kind = 0; kind < 2; ++kind) {
                    hipLaunchKernelGGL(k_mapped_fill, dim3(grid), dim3(256), 0, s->stream, src, (int)R, D, n, (uint32_t)rank, round * 2 + kind);
                    if (mapped_put(m, kind, (const float *)src, pre.data(), D, kind == 0 ? MP::W_CACHE : MP::W_GRADS, kind == 0 ? slot0.data() : grow.data(), false, true, s->stream) != PS_OK) { e = hipErrorUnknown; break; }
                    hipLaunchKernelGGL(k_mapped_check, dim3(grid), dim3(256), 0, s->stream, (const uint32_t *)(kind == 0 ? sh.x_cache : sh.x_recv_grads),
                                       kind == 0 ? R : mp.per_peer, (int)R, D, n, (uint32_t)rank, round * 2 + kind, -1, bad);
                }
            if (e == hipSuccess) e = hipMemcpyAsync(&nbad, bad, sizeof nbad, hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            if (e != hipSuccess) { (void)hipGetLastError(); nbad = 0xFFFFFFFFu; }
            if (store_check_bad_ids(s) == PS_E_STATE) nbad = 0xFFFFFFFFu;         // (a put's wait ran into its bound)
        } else { (void)hipGetLastError(); nbad = 0xFFFFFFFFu; }
        (void)hipFree(src); (void)hipFree(bad);
        mp.selfcheck_bad = nbad; mp.checked = true;
        memset(mine_ok, 0, sizeof mine_ok);
        mine_ok[0] = nbad == 0 ? 1u : 0u;
        rc = mapped_gather(s, comm, stage, mine_ok, oks.data(), sizeof mine_ok);
        (void)hipFree(stage);
        if (rc != PS_OK) { mapped_close(m); return rc; }
        for (int p = 0; p < n; ++p) every = every && oks[(size_t)p * 64] != 0;
        if (!every) { mapped_close(m); mp.selfcheck_failed = true; return PS_OK; }
    }
    return PS_OK;
}

// one exchange of `kind`: peer p's part of src (rows [pre[p], pre[p + 1]) of rowD floats; bcast: the same pre[1] rows for every peer) into
// window `win` of peer p from row dst_row[p] on
void mapped_put_fill(ps_model *m, int kind, const float *src, const int64_t *pre /* [n + 1] */, int rowD, int win, const int64_t *dst_row /* [n] */, bool bcast, bool self, PeerPutArgs &a) {
    ps_model::Shard::Mapped &mp = m->sh.mp;
    static const char *names[] = {"peer_put_rows", "peer_put_grads", "peer_put_blocks", "peer_put_full_blocks", "peer_put_flat"};
    const int n = mp.nranks;
    memset(&a, 0, sizeof a);
    a.npeers = n; a.rank = mp.rank; a.LPR = rowD / 4; a.D = rowD; a.self = (self || mp.self) ? 1 : 0; a.bcast = bcast ? 1 : 0;
    a.src = src; a.ablate = g_mapped_ablate;
    for (int p = 0; p <= n; ++p) a.start[p] = (uint32_t)pre[p];
    for (int p = 0; p < n; ++p) {
        a.dst[p] = (float *)mp.win[win][p];
        a.dst_row[p] = (long long)dst_row[p];
        a.flag_peer[p] = mp.flags(p) + ((size_t)kind * PS_MAX_MAPPED + mp.rank) * PS_PUT_WGS;
    }
    a.flag_mine = mp.flags_local + (size_t)kind * PS_MAX_MAPPED * PS_PUT_WGS;
    if (++mp.epoch[kind] == 0) ++mp.epoch[kind];
    a.epoch = mp.epoch[kind];
    a.bound = wait_bound(m->s->werr(), 120u + (unsigned int)kind);
    a.ts = stamp_next(names[kind]);
    ++mp.puts[kind];
}
int mapped_put(ps_model *m, int kind, const float *src, const int64_t *pre /* [n + 1] */, int rowD, int win, const int64_t *dst_row /* [n] */, bool bcast, bool self, hipStream_t st) {
    PeerPutArgs a;
    mapped_put_fill(m, kind, src, pre, rowD, win, dst_row, bcast, self, a);
    hipLaunchKernelGGL(k_peer_put, dim3(PS_PUT_WGS), dim3(256), 0, st, a);       // (always PS_PUT_WGS workgroups: every rank polls that many words per peer)
    HIPCHK(hipGetLastError());
    return PS_OK;
}

// the id blocks of a step (fixed size: one block per peer) -- wire blocks on the list chain, full-size ones in front of the gather
int mapped_put_blocks(ps_model *m, bool full, int set, bool self, hipStream_t st) {
    ps_model::Shard &sh = m->sh;
    typedef ps_model::Shard::Mapped MP;
    const int n = sh.mp.nranks;
    const int64_t rows = (full ? sh.full_words : sh.blk_words) / 4;        // 16-byte "rows" (the blocks are 64-byte multiples)
    int64_t pre[PS_MAX_MAPPED + 1], drow[PS_MAX_MAPPED];
    for (int p = 0; p <= n; ++p) pre[p] = (int64_t)p * rows;
    for (int p = 0; p < n; ++p) drow[p] = (int64_t)sh.mp.rank * rows;
    return mapped_put(m, full ? MP::K_FULL : MP::K_BLK, (const float *)(full ? sh.x_send_full[set] : sh.x_send_blk[set]), pre, 4,
                      (full ? MP::W_FULL0 : MP::W_BLK0) + set, drow, false, self, st);
}

// flat[i] = slab_0[i] + slab_1[i] + ... in RANK order (the PS's arrival order, net/PServer.java:164-214) -- a ring's order is its own
__global__ __launch_bounds__(256) void k_flat_sum(float *__restrict__ flat, const float *__restrict__ slabs, int64_t slab_floats, int n, int64_t elems, unsigned long long *ts) {
    StampScope stamp(ts);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    float s = slabs[i];
    for (int p = 1; p < n; ++p) s = s + slabs[(size_t)p * slab_floats + i];
    flat[i] = s;
}
// the dense + wide reduction without a collective: every rank stores its flat gradient into every rank's slab of this step's
// parity (its own included), then sums the slabs it holds
int mapped_flat_reduce(ps_model *m, hipStream_t st) {
    ps_model::Shard &sh = m->sh;
    ps_model::Shard::Mapped &mp = sh.mp;
    typedef ps_model::Shard::Mapped MP;
    const int n = mp.nranks;
    const int64_t parity = (int64_t)((mp.epoch[MP::K_FLAT] + 1u) & 1u);
    int64_t pre[PS_MAX_MAPPED + 1], drow[PS_MAX_MAPPED];
    for (int p = 0; p <= n; ++p) pre[p] = (int64_t)p * mp.flat_rows;
    for (int p = 0; p < n; ++p) drow[p] = (parity * n + mp.rank) * mp.flat_rows;
    PSCHK(mapped_put(m, MP::K_FLAT, sh.flat, pre, 4, MP::W_FLAT, drow, true, true, st));
    hipLaunchKernelGGL(k_flat_sum, dim3((unsigned int)cdiv(sh.flat_elems, 256)), dim3(256), 0, st, sh.flat, mp.flat_recv + (size_t)parity * n * mp.flat_rows * 4, mp.flat_rows * 4, n,
                       sh.flat_elems, stamp_next("flat_sum"));
    HIPCHK(hipGetLastError());
    return PS_OK;
}
}  // namespace

// [0] 1 when this model's rows / gradient exchanges go through mapped peer memory, [1] 1 when a rank's own part does too (a 1-rank
// table under mapped_peer = 2), [2] / [3] put launches so far (rows, gradients), [4] 1 when the flag words are fine-grained memory
extern "C" int ps_shard_mapped_info(const ps_model_t *m, int64_t *out5) {
    if (!m || !out5) return ps_set_err(PS_E_BAD_ARG, "null argument");
    const ps_model::Shard::Mapped &mp = m->sh.mp;
    // (the set-up's wire check launches 6 puts of its own: not counted)
    out5[0] = mp.on ? 1 : mp.selfcheck_failed ? -1 : 0; out5[1] = mp.self ? 1 : 0; out5[2] = mp.puts[0] - (mp.checked ? 3 : 0); out5[3] = mp.puts[1] - (mp.checked ? 3 : 0);
    if (mp.on && mp.with_lists) out5[1] |= 2;       // (bit 1: the id blocks and the flat reduction go this way too) out5[4] = mp.flags_fine ? 1 : 0;
    return PS_OK;
}
void shard_mapped_release(ps_model *m) {        // (ps_model_destroy)
    mapped_close(m);
    if (m->sh.mp.flags_local) (void)hipFree(m->sh.mp.flags_local);
    if (m->sh.mp.arrive) (void)hipFree(m->sh.mp.arrive);
    if (m->sh.mp.flat_recv) (void)hipFree(m->sh.mp.flat_recv);
    m->sh.mp.flags_local = nullptr; m->sh.mp.arrive = nullptr; m->sh.mp.flat_recv = nullptr;
}

int g_blk_factor = 2;       // ps_tune_set("blk_factor", f): a wire block holds f * nnz_cap / nranks rows (0: always full-size blocks)
int g_blk_cap = 0;          // ps_tune_set("blk_cap", rows): the wire block's capacity outright (tests: force the overflow exchange)
namespace {
// One thread per unique key: its owner-local row into the owner's wire block (when it fits) and full block; thread u < nranks
// writes owner u's headers [count | has this worker a list longer than a wire block?].
__global__ __launch_bounds__(256) void k_pack_blocks(const uint32_t *__restrict__ send_rows, const uint32_t *__restrict__ owner_start, int nranks,
                                                      int64_t blk_words, uint32_t blk_cap, uint32_t *__restrict__ blk,
                                                      int64_t full_words, uint32_t *__restrict__ full /* or == blk */, unsigned long long *ts) {
    StampScope stamp(ts);
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u < nranks) {
        uint32_t ovf = 0;
        for (int o = 0; o < nranks; ++o) ovf |= (owner_start[o + 1] - owner_start[o] > blk_cap) ? 1u : 0u;
        const uint32_t cnt = owner_start[u + 1] - owner_start[u];
        blk[(size_t)u * blk_words] = cnt; blk[(size_t)u * blk_words + 1] = ovf; blk[(size_t)u * blk_words + 2] = owner_start[u];      // (2: the worker's first cache slot of this owner's rows)
        if (full != blk) { full[(size_t)u * full_words] = cnt; full[(size_t)u * full_words + 1] = ovf; full[(size_t)u * full_words + 2] = owner_start[u]; }
    }
    if (u >= (int64_t)owner_start[nranks]) return;
    int o = 0;
    while (o + 1 < nranks && (uint32_t)u >= owner_start[o + 1]) ++o;
    const uint32_t i = (uint32_t)u - owner_start[o], row = send_rows[u];
    if (i < blk_cap) blk[(size_t)o * blk_words + PS_BLK_HDR + i] = row;
    if (full != blk) full[(size_t)o * full_words + PS_BLK_HDR + i] = row;
}
// owner_start[0..n] of this rank's plan, the received blocks' counts and the OR of every worker's overflow flag -> pinned
// host memory, then the epoch word (the host spins on it: a copy + event record + event wait woke the host 20-40 us late
// in some processes, round 2).  host: [owner_start 0..n | received counts 0..n-1 | overflow | epoch | the workers' cache slots 0..n-1]
// done_flag (round 5, or NULL): "the plan head in front of this launch on the list chain is done" for the device -- the plan's
// tail on side chain 0 (slots, entry lists) waits for it (start_flag[7]); like k_flag_set it stands for the launches in front
// of it on its stream, which have finished and released their writes.
__global__ void k_publish_counts(const uint32_t *__restrict__ owner_start, const uint32_t *__restrict__ recv_blk, const uint32_t *__restrict__ send_blk, int nranks,
                                 int rank, int64_t blk_words, uint32_t *host, uint32_t epoch, unsigned long long *ts, unsigned int *done_flag, unsigned int done_val) {
    StampScope stamp(ts);
    __shared__ unsigned int ovf_s;
    if (threadIdx.x == 0) ovf_s = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i <= nranks; i += blockDim.x) host[i] = owner_start[i];
    for (int i = threadIdx.x; i < nranks; i += blockDim.x) {
        // (this rank's own block is read where it was packed: its self part of the exchange need not have travelled)
        const uint32_t *b = (i == rank ? send_blk : recv_blk) + (size_t)i * blk_words;
        host[nranks + 1 + i] = b[0];
        host[2 * nranks + 3 + i] = b[2];          // where worker i wants its rows from this owner (its first cache slot): the mapped-peer pull's destination
        if (b[1]) atomicOr(&ovf_s, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) host[2 * nranks + 1] = ovf_s;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(host + 2 * nranks + 2, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (done_flag) __hip_atomic_store(done_flag, done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

// begin: everything of a step that reads no weight -- plan, the exchange of the key lists and the publication of the
// exchange's counts -- enqueued without a host wait.  use_side != 0 runs it on the store's prefetch stream (and the
// side communicator) so it can run beside the previous step's training: call begin for step t+1 (on ANOTHER model of
// the same store: its own key lists and activations) before finish of step t.
// Round 5: a begin has two halves.  HEAD: keys, presence map, unique lists, the id blocks, their exchange and the counts'
// publication -- everything that reads the ids only.  TAIL: what overwrites lists the running step's backward still reads
// (slots, entry lists), parked on side chain 0 behind that step's push.  ps_shard_step_finish_begin hands the head of step
// t+1 to step t's forward (BEGIN_HEAD_HOOK: enqueued between the forward's and the backward's launches, on the list chain
// side[2], released by the first forward GEMM's start) and calls BEGIN_TAIL behind the backward; any other begin is BEGIN_ALL.
enum { BEGIN_ALL = 0, BEGIN_HEAD_HOOK = 1, BEGIN_TAIL = 2 };
static int shard_step_begin(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int use_side, bool inside_finish, int mode = BEGIN_ALL);
extern "C" int ps_shard_step_begin(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int use_side) {
    return shard_step_begin(m, batch, comm, use_side, false);
}
int shard_step_begin_hook(ps_model *m) {
    ps_model::Shard &sh = m->sh;
    if (!m->fwd_flag_valid || !sh.hook_batch || !sh.hook_comm) return PS_OK;       // (no forward GEMM announces its start: the begin behind the backward)
    PSCHK(shard_step_begin(m, sh.hook_batch, sh.hook_comm, 0, true, BEGIN_HEAD_HOOK));
    sh.head_done = true;
    return PS_OK;
}
// inside_finish: called by ps_shard_step_finish_begin between a running step's backward and its push
static int shard_step_begin(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int use_side, bool inside_finish, int mode) {
    RoctxRange roctx_range("ps_shard_step_begin");
    if (!m || !batch || !comm || !comm->all_gather || !comm->all_to_all_v || !comm->all_reduce_sum_f32)
        return ps_set_err(PS_E_BAD_ARG, "bad argument");
    ps_store *s = m->s;
    const int nsh = comm->nranks, rank = comm->rank;
    if (nsh < 1 || nsh > PS_PUSH_MAX_PEERS || rank < 0 || rank >= nsh) return ps_set_err(PS_E_BAD_ARG, "bad communicator (1..%d ranks)", PS_PUSH_MAX_PEERS);
    PSCHK(store_enter(s));
    if (use_side && !s->prefetch_stream) {
        PSCHK(pool_stream_acquire(s->device, 1, &s->prefetch_stream));
    }
    ps_model::Shard &sh = m->sh;
    if (!sh.x_ev) {
        HIPCHK(hipEventCreateWithFlags(&sh.x_ev, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sh.done_ev, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sh.flat_ev, hipEventDisableTiming));
    }
    if (!sh.blk_words && m->cfg.kind == PS_MODEL_WIDEDEEP && m->cfg.wide_grad_mode != PS_GRAD_INTENDED && g_wide_slots) {
        // the wide part of the all-reduced buffer as per-worker slots (kernels_emb.h WideUpdArgs.slots): 1 + rows / 24 floats per
        // worker instead of 2 x rows for all of them -- 2.21 -> 1.54 MB at configs[2]; fixed for the model's life, same on every
        // rank (nranks and the table's rows are)
        PSCHK(shard_ensure_state(m, nsh));          // (the flat buffer; the plan would allocate it a few lines further down)
        const int64_t words = cdiv(s->wide.rows, 24), elems = m->dense_elems + 1 + (int64_t)nsh * (1 + words);
        if (elems <= sh.flat_elems) {
            sh.slot_world = nsh; sh.slot_rank = rank; sh.slot_words = words; sh.flat_elems = elems;
            HIPCHK(hipMemsetAsync(sh.flat, 0, sizeof(float) * (size_t)elems, s->stream));
        }
    }
    if (sh.slot_world && (sh.slot_world != nsh || sh.slot_rank != rank)) return ps_set_err(PS_E_STATE, "this model's steps began as rank %d of %d", sh.slot_rank, sh.slot_world);
    if (!sh.blk_words) {
        // a FULL block holds what one worker can ask one owner for at most: every id of a batch, or every row the owner has
        int64_t maxrows = 1;
        for (int o = 0; o < nsh; ++o) {
            int64_t r = 0;
            for (int f = 0; f < s->emb.F; ++f)
                r += s->emb.java_route() ? s->emb.owner_cnt[(size_t)o * s->emb.F + f] : (s->emb.rows[f] > o ? (s->emb.rows[f] - o + nsh - 1) / nsh : 0);
            maxrows = std::max(maxrows, r);
        }
        int64_t full_cap = std::min<int64_t>(m->nnz_cap, maxrows), blk_cap = full_cap;
        // a WIRE block holds blk_factor x the even share of a batch's ids (at configs[2], N = 8: 26 624 rows for an expected
        // ~5 800; round 3 sent 106 560 words per peer and step to carry them: VERDICT r3 weak #9)
        if (nsh > 1 && g_blk_factor > 0) blk_cap = std::min<int64_t>(full_cap, std::max<int64_t>(256, (int64_t)g_blk_factor * ((m->nnz_cap + nsh - 1) / nsh)));
        if (g_blk_cap > 0) blk_cap = std::min<int64_t>(full_cap, g_blk_cap);
        const int64_t blk_words = round_up(PS_BLK_HDR + blk_cap, 16), full_words = round_up(PS_BLK_HDR + full_cap, 16);    // (64-byte multiples)
        const bool has_full = blk_words < full_words;
        if (!has_full) blk_cap = full_cap;
        // overlap mode (decided once per model: every rank must issue a communicator's operations in one order): the key
        // lists of the next step and the all-reduce go to the side chains + their own communicators
        int ov_want = (g_shard_overlap && !use_side && m->multi_stream && !m->cfg.use_graph && dev_waits_ok(s)) ? 1 : 0;
        // ONE all-gather, once per model: every rank's [wire block | full block | wanted overlap mode | header words].  The
        // block sizes are the fixed send and receive sizes of the id exchange -- ranks configured with different max_batch /
        // max_nnz would post mismatched receives and hang -- and the communicator an operation goes to depends
        // on the overlap mode, which depends on per-rank state (dev_waits_ok): the ranks take the minimum.  Every rank sees
        // the same gathered words, so every rank fails here, or none.
        {
            uint32_t *agree_dev = nullptr;
            { RtGuard rt_guard; HIPCHK(hipMalloc((void **)&agree_dev, sizeof(uint32_t) * 4 * (size_t)(nsh + 1))); }
            std::vector<uint32_t> agree((size_t)4 * (nsh + 1));
            // (bit 1 of the mode word: this rank wants, and could do, the rows / gradient exchanges over mapped peer memory)
            const bool mp_want = g_mapped_peer && m->cfg.D % 4 == 0 && (nsh > 1 || g_mapped_peer == 2) && shard_push_grouped_ok(s, nsh);
            agree[0] = (uint32_t)blk_words; agree[1] = (uint32_t)full_words; agree[2] = (uint32_t)ov_want | (mp_want ? 2u : 0u) | (sh.slot_world ? 4u : 0u); agree[3] = PS_BLK_HDR;      // (bit 2: the wide part of the flat buffer travels as per-worker slots)
            int arc = PS_OK;
            bool mp_all = true;
            if (hipMemcpyAsync(agree_dev, agree.data(), 16, hipMemcpyHostToDevice, s->stream) != hipSuccess) arc = ps_set_err(PS_E_HIP, "upload of the agreement words failed");
            if (arc == PS_OK) { comm_select(comm, 0, false); arc = comm->all_gather(comm->ctx, agree_dev, agree_dev + 4, 16, s->stream); }
            if (arc == PS_OK && (hipMemcpyAsync(agree.data() + 4, agree_dev + 4, 16 * (size_t)nsh, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                                 hipStreamSynchronize(s->stream) != hipSuccess)) arc = ps_set_err(PS_E_HIP, "readback of the agreement words failed");
            { RtGuard rt_guard; (void)hipStreamSynchronize(s->stream); (void)hipFree(agree_dev); }
            PSCHK(arc);
            for (int r = 0; r < nsh; ++r) {
                const uint32_t *a = agree.data() + 4 * (size_t)(r + 1);
                if (a[0] != (uint32_t)blk_words || a[1] != (uint32_t)full_words || a[3] != PS_BLK_HDR)
                    return ps_set_err(PS_E_BAD_ARG, "rank %d exchanges id blocks of %u / %u words, this rank (%d) of %lld / %lld: the ranks' models differ in "
                                      "max_batch / max_nnz (or blk_factor)", r, a[0], a[1], rank, (long long)blk_words, (long long)full_words);
                if (((a[2] >> 2) & 1u) != (sh.slot_world ? 1u : 0u))      // (ranks that all-reduce flat buffers of two layouts would add slots to dense vectors)
                    return ps_set_err(PS_E_BAD_ARG, "rank %d reduces the wide part of the flat gradient %s, this rank (%d) %s: the ranks differ in wide_grad_mode or ps_tune_set(\"wide_slots\")",
                                      r, (a[2] & 4u) ? "as per-worker slots" : "as dense vectors", rank, sh.slot_world ? "as per-worker slots" : "as dense vectors");
                ov_want = std::min<int>(ov_want, (int)(a[2] & 1u));
                mp_all = mp_all && (a[2] & 2u) != 0;
            }
            sh.mp.want_all = mp_all;
        }
        {
            RtGuard rt_guard;
            for (int k = 0; k < 2; ++k) {
                HIPCHK(hipMalloc((void **)&sh.x_send_blk[k], sizeof(uint32_t) * (size_t)blk_words * nsh));
                HIPCHK(hipMalloc((void **)&sh.x_recv_blk[k], sizeof(uint32_t) * (size_t)blk_words * nsh));
                HIPCHK(hipMemsetAsync(sh.x_send_blk[k], 0, sizeof(uint32_t) * (size_t)blk_words * nsh, s->stream));
                HIPCHK(hipMemsetAsync(sh.x_recv_blk[k], 0, sizeof(uint32_t) * (size_t)blk_words * nsh, s->stream));
                s->bytes += 2 * (int64_t)sizeof(uint32_t) * blk_words * nsh;
                if (has_full) {
                    HIPCHK(hipMalloc((void **)&sh.x_send_full[k], sizeof(uint32_t) * (size_t)full_words * nsh));
                    HIPCHK(hipMalloc((void **)&sh.x_recv_full[k], sizeof(uint32_t) * (size_t)full_words * nsh));
                    HIPCHK(hipMemsetAsync(sh.x_send_full[k], 0, sizeof(uint32_t) * (size_t)full_words * nsh, s->stream));
                    HIPCHK(hipMemsetAsync(sh.x_recv_full[k], 0, sizeof(uint32_t) * (size_t)full_words * nsh, s->stream));
                    s->bytes += 2 * (int64_t)sizeof(uint32_t) * full_words * nsh;
                } else { sh.x_send_full[k] = sh.x_send_blk[k]; sh.x_recv_full[k] = sh.x_recv_blk[k]; }
            }
            HIPCHK(hipHostMalloc((void **)&sh.counts_host, sizeof(uint32_t) * (size_t)(3 * nsh + 3) + 64, hipHostMallocDefault));
            memset(sh.counts_host, 0, sizeof(uint32_t) * (size_t)(3 * nsh + 3) + 64);
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        sh.blk_words = blk_words; sh.full_words = full_words; sh.blk_cap = blk_cap; sh.full_cap = full_cap; sh.has_full = has_full;
        sh.ov_mode = ov_want;
    }
    const bool ov = sh.ov_mode == 1 && !use_side && !m->profile;
    // overlap: on the list chain (round 5; side chain 0 in rounds 3-4: ps_tune_set("plan_mid", 0)), in order behind the plan
    // head's kernels (which run there while the step trains) -- the counts reach the host long before the running step's
    // push, and nothing waits across streams for the plan head
    hipStream_t st = use_side ? s->prefetch_stream : ov ? (g_plan_mid ? m->side[2] : m->side[0]) : s->stream;
    if (mode == BEGIN_HEAD_HOOK && (!ov || st != m->side[2] || !sh.blk_words)) return ps_set_err(PS_E_STATE, "the plan head's hook needs the overlap mode");
    // this model's previous step still reads its key lists until its finish has run on the training stream
    if (use_side && sh.done_recorded) HIPCHK(hipStreamWaitEvent(st, sh.done_ev, 0));
    {   // the owner side receives at most min(ids of a batch, rows held here) keys from every worker (the worst
        // case used to be nnz_cap per worker whatever the shard held)
        const int64_t per_peer = std::min<int64_t>(m->nnz_cap, std::max<int64_t>(s->emb.total_rows, 1)), rmax = per_peer * nsh, D = m->cfg.D;
        PSCHK(size_once(s, &sh.x_rows_out, &sh.x_rows_cap, rmax * D, sizeof(float)));
        PSCHK(size_once(s, &sh.x_recv_grads, &sh.x_grads_cap, rmax * D, sizeof(float)));
        PSCHK(size_once(s, &sh.x_cache, &sh.x_cache_cap, m->nnz_cap * D, sizeof(float)));
        sh.x_recv_cap = rmax;
        PSCHK(shard_push_reserve(s, nsh));
        // grouped or sorted owner-side push: decided once per model (push_grouped_max_mb is a mutable knob; a finish that
        // re-evaluated it could find the sorted push's list unallocated -- ADVICE r4)
        if (sh.push_grouped < 0) sh.push_grouped = shard_push_grouped_ok(s, nsh) ? 1 : 0;
        // rows and gradients over mapped peer memory: set up once, behind the buffers it maps (collective: every rank asked for it)
        if (!sh.mp.tried) PSCHK(mapped_setup(m, comm, sh.mp.want_all));
        if (!sh.push_grouped) PSCHK(size_once(s, &sh.x_recv_rows, &sh.x_recv_rows_cap, rmax + 1, sizeof(uint32_t)));   // (the sorted push's list)
    }
    // (overlap mode without an early plan -- the first step, a store that fell back to events: the plan's kernels go to side
    //  chain 1 too and are ordered behind the running step's backward, whose lists they overwrite: order_after_main)
    if (mode == BEGIN_TAIL) {
        // the head of this batch's plan went with the running step's forward: stage the batch (the running step's backward is
        // enqueued by now) and go on with the tail
        if (!sh.tail_due) return ps_set_err(PS_E_STATE, "no plan head in front of this tail");
        PSCHK(stage_batch(m, batch, true));
        if (m->cur_nnz != sh.plan_nnz) return ps_set_err(PS_E_STATE, "the plan head was made for another batch");
    } else {
    sh.x_set ^= 1;
    const int set = sh.x_set;
    sh.pack_blk = sh.x_send_blk[set]; sh.pack_full = sh.x_send_full[set];      // (the plan's one launch packs them when it can)
    {
        const int prc = shard_plan_enqueue(m, batch, nsh, st, false, !use_side, ov, mode == BEGIN_HEAD_HOOK);
        sh.pack_blk = sh.pack_full = nullptr;
        PSCHK(prc);
    }
    if (!sh.packed)
    hipLaunchKernelGGL(k_pack_blocks, dim3(cdiv(std::max<int64_t>(sh.plan_nnz, nsh), 256)), dim3(256), 0, st, sh.send_rows, sh.owner_start, nsh, sh.blk_words,
                       (uint32_t)sh.blk_cap, sh.x_send_blk[set], sh.full_words, sh.x_send_full[set], stamp_next("pack_blocks"));
    HIPCHK(hipGetLastError());
    // (a plugged-in table's collective may BLOCK THE HOST -- the tests' gloo tables drain the stream and stage through the host --
    //  while the running step's last delta GEMM holds its slot for side chain 0's "small kernels done": raised now, not by the
    //  tail's spinner behind the exchange.  With rank processes time-slicing one GPU that wait ran into its 2 s bound.)
    if (!comm_is_rccl(comm)) PSCHK(shard_flush_deferred_flag(m));
    {   // the id exchange: fixed size, no host wait.  (Own keys in place: this rank's own block stays where it was packed.)
        std::vector<int64_t> fixed((size_t)nsh, sh.blk_words);
        comm_select(comm, (use_side || ov) ? 1 : 0, comm_own_in_place(comm));
        int rc;
        if (sh.mp.on && sh.mp.with_lists)        // mapped peer: this rank's blocks straight into the peers' receive sets (no RCCL call in the step)
            rc = timed_coll(m, 0, st, [&]() { return mapped_put_blocks(m, false, set, !comm_own_in_place(comm), st); });
        else
            rc = timed_coll(m, 0, st, [&]() { return comm->all_to_all_v(comm->ctx, sh.x_send_blk[set], fixed.data(), sh.x_recv_blk[set], fixed.data(), sizeof(uint32_t), st); });
        comm_select(comm, 0, false);
        PSCHK(rc);
    }
    if (++sh.x_epoch == 0) ++sh.x_epoch;
    // (a plan head on the list chain: this launch also tells the DEVICE that the head is done -- the tail, on side chain 0, waits for it)
    hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(64), 0, st, sh.owner_start, sh.x_recv_blk[set], sh.x_send_blk[set], nsh, rank, sh.blk_words,
                       sh.counts_host, sh.x_epoch, stamp_next("publish_counts"), (sh.tail_due && sh.head_on_list) ? m->start_flag + 7 : (unsigned int *)nullptr, sh.plan_epoch);
    HIPCHK(hipGetLastError());
    sh.x_stream = st;
    if (mode == BEGIN_HEAD_HOOK) {
        if (!sh.tail_due) return ps_set_err(PS_E_STATE, "the hooked plan head left no tail");
        return PS_OK;       // (the tail: BEGIN_TAIL, behind the running step's backward)
    }
    }
    if (sh.tail_due) {          // an early plan: its second half (slots, the backward's entry lists) waits on side chain 0 for
        if (++m->start_epoch == 0) ++m->start_epoch;          // "the running step's backward has finished", raised by that
        sh.pub_epoch = m->start_epoch;                        // step's push (ps_shard_step_finish_begin)
        sh.tail_flag_due = inside_finish;
        // (no running step -- a begin of its own, e.g. the first step of a run on a model that trained before: whatever
        //  read the lists is in front of this launch on the training stream)
        if (!inside_finish) PSCHK(launch_flag_set(m->start_flag + 6, sh.pub_epoch, s->stream));
    }
    PSCHK(shard_plan_enqueue_tail(m, nsh, st));
    if (use_side) HIPCHK(hipEventRecord(sh.x_ev, st));       // (the training stream orders itself behind the prefetch stream)
    sh.x_begun = true; sh.x_side = use_side != 0; sh.x_ov = ov;
    return PS_OK;
}

// finish, with the NEXT step's begin slipped in between "gradients ready" and "push": the plan of batch t+1 and the
// exchange of its key lists read no weight, so they are enqueued before step t's push and updates -- when the host
// then waits for their counts (the step's one host wait) the GPU still has the push, the owner update and the
// replicated update of step t queued, and the host enqueues the start of step t+1 under them.
extern "C" int ps_shard_step_finish_begin(ps_model_t *m, const ps_comm_ops_t *comm, int is_async, const ps_batch_t *next_batch, float *loss) {
    RoctxRange roctx_range("ps_shard_step_finish");
    if (!m || !comm) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    ps_store *s = m->s;
    ps_model::Shard &sh = m->sh;
    if (!sh.x_begun) return ps_set_err(PS_E_STATE, "ps_shard_step_begin first");
    const int nsh = comm->nranks, rank = comm->rank;
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    // PS_HOST_TIMING=1 (measurement): every 1000 calls, the host time of this call outside / inside the wait for the counts
    static const bool host_timing = getenv("PS_HOST_TIMING") != nullptr;
    struct HostTimer {
        bool on; double t0, wait = 0;
        double lap_t = 0, seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // where the enqueue time goes: gather | forward + hook + backward | next begin | gradients | push | flat
        void lap(int k) { if (!on) return; const double t = now(); seg[k] += t - lap_t; lap_t = t; }
        static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
        explicit HostTimer(bool o) : on(o), t0(o ? now() : 0) {}
        ~HostTimer() {
            if (!on) return;
            static thread_local double sum_all = 0, sum_wait = 0, sum_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; static thread_local long calls = 0;
            sum_all += now() - t0; sum_wait += wait;
            for (int k = 0; k < 8; ++k) sum_seg[k] += seg[k];
            if (++calls % 1000 == 0) {
                fprintf(stderr, "[ps_shard_step] host: %.1f us per step enqueueing, %.1f us waiting for the counts  [gather + rows %.1f | forward, plan head, backward %.1f | next begin %.1f | gradients %.1f | push %.1f | flat %.1f]\n",
                        (sum_all - sum_wait) / 1000, sum_wait / 1000, sum_seg[0] / 1000, sum_seg[1] / 1000, sum_seg[2] / 1000, sum_seg[3] / 1000, sum_seg[4] / 1000, sum_seg[5] / 1000);
                sum_all = sum_wait = 0;
                for (int k = 0; k < 8; ++k) sum_seg[k] = 0;
            }
        }
    } host_timer(host_timing);
    const double wait_t0 = host_timing ? HostTimer::now() : 0;
    {   // the step's one host wait: the counts of the rows / gradient exchanges (spin on the epoch word k_publish_counts raises)
        volatile uint32_t *flag = sh.counts_host + 2 * nsh + 2;
        int64_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != sh.x_epoch) {
            if (++spins > (1ll << 22)) {                 // ~seconds: the kernel never ran -- surface the stream's error instead of hanging
                // (the stream the counts' publication was enqueued on: the LIST chain since round 5 -- this named side chain 0 until round 6, so a
                //  slow exchange, e.g. eight rank processes sharing one GPU, was reported as lost after ~0.3 s of spinning)
                HIPCHK(hipStreamSynchronize(sh.x_stream ? sh.x_stream : st));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != sh.x_epoch) return ps_set_err(PS_E_STATE, "the counts of the exchange never arrived");
                break;
            }
            __builtin_ia32_pause();
        }
    }
    if (host_timing) { host_timer.wait = HostTimer::now() - wait_t0; host_timer.lap_t = HostTimer::now(); }
    if (sh.x_side) HIPCHK(hipStreamWaitEvent(st, sh.x_ev, 0));
    const bool was_side = sh.x_side;
    m->dev_ok = dev_waits_ok(s);
    // (overlap: the id blocks arrived on side chain 0.  The host has SEEN the counts the kernel behind that exchange
    //  published, so the blocks are there -- what the training stream enqueues from here on starts later still)
    sh.x_begun = false; sh.plan_pending = false;
    const int D = m->cfg.D, set = sh.x_set;
    std::vector<int64_t> sc((size_t)nsh), rc((size_t)nsh), scpre((size_t)nsh + 1, 0), rcpre((size_t)nsh + 1, 0);
    for (int o = 0; o < nsh; ++o) {
        sc[o] = (int64_t)sh.counts_host[o + 1] - (int64_t)sh.counts_host[o];     // what I request from / push to owner o
        rc[o] = (int64_t)sh.counts_host[nsh + 1 + o];                             // what worker o requests from / pushes to me
        if (sc[o] < 0 || rc[o] < 0 || sc[o] > sh.full_cap || rc[o] > sh.full_cap) return ps_set_err(PS_E_STATE, "bad key count in the exchange (%lld, %lld)", (long long)sc[o], (long long)rc[o]);
        scpre[o + 1] = scpre[o] + sc[o]; rcpre[o + 1] = rcpre[o] + rc[o];
    }
    const int64_t U = scpre[nsh], nrecv = rcpre[nsh];
    sh.U = U;
    if (sc[rank] != rc[rank]) return ps_set_err(PS_E_STATE, "the exchange's self counts differ");
    if (U > m->nnz_cap || nrecv > sh.x_recv_cap)
        return ps_set_err(PS_E_STATE, "exchange counts (%lld requested, %lld received) exceed the buffers sized at ps_shard_step_begin", (long long)U, (long long)nrecv);
    // Some worker's list for some owner did not fit its wire block (every rank has seen every worker's flag: the same
    // decision everywhere): the full-size blocks of this step -- packed beside the wire blocks -- are exchanged now, on the
    // training stream in front of the gather, and every list of this step is read from them.
    const bool ovf = sh.counts_host[2 * nsh + 1] != 0;
    for (int o = 0; o < nsh && !ovf; ++o)
        if (sc[o] > sh.blk_cap || rc[o] > sh.blk_cap) return ps_set_err(PS_E_STATE, "a key list of %lld / %lld rows arrived without its overflow flag (wire blocks hold %lld)", (long long)sc[o], (long long)rc[o], (long long)sh.blk_cap);
    if (ovf && !sh.has_full) return ps_set_err(PS_E_STATE, "overflow flag on full-size blocks");
    sh.x_ovf = ovf;
    // a rank's own keys are read where they are when the table says so (RCCL's always does; ps_native.h PS_COMM_OWN_IN_PLACE)
    const bool alias = comm_own_in_place(comm);
    if (ovf) {
        std::vector<int64_t> fixed((size_t)nsh, sh.full_words);
        comm_select(comm, 0, alias);
        int xrc;
        if (sh.mp.on && sh.mp.with_lists) xrc = timed_coll(m, 0, st, [&]() { return mapped_put_blocks(m, true, set, !alias, st); });
        else xrc = timed_coll(m, 0, st, [&]() { return comm->all_to_all_v(comm->ctx, sh.x_send_full[set], fixed.data(), sh.x_recv_full[set], fixed.data(), sizeof(uint32_t), st); });
        comm_select(comm, 0, false);
        PSCHK(xrc);
        sh.stat[7] += 1;
        sh.stat[1] += (int64_t)(nsh - 1) * sh.full_words * (int64_t)sizeof(uint32_t);
    }
    {   // wire accounting (this rank's own part never travels)
        const int64_t peers = nsh - 1, self = sc[rank];
        sh.stat[0] += 1;
        sh.stat[1] += peers * sh.blk_words * (int64_t)sizeof(uint32_t);
        sh.stat[2] += (U - self) * D * (int64_t)sizeof(float);
        sh.stat[3] += (U - self) * D * (int64_t)sizeof(float);
        sh.stat[4] += nsh > 1 ? sh.flat_elems * (int64_t)sizeof(float) : 0;
        sh.stat[5] += U; sh.stat[6] += nrecv;
    }
    const uint32_t *rows_p[PS_PUSH_MAX_PEERS];
    const float *grads_p[PS_PUSH_MAX_PEERS];
    for (int p = 0; p < nsh; ++p) {
        rows_p[p] = (ovf ? sh.x_recv_full[set] + (size_t)p * sh.full_words : sh.x_recv_blk[set] + (size_t)p * sh.blk_words) + PS_BLK_HDR;
        // (mapped peer: worker p's gradients land in region p of the receive buffer -- a sender cannot know the other workers' counts)
        grads_p[p] = sh.x_recv_grads + (size_t)(sh.mp.on ? (int64_t)p * sh.mp.per_peer : rcpre[p]) * D;
    }
    // (own keys: the FULL block this rank packed for itself -- complete whatever the wire block holds)
    if (alias) { rows_p[rank] = sh.x_send_full[set] + (size_t)rank * sh.full_words + PS_BLK_HDR; grads_p[rank] = m->grads_out + (size_t)scpre[rank] * D; }
    // getList: the owner gathers the requested rows, rows back
    bool rows_fused = false;
    {   // (the gather's launch also holds the join with side chain 0 -- "this step's slots are written" -- for the forward
        //  behind it: a wait on an event that fired long ago still costs the training stream ~3.5 us, round 2)
        LaunchOpts lo;
        if (sh.slot_ev && sh.slot_flag && m->dev_ok) { lo.wait = m->start_flag + 10; lo.wait_val = sh.slot_epoch; }
        GatherSlots gsl;
        memset(&gsl, 0, sizeof gsl);
        if (sh.slots_due) {        // (the plan's slot kernel rides on this launch: shard_plan_enqueue_tail)
            gsl.keys = sh.slots_keys; gsl.nnz = sh.slots_nnz; gsl.bitmap = sh.bitmap; gsl.word_prefix = sh.word_prefix; gsl.slot = sh.slot;
            gsl.wait = m->start_flag + 7; gsl.wait_val = sh.plan_epoch;
            sh.slots_due = false;
        }
        // mapped peer, fused (round 6): the gather's own stores go into the workers' caches and its last workgroup exchanges the
        // flags -- the put launch (one more kernel start + boundary on the chain, ~8 us whatever it moves: tools/r06_put_ablate.py) is gone
        GatherPut gput;
        memset(&gput, 0, sizeof gput);
        if (sh.mp.on && g_mapped_fuse && nrecv > 0 && D % 4 == 0) {
            ps_model::Shard::Mapped &mp = sh.mp;
            rows_fused = true;
            gput.on = 1; gput.rank = rank; gput.self = (!alias || mp.self) ? 1 : 0;
            for (int p = 0; p < nsh; ++p) {
                const int64_t slot0 = (int64_t)sh.counts_host[2 * nsh + 3 + p];
                if (slot0 + rc[p] > m->nnz_cap) return ps_set_err(PS_E_STATE, "worker %d wants %lld rows from slot %lld on: beyond a cache of %lld rows", p, (long long)rc[p], (long long)slot0, (long long)m->nnz_cap);
                gput.dst[p] = mp.cache(p) + (size_t)slot0 * D;
                gput.flag_peer[p] = mp.flags(p) + ((size_t)ps_model::Shard::Mapped::K_ROWS * PS_MAX_MAPPED + rank) * PS_PUT_WGS;
            }
            gput.flag_mine = mp.flags_local;
            if (++mp.epoch[0] == 0) ++mp.epoch[0];
            gput.epoch = mp.epoch[0];
            gput.arrive = mp.arrive;
            ++mp.puts[0];
        }
        PSCHK(shard_serve_pull_lists(s, rows_p, rc.data(), nsh, sh.x_rows_out, &lo, &gsl, rows_fused ? &gput : nullptr));
        if (lo.wait && lo.launched) sh.slot_ev = nullptr;        // (else ps_shard_forward_backward waits for the event)
    }
    int crc = PS_OK;
    if (rows_fused) {
        // (nothing: the gather did it)
    } else if (sh.mp.on) {
        // mapped peer: every worker's rows straight into its cache, at the slot its id block's header names
        // (the slot came in the id block's header and reached the host with the counts: k_publish_counts)
        int64_t slot0[PS_PUSH_MAX_PEERS];
        for (int p = 0; p < nsh; ++p) {
            slot0[p] = (int64_t)sh.counts_host[2 * nsh + 3 + p];
            if (slot0[p] + rc[p] > m->nnz_cap) return ps_set_err(PS_E_STATE, "worker %d wants %lld rows from slot %lld on: beyond a cache of %lld rows", p, (long long)rc[p], (long long)slot0[p], (long long)m->nnz_cap);
        }
        crc = timed_coll(m, 1, st, [&]() { return mapped_put(m, ps_model::Shard::Mapped::K_ROWS, sh.x_rows_out, rcpre.data(), D, ps_model::Shard::Mapped::W_CACHE, slot0, false, !alias, st); });
    } else {
    comm_select(comm, 0, alias);
    crc = timed_coll(m, 1, st, [&]() { return comm->all_to_all_v(comm->ctx, sh.x_rows_out, rc.data(), sh.x_cache, sc.data(), sizeof(float) * (size_t)D, st); });
    comm_select(comm, 0, false);
    }
    PSCHK(crc);
    host_timer.lap(0);
    // train on the cache (this rank's own rows straight from the gather's output)
    sh.alt_W = nullptr; sh.alt_lo = sh.alt_hi = 0;
    if (alias && sc[rank] > 0 && !(rows_fused && sh.mp.self)) {      // (fused + a 1-rank table's self mode: the own rows went to the cache too)
        sh.alt_lo = (uint32_t)scpre[rank]; sh.alt_hi = (uint32_t)scpre[rank + 1];
        sh.alt_W = reinterpret_cast<const float *>(reinterpret_cast<intptr_t>(sh.x_rows_out) + (intptr_t)sizeof(float) * D * ((intptr_t)rcpre[rank] - (intptr_t)scpre[rank]));
    }
    // the previous step's replicated update ran on side chain 1: the gather of this step's forward holds the join when the
    // update's end raised a device flag (enqueue_forward: EmbFwdArgs.end_wait), else an event
    if (sh.flat_pending && !(sh.flat_by_flag && m->dev_ok && m->multi_stream && !m->profile)) {
        // (when the update's end raised a flag its event was not recorded -- a record per step for a wait that normally never
        //  happens; nothing has been enqueued on side chain 1 since, so an event recorded NOW stands for the same work)
        if (sh.flat_by_flag) HIPCHK(hipEventRecord(sh.flat_ev, m->side[1]));
        HIPCHK(hipStreamWaitEvent(st, sh.flat_ev, 0));
        sh.flat_pending = false;
    }
    {
        sh.defer_flag5 = next_batch != nullptr;      // (the next step's plan, enqueued below, opens with a spinner on side chain 0)
        // the next step's plan head goes with this step's forward (round 5: BEGIN_HEAD_HOOK) when it can run on the list chain
        sh.head_done = false;
        if (next_batch && sh.ov_mode == 1 && !was_side && !m->profile && sh.blk_words && g_plan_mid && shard_plan_hook_ok(m, next_batch)) {
            sh.hook_batch = next_batch; sh.hook_comm = comm;
        }
        int frc = ps_shard_forward_backward(m, sh.x_cache, nullptr);
        sh.hook_batch = nullptr; sh.hook_comm = nullptr;
        sh.defer_flag5 = false;
        sh.alt_W = nullptr; sh.alt_lo = sh.alt_hi = 0;
        if (frc != PS_OK) { (void)shard_flush_deferred_flag(m); return frc; }
    }
    host_timer.lap(1);
    // (start_flag[5], "side chain 0's small kernels are done", may be deferred to the next plan's opening spinner: every
    //  return between here and that plan's enqueue raises it first -- the running step's last delta GEMM holds its slot
    //  until the flag is up)
    struct DeferredFlagGuard { ps_model *m; ~DeferredFlagGuard() { (void)shard_flush_deferred_flag(m); } } deferred_flag_guard{m};
    const bool ov2 = sh.ov_mode == 1 && !was_side && !m->profile;      // where the replicated tensors' update goes
    // The flat gradient [fc | wide G | wide C | bias] is consumed on side chain 1 in overlap mode.  Normally it was produced
    // there too (the backward's tail behind the dW GEMMs).  When this step's backward could not use device-side joins -- a
    // wait that timed out, a second model on the device, dev_wait / tail_dev switched off -- it went to the TRAINING stream
    // while the overlap mode, agreed once by all ranks, stays: order the reduction behind it.
    if (ov2 && m->flat_stream && m->flat_stream != m->side[1]) {
        hipEvent_t e = m->events[m->next_event++ % m->events.size()];
        HIPCHK(hipEventRecord(e, m->flat_stream));
        HIPCHK(hipStreamWaitEvent(m->side[1], e, 0));
    }
    // the next step's key lists (same order of operations on every rank)
    if (next_batch) {
        const int brc = shard_step_begin(m, next_batch, comm, 0, true, sh.head_done ? BEGIN_TAIL : BEGIN_ALL);
        sh.head_done = false;
        (void)shard_flush_deferred_flag(m);          // (a no-op when the plan's spinner took it)
        PSCHK(brc);
    }
    host_timer.lap(2);
    // push: the per-key gradients to their owners
    PeerPutArgs gput_args;
    bool grads_in_push = false;
    if (sh.mp.on) {
        int64_t grow[PS_PUSH_MAX_PEERS];          // region `rank` of every owner's receive buffer
        for (int p = 0; p < nsh; ++p) grow[p] = (int64_t)rank * sh.mp.peer_per_peer[p];
        if (nsh > 1 && sh.push_grouped == 1 && g_mapped_fuse) {
            // N >= 2: the put as a role of the owner push's first launch (k_push_mark_put): no launch, no boundary of its own on the chain
            mapped_put_fill(m, ps_model::Shard::Mapped::K_GRADS, m->grads_out, scpre.data(), D, ps_model::Shard::Mapped::W_GRADS, grow, false, !alias, gput_args);
            grads_in_push = true;
            crc = PS_OK;
        } else
        crc = timed_coll(m, 2, st, [&]() { return mapped_put(m, ps_model::Shard::Mapped::K_GRADS, m->grads_out, scpre.data(), D, ps_model::Shard::Mapped::W_GRADS, grow, false, !alias, st); });
    } else {
    comm_select(comm, 0, alias);
    crc = timed_coll(m, 2, st, [&]() { return comm->all_to_all_v(comm->ctx, m->grads_out, sc.data(), sh.x_recv_grads, rc.data(), sizeof(float) * (size_t)D, st); });
    comm_select(comm, 0, false);
    }
    PSCHK(crc);
    host_timer.lap(3);
    if (sh.push_grouped == 1) {
        LaunchOpts lo;
        if (sh.tail_flag_due) { lo.flag = m->start_flag + 6; lo.flag_val = sh.pub_epoch; }
        PSCHK(shard_apply_push_lists(s, rows_p, grads_p, rc.data(), nsh, is_async, true, &lo, grads_in_push ? &gput_args : nullptr));
        if (sh.tail_flag_due && !lo.launched) PSCHK(launch_flag_set(m->start_flag + 6, sh.pub_epoch, st));
        sh.tail_flag_due = false;
    } else {
        // The sort-free push needs a [workers][local rows] position table; beyond 4 GB of it (8 workers x 125 M local rows:
        // configs[3]'s table sharded 8 ways) the owner falls back to the stable sort by row of ps_shard_apply_push: the
        // received lists are packed into one contiguous list first (this case used to fail here with
        // PS_E_UNSUPPORTED, after the rows and gradient exchanges and in front of the all-reduce the peers then hang in).
        if (sh.tail_flag_due) { PSCHK(launch_flag_set(m->start_flag + 6, sh.pub_epoch, st)); sh.tail_flag_due = false; }
        for (int p = 0; p < nsh; ++p)
            if (rc[p] > 0) HIPCHK(hipMemcpyAsync(sh.x_recv_rows + rcpre[p], rows_p[p], sizeof(uint32_t) * (size_t)rc[p], hipMemcpyDeviceToDevice, st));
        if (alias && rc[rank] > 0)
            HIPCHK(hipMemcpyAsync(sh.x_recv_grads + (size_t)rcpre[rank] * D, grads_p[rank], sizeof(float) * (size_t)rc[rank] * D, hipMemcpyDeviceToDevice, st));
        PSCHK(shard_apply_push(s, sh.x_recv_rows, sh.x_recv_grads, nrecv, nullptr, nsh, is_async, true));
    }
    host_timer.lap(4);
    // the dense + wide reduction and the replicated update: on side chain 1 + the side communicator (behind the flat
    // gradient's kernel and the next step's id exchange, beside the push), or in line
    hipStream_t fs = ov2 ? m->side[1] : st;
    if (sh.mp.on && sh.mp.with_lists) {
        // mapped peer: every rank's flat gradient into every rank's slab, summed locally in rank order (mapped_flat_reduce)
        crc = timed_coll(m, 3, fs, [&]() { return mapped_flat_reduce(m, fs); });
        PSCHK(crc);
    } else if (comm_wired(comm)) {
        comm_select(comm, ov2 ? 2 : 0, false);
        crc = timed_coll(m, 3, fs, [&]() { return comm->all_reduce_sum_f32(comm->ctx, sh.flat, sh.flat_elems, fs); });
        comm_select(comm, 0, false);
        PSCHK(crc);
    }
    PSCHK(shard_apply_flat(m, nsh, fs));
    if (ov2) {
        sh.flat_by_flag = false;
        if (m->dev_ok && next_batch && !loss) {        // the next step's gather carries the join (EmbFwdArgs.end_wait); the flag from the device
            if (++m->start_epoch == 0) ++m->start_epoch;
            sh.flat_epoch = m->start_epoch;
            PSCHK(launch_flag_set(m->start_flag + 9, sh.flat_epoch, fs));
            sh.flat_by_flag = true;
        }
        if (!sh.flat_by_flag) HIPCHK(hipEventRecord(sh.flat_ev, fs));
        sh.flat_pending = true;
        if (!next_batch || loss) {      // nothing follows that would join: close the step on the training stream
            HIPCHK(hipStreamWaitEvent(st, sh.flat_ev, 0));
            sh.flat_pending = false;
        }
    }
    host_timer.lap(5);
    if (was_side) {         // (two-model prefetch: this model's next begin, on the prefetch stream, must not overtake this step)
        HIPCHK(hipEventRecord(sh.done_ev, st));
        sh.done_recorded = true;
    }
    if (loss) {
        HIPCHK(hipMemcpyAsync(loss, m->loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        PSCHK(store_check_bad_ids(s));
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

// Wire accounting of this model's ps_shard_step calls so far: out[0] steps, [1] id-block bytes sent, [2] row bytes received,
// [3] gradient bytes sent, [4] all-reduce payload bytes (per rank, before the algorithm's 2 (N-1) / N), [5] unique keys
// requested, [6] keys served as an owner, [7] words of one id block.
extern "C" int ps_shard_exchange_stats(const ps_model_t *m, int64_t *out, int n) {
    if (!m || !out || n < 8) return ps_set_err(PS_E_BAD_ARG, "bad argument (8 values)");
    for (int i = 0; i < 7; ++i) out[i] = m->sh.stat[i];
    out[7] = m->sh.blk_words;
    if (n >= 10) { out[8] = m->sh.stat[7]; out[9] = m->sh.full_words; }
    return PS_OK;
}
