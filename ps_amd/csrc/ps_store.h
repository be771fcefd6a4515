// ps_store.h -- the GPU-resident KVStore shard (store/KVStore.java) and the
// model graph built over it (model/DNN.java, model/WideDeepNN.java).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels_emb.h"
#include "ps_common.h"

struct EmbTables {
    int F = 0, D = 0, state_slots = 0;
    int shard = 0, nshards = 1, route_mode = PS_ROUTE_ID_MOD;
    std::vector<int64_t> rows;      // vocabulary per field
    std::vector<int64_t> row_base;  // [F+1] first LOCAL global row of each field
    int64_t total_rows = 0;         // rows held by this shard
    float *W = nullptr;             // [total_rows][D]
    float *state = nullptr;         // [total_rows][2][D]  {M,V} or {Z,N}
    int64_t *row_base_dev = nullptr;
    // PS_ROUTE_JAVA_STRING with nshards > 1 (net/Mod.java:13-15: the owner of a key is a hash of its STRING, so a
    // shard's ids are not an arithmetic progression): per global id g = grow_base[f] + id
    std::vector<int64_t> grow_base;     // [F+1] prefix of the vocabularies
    std::vector<uint8_t> owner_h;       // [G] owner shard of every (field, id)
    std::vector<uint32_t> local_h;      // [G] index of the id among its owner's ids of that field (ascending id)
    std::vector<uint32_t> ids_local_h;  // [total_rows] id held by every local row of THIS shard
    std::vector<int64_t> owner_cnt;     // [nshards][F] ids of field f owned by shard o
    uint8_t *owner_dev = nullptr; uint32_t *local_dev = nullptr; int64_t *grow_base_dev = nullptr;
    bool java_route() const { return route_mode == PS_ROUTE_JAVA_STRING && nshards > 1; }
};

struct WideTable {
    int64_t rows = 0;
    float *W = nullptr, *state = nullptr;  // [rows], [rows][2] {Z,N}
    uint8_t *touched = nullptr;            // LRLayer.weights membership (never cleared)
    float *bias = nullptr, *bias_state = nullptr;
};

struct FcParams {
    bool present = false;
    int K = 0, N = 0;        // in, out
    int Kpad = 0;            // round_up(K+1,16): rows of W' (row K = bias), row stride of Wt
    int ldw = 0;             // round_up(N,16): row stride of W'
    float *W = nullptr;      // W'  [Kpad][ldw]   (reference layout [in][out] + bias row)
    float *Wt = nullptr;     // W'^T [N][Kpad]
    float *Wp = nullptr;     // Wt in MFMA-fragment order [N/16][Kpad/16][64 lanes][4] for k_fwd_panel (N % 16 == 0 only; kernels_panel.hip)
    float *S1 = nullptr, *S2 = nullptr;  // updater state, W' layout
    // KVStore.sum of the layer-granular path (ps_fc_backward): flat [(K+1)][N] sum and its count, until ps_dense_update
    float *pending = nullptr; int pending_cnt = 0;
};

struct ps_store {
    int device = 0;
    uint64_t seed = 0;
    hipStream_t stream = nullptr;      // the stream everything is enqueued on (own_stream, or one adopted from the host)
    hipStream_t own_stream = nullptr;
    hipStream_t prefetch_stream = nullptr;   // ps_shard_step_begin(use_side): the next step's key lists, beside training
    EmbTables emb;
    WideTable wide;
    std::vector<FcParams> fc;
    std::map<std::string, ps_updater_t> updaters;
    int64_t global_step = 0;
    int64_t bytes = 0;
    int *err_dev = nullptr;     // device words: [0] ids outside their table | [1] bounded device-side waits that timed out, [2] which wait
    unsigned int *werr() const { return reinterpret_cast<unsigned int *>(err_dev) + 1; }
    // The fused step's LAST join -- its dense update, on a side chain, is done -- is not paid at the end of the step but by
    // whatever touches the store's stream next: the next training step hangs it on its gather's launch (device flag: the
    // gather and the update's tail run beside each other), every other entry point waits for the event first (store_enter).
    hipEvent_t pending_ev = nullptr; const unsigned int *pending_flag = nullptr; unsigned int pending_val = 0;
    const unsigned int *pending_start = nullptr;    // "the update has STARTED" (same value): its chain's dW GEMMs, which read the
                                                    // activations the next gather overwrites, are done
    bool fwd_pair_off = false;  // a workgroup of k_fc_fwd_pair was not on XCD blockIdx % 8: the two forward GEMMs are launched separately
    bool dev_wait_off = false;  // a device-side wait timed out on this store: every join takes its event form from then on
    int64_t wait_timeouts = 0;
    // scratch for row get/put
    int64_t *idx_dev = nullptr; float *rowbuf_dev = nullptr; int64_t scratch_rows = 0; int scratch_D = 0;
    // scratch for push application
    SortWorkspace push_ws; uint32_t *push_keys = nullptr, *push_ents = nullptr, *push_seg_start = nullptr,
                              *push_seg_id = nullptr, *push_nseg = nullptr; int64_t push_cap = 0;
    uint32_t *push_mask = nullptr, *push_pos = nullptr; int push_pos_peers = 0;   // sort-free push (worker-grouped lists)
    // scratch of the layer-granular backward operators (ps_layer_ops.hip), grow-only
    struct OpScratch {
        float *part = nullptr; int64_t part_cap = 0;           // split-K slabs of ps_fc_backward
        float *masked = nullptr; int64_t masked_cap = 0;       // relu'-masked delta of ps_emb_backward_update
        SortWorkspace ws; int64_t nnz_cap = 0, last_nnz = 0;
        uint32_t *keys = nullptr, *ents = nullptr, *ent_bag = nullptr, *seg_start = nullptr, *seg_id = nullptr, *nseg = nullptr, *uniq_row = nullptr;
        float *partials = nullptr, *partials2 = nullptr, *grads = nullptr;
    } ops;
};

int store_dev_alloc(ps_store *s, void **p, size_t bytes, bool zero);
// "emF<f>.<id>.0" | "wide.weights.<id>.0" | "wide.bias" | "fc<i>.weights" | "fc<i>.bias"
struct ParsedKey { int kind; int idx; int64_t id; };  // kind 0 emb, 1 wide w, 2 wide bias, 3 fc w, 4 fc b
bool store_parse_key(const char *key, ParsedKey *k);
// PServer.push (+ psUpdate) of a list of (owner-local row, gradient) pairs; bump_step: globalStep.incrementAndGet()
int shard_apply_push(ps_store *s, const uint32_t *rows_dev, const float *grads_dev, int64_t n, const int64_t *peer_counts,
                     int npeers, int is_async, bool bump_step);
// updater resolution as KVStore.update(Map): exact key, then prefix, then "default"
int store_resolve_updater(const ps_store *s, const char *key, ps_updater_t *out);
// The embedding rows' updaters, resolved per FIELD (probe "emF<f>." through store_resolve_updater, i.e. exact key, prefix,
// default like KVStore.update(Map), store/KVStore.java:240-252): group 0 -> *upd, the others -> fu->alt; fu->ngroups == 1
// (and no lookup in the kernels) when every field resolves to the same updater.  *stateful: any group other than Simple.
// An updater key naming single rows ("emF3.17") cannot be honoured per field, more than PS_EMB_UPD_GROUPS distinct
// updaters neither: PS_E_UNSUPPORTED.
int store_fill_field_upd(const ps_store *s, UpdParams *upd, FieldUpd *fu, bool *stateful = nullptr);
// local global row of (field, id) on this shard, or -1 when not held here
int64_t store_local_row(const ps_store *s, int field, int64_t id);
int store_ensure_scratch(ps_store *s, int64_t rows, int D);
// after a host wait on the store's stream: report (and clear) the device-side count of ids that were outside their table
int store_check_bad_ids(ps_store *s);      // (... and of bounded device-side waits that timed out: PS_E_STATE)
// every entry point that enqueues on (or waits for) the store's stream: hipSetDevice + the pending join of the last fused step
int store_enter(ps_store *s);
// Streams come from a process-wide pool per device and priority class and go back to it (idle) when their store / model is
// destroyed: 0 = least urgent (a model's side chains), 1 = most urgent (a store's training stream, the prefetch stream), 2 = default
// (copy streams).  A process that builds store after store (tests, bench.py's legs) keeps driving the SAME few streams -- and
// nothing in this library ever touches the NULL stream: one hipMemsetAsync(.., 0) + hipStreamSynchronize(0) in a workspace
// allocator (round 5) brought the default stream's hardware queue into play, and every store + model created after it ran its
// multi-stream steps 1.6-2.2x slower (multi-hot 0.60 ms instead of 0.378, sharded 0.33 instead of 0.15: tools/r05_mh_inproc.py,
// profiles/r05_null_stream.txt; tests/test_abi.py greps the sources for it).
int pool_stream_acquire(int device, int cls, hipStream_t *out);
void pool_stream_release(int device, int cls, hipStream_t st);
int store_settle(ps_store *s);             // the pending join alone (an event wait on the store's stream)
// may this store's models join their streams by device-side flags?  (g_dev_wait, no timeout so far, one live model on the device)
#define PS_MAX_DEVICES 64
#define PS_MAX_MAPPED 32       // (= PS_PUSH_MAX_PEERS = PS_COMM_MAX_RANKS)
#define PS_BLK_HDR 4        // header words of an id block: [count | overflow flag of the sending worker | where the sending worker wants this
                            // owner's rows in its cache (first slot: its owner_start[owner]) -- the mapped-peer pull stores them there | 0]
#include <atomic>
extern std::atomic<int> g_models_on_device[PS_MAX_DEVICES];
bool dev_waits_ok(const ps_store *s);

struct FcBuf {
    float *A = nullptr;  int ldA = 0;     // input activations of layer l: [Bcap][ldA]
    float *dOut = nullptr; int ldD = 0;   // delta at the output of layer l: [Bcap][ldD]
    float *part = nullptr; int nsplit = 1; int64_t part_stride = 0; int ldp = 0;
};

struct ps_model {
    ps_store *s = nullptr;
    bool counted = false;       // in g_models_on_device
    unsigned int *pair_ctr = nullptr, pair_epoch = 0;      // k_fc_fwd_pair: per-row-panel tile counters (never reset) and their launch count
    bool dev_ok = false;        // this step joins its streams by device-side flags (decided once per step in stage_batch)
    ps_model_config_t cfg;
    int Bcap = 0;
    int64_t nnz_cap = 0;
    std::vector<FcBuf> fc;      // nfc entries
    float *out_last = nullptr; int ld_last = 0;   // output of the last FcLayer [Bcap][ld_last]
    float *dx = nullptr; int ldx = 0;             // delta wrt the embedding columns [Bcap][ldx]
    float *P = nullptr, *wide_z = nullptr, *terms = nullptr, *loss_dev = nullptr, *gbar_dev = nullptr;
    int *skip_dev = nullptr;
    // staged inputs
    int64_t *ids_dev = nullptr, *offsets_dev = nullptr, *wide_ids_dev = nullptr;
    float *dense_dev = nullptr, *labels_dev = nullptr;
    // current batch (device views)
    const int64_t *cur_ids = nullptr, *cur_offsets = nullptr, *cur_wide = nullptr;
    const float *cur_dense = nullptr, *cur_labels = nullptr;
    int cur_B = 0; int64_t cur_nnz = 0; bool fwd_done = false, bwd_done = false;
    bool emb_started_valid = false; unsigned int emb_started_epoch = 0;    // the last backward's embedding launch raises start_flag[2] to this when it starts
    bool cur_on_device = false;        // the staged batch is the caller's device-resident batch (complete when handed over: ps_native.h)
    // embedding backward workspaces
    SortWorkspace ws;
    int seg_fits = -1;                 // do the tables' shapes allow the segmented sort (kernels_sort.hip seg_sort_fits); -1: not asked yet
    SegSortWs seg; bool seg_sorted = false;      // multi-hot: the segmented two-pass sort's buffers (allocated at the first such batch) | used by this step
    uint32_t *keys = nullptr, *ents = nullptr, *ent_bag = nullptr, *seg_start = nullptr, *seg_id = nullptr,
             *nseg_dev = nullptr, *uniq_row = nullptr, *uniq_cnt = nullptr;
    uint32_t *nseg_cur = nullptr;      // nseg_dev or nseg_dev + 4: the run count of the plan the NEXT backward uses (an early plan of
                                       // step t+1 is written while step t's backward still reads its own)
    bool sort_deferred = false;        // the field sort of this step is enqueued by the backward (late sort)
    bool fwd_flag_valid = false;       // the running step's first forward GEMM raises start_flag[4] = fwd_epoch when it starts
    uint32_t *sorted_keys = nullptr, *sorted_ents = nullptr;   // where the last sort left its result
    uint32_t *seg_nseg_scratch = nullptr;                     // run count of a side sort whose nseg the bitmap plan already wrote
    // single-hot batches: one-launch field sort (kernels_sort.hip field_sort_segments)
    uint32_t *fs_keys = nullptr, *fs_ents = nullptr, *long_list = nullptr;
    unsigned long long *fs_pub = nullptr; uint32_t fs_epoch = 0;
    int list_min = 0;                                         // chunks: the chunked order's list holds every run above this many (0: PS_EMB_SUPER_MIN)
    bool long_list_valid = false;                             // the last sort filled long_list / *nlong_ptr
    bool field_sorted = false;                                // ... and it was the one-launch field sort
    const uint32_t *nlong_ptr = nullptr;
    float *partials = nullptr, *partials2 = nullptr, *grads_out = nullptr;
    float *dense_grad_flat = nullptr; int64_t dense_elems = 0;
    // wide_grad_mode = intended: sort of the batch's wide ids (allocated on first use)
    SortWorkspace wws;
    uint32_t *wkeys = nullptr, *wents = nullptr, *wseg_start = nullptr, *wseg_id = nullptr, *wnseg = nullptr;
    // per-kernel-group event timing (ps_model_set_profile)
    struct ProfEvent { const char *name; hipEvent_t a, b; };
    bool profile = false;
    bool keep_grads = false;        // ps_model_set_keep_grads: the fused step also writes every key's gradient (grads_out / uniq_row / uniq_cnt) for ps_model_get_emb_grads
    bool grads_kept = false;        // the last backward did
    std::string prof_filter;    // when set: only these kernel groups (comma-separated names) are bracketed
    std::vector<ProfEvent> prof_events;
    std::map<std::string, std::pair<long, double>> prof_acc;
    // sharded (multi-GPU) step state: the worker half of PSRouterClient.getList / push
    struct Shard {
        bool active = false;          // forward reads rows from `cache` through `slot`
        int nshards = 1;
        const float *cache = nullptr; // [U][D] rows pulled from their owners, in send order
        uint32_t *slot = nullptr;     // [nnz] unique slot of every entry
        uint32_t x_epoch = 0;         // epoch of the last counts publication (host spins on it)
        hipEvent_t slot_ev = nullptr; // set while the slots are being written on a side stream (one of the model's events)
        bool slot_flag = false; uint32_t slot_epoch = 0;   // ... and start_flag[10] = slot_epoch is raised behind them
        uint32_t *send_rows = nullptr;// [U] owner-local row of every unique key, grouped by owner
        uint32_t *owner_start = nullptr; // [nshards+1] device
        int64_t *lrb_dev = nullptr;   // [nshards][F+1] local row bases of every shard
        float *flat = nullptr;        // [dense_elems | wideG | wideC | wide bias g] for the all-reduce
        int64_t flat_elems = 0;
        // slot form of the wide part (kernels_emb.h WideUpdArgs.slots): decided at the model's first ps_shard_step_begin, where the
        // worker learns its rank -- flat = [dense_elems | wide bias g | world x (gbar_w, touched_w in 24-bit words)], flat_elems shrinks
        int slot_world = 0, slot_rank = 0; int64_t slot_words = 0;
        int sbits = 0;
        int64_t U = 0;
        uint32_t *bitmap = nullptr, *word_prefix = nullptr, *blk_sum = nullptr; int64_t bm_words = 0;   // sort-free plan
        uint8_t *stamp = nullptr, epoch = 0;    // presence bytes of the composite key space, stamped with the plan's epoch
        unsigned long long *plan_pub = nullptr; uint32_t plan_seq = 0;   // k_plan_fused: per-workgroup totals tagged with the plan's number
        // where the NEXT plan packs its unique keys (set by ps_shard_step_begin; NULL: plan only) | it did (one launch with the plan)
        // ps_shard_step's pipeline: "side chain 0's small kernels are done" (start_flag[5]) is not raised by a launch of its own
        // but by the spinner that opens the NEXT step's plan on the same stream (launch_set_then_spin) -- requested by
        // ps_shard_step_finish_begin around the backward, consumed by shard_plan_enqueue, flushed as a plain launch otherwise
        bool defer_flag5 = false, deferred = false; unsigned int *def_flag = nullptr; unsigned int def_val = 0;
        // The field sort of a step's plan (the backward's entry lists) is needed ~100 us after the slots: it is not launched with
        // them but by the step's FORWARD, behind a spinner on its first GEMM's start -- like the fused step's sort.  Launched
        // with the slots it ran beside the gather and the first forward GEMM, whose 512 workgroups then found 26 CUs taken:
        // that GEMM took 24.1 us in the sharded step against 18.7 in the fused one (profiles/r04_shard_gpu_timeline.txt).
        bool sort_due = false; int sort_kb = 0; bool sort_based = false;
        uint32_t *pack_blk = nullptr, *pack_full = nullptr; bool packed = false;
        uint32_t *owner_start_host = nullptr;   // pinned readback of owner_start
        hipEvent_t plan_ev = nullptr;           // the plan's kernels + readback are done
        bool plan_pending = false;
        // ps_shard_step (library-driven exchange).  The key lists travel as FIXED-SIZE blocks, one per peer:
        // [count | count owner-local rows | padding], blk_words words each (1 + the most rows any owner can be asked for), so
        // the id exchange needs no split sizes -- it is enqueued without a host wait, before the running step's push --
        // and carries the counts of the two weight-dependent exchanges (rows back, gradients out) with it.  Two sets:
        // step t+1's lists are exchanged while step t's push still reads its own.
        // Round 4: a WIRE block holds blk_cap rows = blk_factor * nnz_cap / nranks (the expected list is ~ unique keys / nranks:
        // a 16th of what a full block holds at N = 8), behind PS_BLK_HDR header words [count | this worker overflowed
        // somewhere].  The FULL blocks (full_cap = min(ids of a batch, rows of the largest shard) rows: round 3's blocks) are
        // packed beside them and travel only in a step where some worker's list for some owner did not fit -- every rank sees
        // every worker's flag, so all of them take that second exchange or none does.  has_full == false: the wire block IS
        // the full block (one rank; tiny tables).
        int64_t blk_words = 0, full_words = 0, blk_cap = 0, full_cap = 0;
        bool has_full = false;
        uint32_t *x_send_blk[2] = {nullptr, nullptr}, *x_recv_blk[2] = {nullptr, nullptr};    // [nranks][blk_words]
        uint32_t *x_send_full[2] = {nullptr, nullptr}, *x_recv_full[2] = {nullptr, nullptr};  // [nranks][full_words] (== the wire blocks when !has_full)
        bool x_ovf = false;                                          // the step begun last (set by its finish): its lists came at full size
        int x_set = 0;                                               // the set of the step begun last
        // what this rank put on / took off the wire so far (ps_shard_exchange_stats): steps, id-block bytes sent, row bytes received,
        // gradient bytes sent, all-reduce payload bytes, unique keys requested, keys served
        int64_t stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};                  // [7]: steps that took the full-size second exchange
        // ps_tune_set("comm_timing", 1) (measurement): HIP events around every collective, by kind (0 id blocks, 1 rows, 2 gradients,
        // 3 all-reduce) -> coll_acc[2 k] calls, [2 k + 1] ms (ps_shard_collective_times)
        struct CollEv { hipEvent_t a, b; int kind; };
        std::vector<CollEv> coll_ev; double coll_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int ov_mode = -1;                                            // 1: key lists of t+1 and the all-reduce on side chain 1 + the side communicator (decided at the first begin)
        bool x_ov = false;                                           // the step begun last enqueued its id exchange on side chain 1
        bool tail_flag_due = false;                                  // the running step's push must raise start_flag[6] = pub_epoch
        bool flat_by_flag = false;                                   // the replicated update's end also raised start_flag[9] = flat_epoch
        const float *alt_W = nullptr; uint32_t alt_lo = 0, alt_hi = 0;   // this rank's own rows of the running step (EmbFwdArgs.W_alt)
        uint32_t *counts_host = nullptr;                             // pinned: [owner_start 0..nranks | received counts 0..nranks-1 | epoch]
        hipEvent_t flat_ev = nullptr;                                // the replicated tensors' update was enqueued on side chain 1
        bool flat_start_valid = false; uint32_t flat_start_epoch = 0;   // the last backward's flat-gradient launch raises start_flag[12] when it starts
        bool flat_pending = false; uint32_t flat_epoch = 0;          // ... and the main chain has not joined it yet
        uint32_t *x_recv_rows = nullptr; int64_t x_recv_cap = 0;     // (x_recv_rows: the sorted push's contiguous copy of the received lists)
        int64_t x_recv_rows_cap = 0;
        int push_grouped = -1;        // the sort-free owner-side push serves this model (decided ONCE, at its first begin: the knob behind it may move later)
        float *x_rows_out = nullptr; int64_t x_rows_cap = 0;
        float *x_recv_grads = nullptr; int64_t x_grads_cap = 0;
        float *x_cache = nullptr; int64_t x_cache_cap = 0;
        hipEvent_t x_ev = nullptr, done_ev = nullptr;   // begin's work is done | finish's work was enqueued
        bool x_begun = false, x_side = false, done_recorded = false;
        hipStream_t x_stream = nullptr;                 // where the last counts' publication was enqueued (the host's wait falls back to a sync of it)
        // the NEXT step's plan enqueued on side chain 0 while this step trains (shard_plan_enqueue, early): its slot /
        // entry-list half still to be enqueued behind the counts' publication | epoch of "plan done" (start_flag word 7)
        bool tail_due = false;
        uint32_t plan_epoch = 0, pub_epoch = 0;
        int64_t tail_nnz = 0;
        // Round 5: the plan head of step t+1 is enqueued BETWEEN the forward and the backward of step t (ps_shard_forward_backward
        // calls back into ps_comm.hip: hook_*), on the list chain side[2], released by step t's first forward GEMM: the counts of
        // step t+1 reach the host ~50 us into step t instead of ~116 (they were behind side chain 0's small kernels), so the
        // host -- which needs them to size the rows / gradient exchanges of step t+1 -- is no longer 20 us from starving the GPU.
        // keys2: the plan head writes step t+1's sort keys while step t's field sort still reads its own (m->keys and keys2 swap).
        // head_on_list: the tail (slots, entry lists; side chain 0) also waits for "plan head done" (start_flag[7] = plan_epoch,
        // raised by the counts' publication on side[2]).
        uint32_t *keys2 = nullptr;
        const ps_batch_t *hook_batch = nullptr; const ps_comm_ops_t *hook_comm = nullptr;
        bool head_done = false, head_on_list = false;
        int64_t plan_nnz = 0;                                        // ids of the batch the last plan head was made for
        // slots_due (round 5, ps_tune_set("slots_in_gather")): the plan's slot kernel did not get a launch of its own (behind a spinner on side
        // chain 0, a flag setter behind it): the NEXT owner-side gather's launch computes the slots in extra workgroups (ps_shard.hip)
        bool slots_due = false; const uint32_t *slots_keys = nullptr; int64_t slots_nnz = 0;
        // Round 6: rows and gradients over MAPPED PEER MEMORY (ps_comm.hip, "mapped peer"): every rank maps every peer's row cache,
        // gradient receive buffer and flag words (hipIpcOpenMemHandle; the handles travel in one all-gather at the model's first
        // begin) and the two exchanges on the step's critical chain become ONE launch each of this rank's own -- 16-byte
        // write-through stores into the peers' buffers, a flag per peer behind the drained stores, then a bounded wait for the
        // peers' flags -- instead of a grouped ncclSend / ncclRecv (16-18 us of launch and handshake each, profiles/r05_rccl_env_sweep.txt).
        // mp_on: agreed by all ranks (the minimum of what each wanted and could do).  mp_self: a 1-rank table moves its own part the
        // same way (measurement: the `mapped_peer` mode of bench.py's sharded_n1 leg).
        struct Mapped {
            bool on = false, self = false, tried = false, want_all = false;
            int nranks = 0, rank = 0;
            // the peers' buffers this rank stores into, as mapped here (win[w][rank]: the local pointer): the row cache, the gradient
            // receive buffer, the received id blocks (two sets, wire and full size), the flat-gradient slabs, the flag words
            enum { W_CACHE = 0, W_GRADS, W_BLK0, W_BLK1, W_FULL0, W_FULL1, W_FLAT, W_FLAGS, NWIN };
            void *win[NWIN][PS_MAX_MAPPED] = {};
            bool opened[PS_MAX_MAPPED] = {};                                    // peer p's windows came from hipIpcOpenMemHandle
            // exchange kinds (each its own flag words and epoch: two kinds may be in flight on two streams)
            enum { K_ROWS = 0, K_GRADS, K_BLK, K_FULL, K_FLAT, NKIND };
            unsigned int *flags_local = nullptr; bool flags_fine = false;       // this rank's flag words [NKIND][PS_MAX_MAPPED senders][PS_PUT_WGS] (fine-grained when the runtime gives it)
            unsigned int *arrive = nullptr;                                     // the fused gather's workgroups that have drained their stores
            unsigned int epoch[NKIND] = {};                                     // exchanges of each kind so far (the same on every rank)
            int64_t per_peer = 0;                                               // rows of one worker's region in this rank's x_recv_grads
            int64_t peer_per_peer[PS_MAX_MAPPED] = {};                          // ... and in peer p's (shards differ by a row per field)
            float *flat_recv = nullptr; int64_t flat_rows = 0;                  // [2 parities][nranks][flat_rows x 4 floats]: every rank's flat gradient, summed here in RANK order
            bool with_lists = false;                                            // the id blocks and the flat reduction go this way too (no RCCL call in the step)
            int64_t puts[NKIND] = {};                                           // launches so far (ps_shard_mapped_info)
            unsigned int selfcheck_bad = 0; bool selfcheck_failed = false, checked = false;   // the set-up's wire check: wrong words seen here | some rank saw some
            float *cache(int p) const { return (float *)win[W_CACHE][p]; }
            float *grads(int p) const { return (float *)win[W_GRADS][p]; }
            unsigned int *flags(int p) const { return (unsigned int *)win[W_FLAGS][p]; }
        } mp;
    } sh;
    // host batches: pinned staging + two device slots on a copy stream (stage_batch)
    struct HostStage {
        hipStream_t copy_stream = nullptr;
        char *pin[2] = {nullptr, nullptr}, *dev[2] = {nullptr, nullptr};
        hipEvent_t copied[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
        bool done_rec[2] = {false, false};
        size_t off[5] = {0, 0, 0, 0, 0}, bytes = 0;
        uint64_t turn = 0;
    } hstage;
    // side streams: independent chains of the step (sort | dW + dense update | wide update) run
    // beside the main FC chain; fork/join through events (also what the captured graph records)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};     // [2]: the sharded step's LIST chain (the next step's plan head, the id exchange, the
                                                           // counts' publication: ps_comm.hip), beside the running step's forward
    hipStream_t flat_stream = nullptr;   // where the last backward's dense-gradient launch went (ps_shard_step orders its all-reduce behind it)
    std::vector<hipEvent_t> events; size_t next_event = 0;
    bool multi_stream = true;
    bool side0_pending = false;   // a forward forked the sort chain and no backward joined it yet
    HeadArgs head_args;           // the head of the last forward (the loss reduction may be launched by the backward)
    bool head_bwd_done = false;   // the head's launch also did the out = 1 layer's backward
    bool loss_pending = false;    // loss / gbar / stop flag not reduced yet
    hipEvent_t loss_ev = nullptr, s0_ev = nullptr, dw_ev = nullptr, tail_ev = nullptr;
    unsigned int sort_epoch = 0, fwd_epoch = 0, s0_epoch = 0;
    unsigned int *start_flag = nullptr, start_epoch = 0;      // device word + host epoch of the spinner in front of the dW chain
    hipEvent_t head_ev = nullptr;                             // carried by the head's launch (one of `events`), consumed by the backward
    // graph replay: one instantiated graph per (batch pointers, B, nnz)
    struct GraphEntry { const void *sig[5]; int B; int64_t nnz; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
};

// shared between ps_model.hip and ps_shard.hip
int stage_batch(ps_model *m, const ps_batch_t *b, bool need_labels);
int shard_plan_enqueue(ps_model *m, const ps_batch_t *batch, int nshards, hipStream_t st, bool readback, bool early = false,
                       bool order_after_main = false, bool hook = false);   // ps_shard.hip
int shard_apply_flat(ps_model *m, int nworkers, hipStream_t st);
int shard_ensure_state(ps_model *m, int nshards);      // the worker half's buffers (idempotent for one shard count)
int shard_flush_deferred_flag(ps_model *m);       // ps_shard.hip
bool shard_plan_hook_ok(const ps_model *m, const ps_batch_t *batch);     // ps_shard.hip: may this batch's plan head go through the hook
int shard_step_begin_hook(ps_model *m);           // ps_comm.hip: the next step's plan head, called by ps_shard_forward_backward between forward and backward
int shard_launch_deferred_sort(ps_model *m, bool behind_fwd_flag);   // ps_shard.hip: the plan's field sort, enqueued by the forward
int shard_plan_enqueue_tail(ps_model *m, int nshards, hipStream_t st);      // early plans: the slots + the backward's entry lists
int enqueue_forward(ps_model *m, bool train, bool defer_loss);   // defer_loss: enqueue_backward launches the loss reduction
int enqueue_backward(ps_model *m, bool apply);
int shard_push_reserve(ps_store *s, int npeers);   // ps_shard.hip
bool shard_push_grouped_ok(const ps_store *s, int npeers);
// the slot kernel's arguments when it rides on the gather's launch (Shard::slots_due); keys == NULL: none
#define PS_PUT_WGS 128      // flag words per (exchange kind, sending rank): one per workgroup of a put launch (ps_comm.hip)
// mapped peer, fused form (ps_comm.hip): the owner-side gather stores worker p's rows straight into that worker's cache -- no
// x_rows_out round trip, no put launch -- and its last workgroup raises this rank's flag words at every peer and waits for theirs
struct GatherPut {
    int on, rank, self;
    float *dst[PS_MAX_MAPPED];                  // worker p's cache at the slot its id block named
    unsigned int *flag_peer[PS_MAX_MAPPED];     // peer p's PS_PUT_WGS flag words for (rows, this rank)
    const unsigned int *flag_mine;              // this rank's: [p][PS_PUT_WGS]
    unsigned int epoch;
    unsigned int *arrive;                       // gather workgroups that have drained their stores
};
struct GatherSlots { const uint32_t *keys; int64_t nnz; const uint32_t *bitmap, *word_prefix; uint32_t *slot; const unsigned int *wait; unsigned int wait_val; };
void shard_mapped_release(ps_model *m);     // ps_comm.hip: unmap the peers' buffers, free the flag words
extern int g_mapped_peer, g_mapped_ablate, g_mapped_fuse, g_mapped_lists;
int shard_serve_pull_lists(ps_store *s, const uint32_t *const *rows_p, const int64_t *counts, int npeers, float *rows_out_dev,
                           LaunchOpts *lo, const GatherSlots *gs = nullptr, const GatherPut *gp = nullptr);       // lo: wait (an END wait of the gather's launch)
struct PeerPutArgs;
int shard_apply_push_lists(ps_store *s, const uint32_t *const *rows_p, const float *const *grads_p, const int64_t *counts, int npeers,
                           int is_async, bool bump_step, LaunchOpts *lo, const PeerPutArgs *put = nullptr);     // put: ps_put.h, kernels_emb.h launch_push_apply
int finish_step(ps_model *m, float *loss);
