// kernels_emb.h -- argument blocks and launchers of kernels_emb.hip
#pragma once
#include "ps_common.h"

#define PS_EMB_CHUNK 32  // per-key runs longer than this are summed as CH-chunks (oracle: orc_emb_geff chunk=32)
#define PS_EMB_SUPER 32        // runs above PS_EMB_SUPER_MIN chunks: chunk partials are first folded 32 at a time
#define PS_EMB_SUPER_MIN 128   // (oracle: ORC_SUPER / ORC_SUPER_MIN_CHUNKS)

struct EmbFwdArgs {
    const float *W;            // [rows_total][D] all fields' tables back to back
    const int64_t *row_base;   // [F+1] first global row of each field
    const int64_t *ids;        // [nnz]
    const int64_t *offsets;    // nullptr (single-hot, nnz = B*F) or [B*F+1]
    int B, F, D, X, act;
    float *out; int ld;        // [B][ld]
    const float *dense;        // [B][X] or nullptr
    uint32_t *key_out;         // [nnz] global row per entry (sort key) or nullptr
    uint32_t *ent_bag;         // [nnz] bag of each entry (multi-hot) or nullptr
    const uint32_t *slot;      // when set: row of entry p is slot[p] (W = pulled-row cache), ids unused
    const float *W_alt; uint32_t alt_lo, alt_hi;   // slots in [alt_lo, alt_hi) are read from W_alt + slot * D instead (the rows this
                               // rank owns itself never travel: they stay in the owner-side gather's output)
    int *err;                  // out-of-range id counter
    size_t table_bytes;        // size of W (decides the streaming hints)
    int nt;                    // bit 0: non-temporal row loads, bit 1: non-temporal output stores (set by the launcher)
    int LPR, gather_blocks;    // filled by the launcher
    int order;                 // multi-hot (launcher): bit 0 workgroups of one XCD take neighbouring bags, bit 1 bags field-major (k_emb_fwd)
    unsigned long long *ts;    // stamp slot (ps_common.h) or nullptr, set by the launcher
    const unsigned int *end_wait; unsigned int end_val; WaitBound bound;   // the first workgroup ends only once *end_wait reached end_val
    const unsigned int *start_wait; unsigned int start_val;   // no workgroup starts before *start_wait reached start_val (the previous
                               // step's dW GEMMs, on a side chain, still read the activations this launch overwrites)
    // LRLayer.forward as a role of this launch (wide_in_gather): workgroups behind the gather's and the dense features' leave wide_z[b] = the
    // sequential sum of sample b's F wide weights + bias (layer/LRLayer.java:73-84) for the head, which then only adds it
    const int64_t *wide_ids; int64_t wide_rows; const float *wide_w; const float *wide_bias; uint8_t *wide_touched; float *wide_z; int wide_train;
    int wide_blk0, wide_blocks;    // filled by the launcher
};
int launch_emb_fwd(EmbFwdArgs a, hipStream_t st, LaunchOpts *lo = nullptr, unsigned int *werr = nullptr);   // lo: stop_event, wait (an END wait)
int launch_emb_keys(const EmbFwdArgs &a, hipStream_t st);      // multi-hot: key_out / ent_bag from the ids alone
int launch_emb_keys_seg(const EmbFwdArgs &a, const uint32_t *pre, const uint32_t *ftotal, int tile, uint32_t *kp, uint32_t *vp, hipStream_t st);   // (segmented sort)

struct HeadArgs {
    int B, F, wide, train;
    const float *zlast; int ldz;       // last FcLayer output (logit, or P for DNN)
    const int64_t *wide_ids; int64_t wide_rows;
    const float *wide_w; const float *wide_bias; uint8_t *touched;
    const float *labels;
    float *P, *wide_z, *terms;
    const float *wide_z_in;            // when set: LRLayer.forward of this batch was done by the gather's launch (EmbFwdArgs.wide_z): sumW is read, not computed
    float *dlast; int ldd;             // delta at the last FcLayer's output
    int *err;
    // when a_last is set the last FcLayer (out = 1) is computed here as a per-sample dot product
    // (its GEMM would be a 4096 x 1 sliver): z = sum_k a_last[b][k] * w_last[k], bias via the ones column
    const float *a_last; int lda_last; const float *w_last; int k_last; int last_sigmoid;
    float *zout;                       // [B][ldz] column 0 receives the layer's activation
};
struct LastBwdArgs {                   // FcLayer.backward of the out = 1 layer (layer/FcLayer.java:93-110)
    int B, K, Kp, chunk;               // K inputs (+1 ones column), rows per workgroup
    const float *A; int lda;           // the layer's input [B][lda]
    const float *dlast; int ldd;       // delta at its output, column 0
    const float *W; int ldw;           // W' [K+1][ldw], column 0
    float *dprev; int ldp;             // delta for the previous layer [B][ldp] (relu' of A fused); may be dx
    int dprev_cols;                    // columns of dprev to write (K, or F*D for the first layer)
    int mask_cols;                     // columns whose relu' mask applies
    float *part; long long part_stride; int ldpart;   // dW partial slabs, one per workgroup
    const int *skip;
    unsigned long long *ts;
    int prio;                          // raise the waves' priority (the fused step's main chain)
};
int launch_last_bwd(const LastBwdArgs &a, int nsplit, hipStream_t st);
// FcLayer.forward x 2 (+ the head and the out = 1 layer's backward) of 16-row panels in one launch (kernels_panel.hip)
#define PS_PANEL_ROWS 16
struct FwdPanelArgs {
    const float *X; int ldx;           // layer 0's input [B][ldx] (ones column and zero padding included)
    int B, Kpad0, N0, N1;
    const float *W0p, *W1p;            // the two layers' weights in fragment order (FcParams.Wp)
    float *H1; int ld1;                // layer 0's output = layer 1's input [B][ld1] (its ones column is already there)
    float *H2; int ld2;                // layer 1's output [B][ld2]
    unsigned long long *ts;
    unsigned int *flag; unsigned int flag_val;              // "this launch has started" (LaunchOpts.flag)
    const unsigned int *wait_flag; unsigned int wait_val;   // workgroup 0 ends only once *wait_flag reached wait_val
    int prio;
    WaitBound bound;
};
int fwd_panel_shape_ok(int Kpad0, int N0, int N1);
// q / h: the head + last layer's backward of the same rows in the launch (q->chunk must be PS_PANEL_ROWS; head_wgs workgroups
// write a partial slab each, like k_last_bwd's grid), or nullptr
int launch_fwd_panel(const FwdPanelArgs &a, const LastBwdArgs *q, const HeadArgs *h, int head_wgs, hipStream_t st, LaunchOpts *lo = nullptr, unsigned int *werr = nullptr);
int launch_pack_w(const float *Wt, float *Wp, int N, int Kpad, hipStream_t st);      // Wp from Wt (init, load, set; the updates write both)
int launch_head(const HeadArgs &a, float *loss_out, float *gbar_out, int *skip, int force_no_skip, hipStream_t st);   // loss_out NULL: no loss reduction
int launch_loss_reduce(const HeadArgs &a, float *loss_out, float *gbar_out, int *skip, int force_no_skip, hipStream_t st);
int launch_head_last_bwd(const HeadArgs &h, const LastBwdArgs &a, int nsplit, hipStream_t st, LaunchOpts *lo = nullptr);   // lo: stop_event
int head_last_bwd_fusable(int rows_per_wg);

// The updater of an embedding row.  KVStore.update(Map) resolves an updater per KEY (exact key, then any map key that is
// a prefix, then "default": store/KVStore.java:240-252), so "emF3." may name another updater than "emF" does.  The store
// resolves one updater per FIELD (store_fill_field_upd); when they are all the same -- the common case -- ngroups is 1 and
// a kernel uses its `upd` argument with no lookup at all.  Otherwise a row's field (from row_base) picks `upd` (group 0)
// or alt[group - 1].
// An updater key that IS a row's key ("emF3.17.0": Float.toString of the id, layer/EmbeddingField.java:71) is the exact match
// KVStore.update(Map) tries first (store/KVStore.java:242): up to PS_EMB_ROW_OVERRIDES such rows ride along as (local row,
// group) pairs and win over their field's group.  Hard limits of this ABI (refused with PS_E_UNSUPPORTED beyond them, the
// reference's HashMap has none): PS_EMB_UPD_GROUPS distinct embedding updaters, PS_EMB_ROW_OVERRIDES exact-key rows, 64
// fields with per-field updaters; an updater key that ends inside a row's id ("emF3.1": a prefix of emF3.1.0, emF3.10.0,
// emF3.17.0, ...) is refused too.
#define PS_EMB_UPD_GROUPS 8
#define PS_EMB_ROW_OVERRIDES 16
struct FieldUpd {
    const int64_t *row_base;           // [F + 1] first local row of every field (device)
    int F, ngroups;
    unsigned char grp[64];             // field -> group
    UpdParams alt[PS_EMB_UPD_GROUPS - 1];
    int nover;                         // exact-key rows
    uint32_t over_row[PS_EMB_ROW_OVERRIDES];        // local row
    unsigned char over_grp[PS_EMB_ROW_OVERRIDES];   // its group
};
#define PS_EMB_SEQ_TILE 16             // sequential order: runs above this many entries are "long" (own workgroup)
struct EmbBwdArgs {
    int64_t nnz;
    int F, D, grad_mode, apply;
    const uint32_t *sorted_key, *sorted_ent, *seg_start, *seg_id, *nseg;
    const uint32_t *long_list;         // seq_order: (run id, first entry, end) of the runs above PS_EMB_SEQ_TILE entries, *nlong triples (or nullptr)
    const uint32_t *nlong;
    const uint32_t *ftab;              // the field sort's table [F][first run, runs, first long run, long runs] (or nullptr): keys dealt to XCDs by FIELD PAIR
    unsigned long long *ts_partials, *ts_super;   // stamps of the chunked order's first two launches
    const uint32_t *out_slot;          // when set (sharded step, single-hot): grads_out / uniq_* index of a run = out_slot[its first entry]
    const uint32_t *ent_bag;           // nullptr => bag = entry
    const float *delta; int ldd;       // [B][ldd], embedding columns already relu'-masked
    float *partials;                   // [2*ceil(nnz/CH)][D]
    float *partials2;                  // same indexing: super partials of runs above PS_EMB_SUPER_MIN chunks
    int long_runs;                     // a run above PS_EMB_SUPER_MIN chunks is possible (launch k_emb_super)
    int seq_order;                     // 1: every key summed in the reference's strict sample order (one launch)
    int super_blocks;                  // filled by the launcher: workgroups of the chunked order's list role (runs above list_min chunks), or 0
    int list_min;                      // chunked order: a run above this many chunks belongs to a workgroup of the list role (the sort listed every such run);
                                       // PS_EMB_SUPER_MIN: only the runs with super partials, as in round 5 (launcher: 0 -> PS_EMB_SUPER_MIN)
    int long_blocks, short_blocks;     // filled by the launcher: workgroups of the long-key role / of the one-key-per-lane-group role
    int ablate;                        // measurement only (g_seq_ablate)
    float *W, *state;                  // [rows][D], [rows][2][D]
    UpdParams upd;
    FieldUpd fu;
    float *grads_out; uint32_t *uniq_row; uint32_t *uniq_cnt;   // [nseg][D], [nseg], [nseg]
    const int *skip;
    int LPR;
    unsigned long long *ts;
    unsigned int *flag; unsigned int flag_val;      // "this launch has started" for a device-side waiter (set by the launcher)
    const unsigned int *end_wait; unsigned int end_val; WaitBound bound;   // the last launch's first workgroup ends only once *end_wait reached end_val
    int lxcd;                          // filled by the launcher: the long-key role takes the list XCD by XCD too (eighths)
    int xcd;                           // filled by the launcher: 1 = keys / tiles are dealt to workgroups XCD by XCD (kernels_emb.hip emb_vblock)
};
int launch_emb_bwd(EmbBwdArgs a, hipStream_t st, LaunchOpts *lo = nullptr, unsigned int *werr = nullptr);   // lo: flag (the first launch announces its
                                                                                   // start), wait (the last launch's end wait)

struct WideUpdArgs {
    int64_t rows;
    float *W, *state;                  // [rows], [rows][2]
    const uint8_t *touched;
    float *bias, *bias_state;          // [1], [2]
    const float *gbar;
    UpdParams upd;
    const int *skip;
    int mode;                          // 0 local update; 1 fill G/C for the all-reduce; 2 update from reduced G/C; 3 bias only;
                                       // 6 / 7: the same two with the wide part as per-worker SLOTS (below) instead of dense G | C
    float *G, *C;                      // [rows] (+ G[2*rows] = bias gradient), contiguous G|C|bias
    int nworkers;
    // Slot form of the sharded step's wide part (round 5): a worker's contribution to the dense vectors is one number and one bit per
    // key -- G_w[k] = touched_w[k] * gbar_w, C_w[k] = touched_w[k] (compat mode: every touched key carries the batch's mean delta,
    // SURVEY App. A.10) -- so the buffer that is all-reduced holds, per worker, [gbar_w | touched_w packed 24 bits to a float]: every
    // worker fills its own slot and zeroes the others, the SUM is then an all-gather whatever order it adds in (x + 0 is exact; 24-bit
    // integers are exact in f32), and every rank rebuilds G[k] = sum of gbar_w over the workers whose bit is set, in RANK order (the
    // PS's arrival order: net/PServer.java:164-214), and C[k] = their number.  slots = [bias g | world x (1 + slot_words)].
    float *slots; int slot_words, world, rank;
};
struct DenseLayer {
    float *W, *Wt, *S1, *S2;           // W' [K+1 rows][ldw], Wt [N rows][ldwt], state like W'
    float *Wp;                         // Wt in MFMA-fragment order for k_fwd_panel (kernels_panel.hip; FcParams.Wp) or nullptr
    const float *part;                 // split-K partials [nsplit][..][ldp]
    int64_t part_stride;
    int K, N, ldw, ldwt, ldp, nsplit;
    int64_t elem_begin, elem_end;      // flat [k][n] element range, k in [0,K]
    int tile_begin, tile_end;          // 32 x 32 tiles of this layer in the launch (filled by launch_dense_update)
    int row_lo, row_cnt;               // row_cnt > 0: only rows [row_lo, row_lo + row_cnt) of W' (the keyed push updates
                                       // "fc<i>.weights" = rows [0, K) and "fc<i>.bias" = row K separately)
};
struct DenseUpdArgs {
    DenseLayer L[8];
    int nlayers, B, apply;
    UpdParams upd;
    const float *flat_grad;            // when set: use this instead of partials/B
    float flat_div;                    // > 0: divide flat_grad by it (mean over workers after the all-reduce)
    float *grad_out;                   // flat gradient as handed to the updater (or nullptr)
    const int *skip;
    unsigned long long *ts;
    // optional: a wide-table pass in the SAME launch (the sharded step's flat buffer is [fc | wide G | wide C | bias]:
    // filling it and applying it are one kernel each instead of two); wide_blocks = 0: none
    WideUpdArgs wide;
    int wide_blocks, tile_blocks;
    // the fused step's tail: no workgroup starts before *wait_flag reached wait_val (the embedding update has started =
    // the last delta GEMM, which reads W_0, has finished; ps_common.h start_wait); NULL otherwise
    const unsigned int *wait_flag; unsigned int wait_val; WaitBound bound;
    const unsigned int *wait_flag2; unsigned int wait_val2;     // a second start wait (dw_split: the dW GEMM of the other side chain)
    unsigned int *started_flag; unsigned int started_val;       // raised when the launch starts: everything in front of it on its
                                                                // stream -- the dW GEMMs -- has finished
};
int launch_dense_update(const DenseUpdArgs &a, hipStream_t st);
int dense_prereduce(DenseUpdArgs &a, int l, hipStream_t st);     // many slabs -> one, in place (launch_dense_update does it otherwise)

int launch_wide_update(const WideUpdArgs &a, hipStream_t st);
// PServer.push x m + psUpdate for wide keys given as a CSR of pushes per key (net/PServer.java:164-214):
// key_ids[u] (wide row, or `rows` for "wide.bias"), entries [key_off[u], key_off[u+1]) index grads[] in arrival order;
// BSP: g = (sum in arrival order) / count, one updater step; async: one step per push
struct WideListArgs {
    int64_t rows; int nkeys, is_async;
    const int64_t *key_ids; const uint32_t *key_off; const float *grads;
    float *W, *state, *bias, *bias_state;
    UpdParams upd;
};
int launch_wide_list(const WideListArgs &a, hipStream_t st);
int wide_update_blocks(const WideUpdArgs &a);                    // for DenseUpdArgs.wide_blocks

struct WideIntendedArgs {              // wide_grad_mode = intended
    const uint32_t *sorted_key, *sorted_ent, *seg_start, *nseg;
    const float *delta; int ldd;       // delta at the output: [B][ldd], column 0
    int B, F;
    float *W, *state;
    UpdParams upd;
    const int *skip;
    float *G, *C;                      // when set (sharded worker): G[key] = the key's gradient, C[key] = 1, no update here -- the
                                       // owner side averages over the workers that pushed the key (ps_shard_apply_flat)
};
int launch_wide_keys(const int64_t *ids, int64_t n, int64_t rows, uint32_t *keys, int *err, hipStream_t st);
int launch_wide_intended(const WideIntendedArgs &a, int64_t n, hipStream_t st);

int launch_init_emb(float *W, int64_t rows, int D, uint64_t seed, uint64_t table, float scale,
                    int64_t id_first, int64_t id_stride, const uint32_t *ids_dev /* NULL: id_first + r*id_stride */, hipStream_t st);
int launch_init_dense(float *W, float *Wt, int K, int N, int ldw, int ldwt, uint64_t seed,
                      uint64_t table_w, float scale_w, uint64_t table_b, float scale_b, hipStream_t st);
int launch_fill(float *p, int64_t n, float v, hipStream_t st);
int launch_fill_col(float *p, int rows, int ld, int col, float v, hipStream_t st);
int launch_rows_copy(float *table, int64_t row_stride, int64_t col_off, const int64_t *rows_idx_dev,
                     int64_t n, int D, float *buf_dev, int to_table, hipStream_t st);

struct RowsApplyArgs {
    int D, is_async, identity;
    const uint32_t *sorted_key, *sorted_ent, *seg_start, *nseg;
    const float *grads;                // [n][D], indexed by sorted_ent
    float *W, *state;
    UpdParams upd;
    FieldUpd fu;
    const int *skip;
    int LPR;
};
int launch_rows_apply(RowsApplyArgs a, int64_t n, hipStream_t st);

#define PS_PUSH_MAX_PEERS 32
struct PushApplyArgs {
    int D, LPR, is_async, npeers;
    uint32_t peer_start[PS_PUSH_MAX_PEERS + 1];   // entry range of every pushing worker
    int64_t n, R;                      // entries, owner-local rows
    // entry e of worker p = (rows_p[p][e - peer_start[p]], grads_p[p][(e - peer_start[p]) * D ...]): every worker's list
    // where it lies -- the received id blocks of the exchange, the received gradients, and for this rank's own keys the
    // buffers they were produced in (nothing of a rank's own traffic is copied)
    const uint32_t *rows_p[PS_PUSH_MAX_PEERS];    // owner-local row of every entry (unique inside one worker's range)
    const float *grads_p[PS_PUSH_MAX_PEERS];
    unsigned int *flag; unsigned int flag_val;    // "this launch has started" (LaunchOpts.flag), or NULL
    uint32_t *mask;                    // [R] bit w set: worker w pushed this row (all zero between steps)
    uint32_t *pos;                     // [npeers][R] entry of worker w's push for the row
    float *W, *state;
    UpdParams upd;
    FieldUpd fu;
    int *err;
    unsigned long long *ts_mark, *ts_apply;   // stamp slots, set by the launcher
                     // set by the launcher: the mark pass ran in an earlier launch of this push (another updater's)
};
struct PeerPutArgs;
int launch_push_apply(PushApplyArgs a, hipStream_t st, LaunchOpts *lo = nullptr, const PeerPutArgs *put = nullptr);      // put: the mapped-peer gradient put as a role of the mark launch (N >= 2)      // lo: flag (the first launch announces its start)
