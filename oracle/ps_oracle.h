/*
 * ps_oracle.h -- CPU restatement ("oracle") of the wudikua/ps hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ps_amd/ may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (pure Java + jblas/grpc jars) can be neither
 * compiled nor run in this environment (no JVM, no jars) and its own tests
 * hold no expected values for this path (src/test/java: no assertions).
 * The oracle is therefore pinned only by the hand-derivable known-answer
 * tests of SURVEY.md App. B (tests/test_oracle_kat.py) and by an independent
 * numpy restatement (tests/np_restatement.py) that must agree bit-for-bit.
 *
 * All file:line citations are into /root/reference/src/main/java/.
 *
 * Memory layout: a jblas FloatMatrix is column-major rows x cols
 * (index(i,j) = i + rows*j).  The reference keeps features as rows and the
 * batch as columns, so an activation "features x B" is byte-for-byte a C
 * row-major [B][features] array, and a weight "out x in" is row-major
 * [in][out].  Every array crossing this API uses that layout.
 */
#ifndef PS_ORACLE_H
#define PS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- strings / routing ------------------------- */
/* java.lang.String.hashCode over the bytes of an ASCII key. */
int32_t orc_java_hashcode(const char *s);
/* java.lang.Float.toString for finite v (shortest round-trip digits, Java
 * formatting rules: plain decimal in [1e-3,1e7), else d.dddE[-]x). */
int orc_float_to_string(float v, char *buf, int cap);
/* "emF<f>.<Float.toString(id)>"  layer/EmbeddingField.java:71 + EmbeddingLayer.java:52 */
int orc_emb_key(int field, float id, char *buf, int cap);
/* "wide.weights.<Float.toString(id)>"  layer/LRLayer.java:78 */
int orc_wide_key(float id, char *buf, int cap);
/* net/Mod.java:13-15 : key.hashCode() % n  (Java %, may be negative);
 * floor_mod=1 gives the documented fix Math.floorMod. */
int orc_mod_shard(const char *key, int n, int floor_mod);
/* util/MatrixUtil.java:27-33 : Java float % (fmodf) */
float orc_matrixutil_hash(float id, int size);

/* ---- deterministic init shared (by definition) with the HIP library --- */
/* +-U(0, s) from a counter hash of (seed, table, row, col).  Replaces the
 * reference's unseeded RandomUtils (util/MatrixUtil.java:62-74). */
float orc_init_value(uint64_t seed, uint64_t table, uint64_t row, uint64_t col, float scale);
float orc_xavier_scale(int in_dims, int out_dims); /* (float)(4*sqrt(6)/sqrt(in+out)) */
#define ORC_TABLE_WIDE   (1ull << 20)
#define ORC_TABLE_WIDE_B ((1ull << 20) + 1)
#define ORC_TABLE_FC(i)  ((2ull << 20) + 2ull * (uint64_t)(i))     /* weights; +1 = bias */

/* ---- elementwise arithmetic (each op individually rounded, no FMA) ---- */
float orc_sigmoid_clip(float x);                       /* activations/Sigmoid.java:11 */
/* update/AdamUpdater.java:57-70 on n elements; M,V zero on first touch. */
void orc_adam_update(float *w, const float *g, float *M, float *V, int n,
                     float alfa, float beta1, float beta2, float eps);
/* update/FtrlUpdater.java:51-76 on n elements; returns 0 if skipped (g[0]==0). */
int orc_ftrl_update(float *w, const float *g, float *z, float *nn, int n,
                    float alfa, float beta, float l1, float l2);
/* Effective per-key embedding gradient of one training step
 * (layer/EmbeddingField.java:86-104 run twice, store/KVStore.java:192-203,
 * SURVEY App. A.6).  gk = the n masked per-sample gradients [n][D] in batch
 * order.  mode 0 = compat (double-backward quirk), 1 = intended (mean).
 * chunk<=0: strictly sequential f32 sums (the reference order); chunk>0:
 * partial sums over consecutive chunks of that many samples, then the
 * partials summed in order (the order the HIP long-segment path uses). */
void orc_emb_geff(const float *gk, int n, int D, float *out, int mode, int chunk);

/* loss/CrossEntropy.java:10-28 */
float orc_ce_forward(const float *p, const float *y, int B);
void orc_ce_backward(const float *p, const float *y, int B, float *delta);

/* plain column-major sgemm restatement: C[MxN] = A[MxK] * B[KxN]
 * (sequential k, separate mul/add).  Also a float64 twin for tolerances. */
void orc_sgemm_nn(int M, int N, int K, const float *A, const float *Bm, float *C);

/* ---- the reference-shaped model (string-keyed KVStore, per-key maps) --- */
typedef struct orc_store orc_store;
typedef struct orc_model orc_model;

enum { ORC_DNN = 0, ORC_WIDEDEEP = 1 };
enum { ORC_GRAD_COMPAT = 0, ORC_GRAD_INTENDED = 1 };

orc_store *orc_store_new(uint64_t seed);
void orc_store_free(orc_store *);
/* store/KVStore.java:129-159 (standalone).  Returns live pointer or NULL. */
float *orc_store_get(orc_store *, const char *key, int *rows, int *cols);
void orc_store_put(orc_store *, const char *key, const float *data, int rows, int cols);
int orc_store_size(orc_store *);
/* optimizer-state access for tests: which = 0:M 1:V (Adam) 2:Z 3:N (Ftrl) */
float *orc_store_state(orc_store *, const char *key, int which, int *len);

/* model/DNN.java:92-128, model/WideDeepNN.java:105-161 */
orc_model *orc_model_new(orc_store *st, int kind, int F, int D, int X,
                         int nfc, const int *fc_dims, int wide_size);
void orc_model_free(orc_model *);
void orc_model_set_grad_mode(orc_model *, int emb_mode /*ORC_GRAD_*/, int wide_mode, int chunk);
/* Make every key this batch touches exist (same effect as the lazy
 * kvStore.get(key, init) calls of the forward pass); used by tests that
 * then read rows out to seed the GPU tables. */
/* train/TrainerThread.java:29-39 + train/Trainer.java:70-101 for thread=1:
 * pullWeights; train (fwd, loss, bwd); KVStore.update(updaters); clear.
 * E [B][F] float ids, Xd [B][X], Wd [B][F] float wide ids (NULL for DNN),
 * Y [B].  Returns the loss.  do_update=0 stops before KVStore.update so the
 * per-key gradients can be inspected. */
float orc_model_train(orc_model *, const float *E, const float *Xd, const float *Wd,
                      const float *Y, int B, int do_update);
/* model/DNN.java:78-90 : forward only; P [B] */
void orc_model_predict(orc_model *, const float *E, const float *Xd, const float *Wd,
                       int B, float *P);
void orc_model_apply_update(orc_model *);   /* KVStore.update(Map) + clear, after do_update=0 */
/* intermediates of the last train(): layer 0 = embedding A [B][F*D],
 * 1 = concat A, 2.. = fc_i A ; "delta" = the delta each layer produced
 * (fc_i: W^T delta, [B][in]).  wide logit / final P via the named getters. */
const float *orc_model_act(orc_model *, int layer, int *rows, int *cols);
const float *orc_model_delta(orc_model *, int layer, int *rows, int *cols);
const float *orc_model_wide_logit(orc_model *, int *B);
const float *orc_model_p(orc_model *, int *B);
/* gradient handed to the updater for key in the last step (after /cnt) */
const float *orc_model_grad(orc_model *, const char *key, int *len);
int orc_model_num_grad_keys(orc_model *);
const char *orc_model_grad_key(orc_model *, int i);

/* ---- PS semantics (net/PServer.java:164-283, net/PSRouterClient.java) --- */
/* N shards x W workers, BSP: each worker pushes its per-key gradient; the
 * owner shard averages over the workers that pushed the key and applies one
 * updater step per key per global step (SURVEY App. A.9 intended semantics;
 * the never-cleared sum map and the broken barrier test are NOT copied). */
typedef struct orc_ps orc_ps;
orc_ps *orc_ps_new(int nshards, uint64_t seed, int floor_mod);
void orc_ps_free(orc_ps *);
orc_store *orc_ps_shard(orc_ps *, int shard);
int orc_ps_route(orc_ps *, const char *key);
/* worker-side: push g for key with updater name ("adam@..." / "adam@alfa..beta:" ftrl) */
int orc_ps_push(orc_ps *, const char *key, const float *g, int len, const char *updater_name, int is_async);
/* barrier of all workers reached: psUpdate + globalStep++ */
void orc_ps_barrier_update(orc_ps *);
long orc_ps_global_step(orc_ps *);

#ifdef __cplusplus
}
#endif
#endif
