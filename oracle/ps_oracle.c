/*
 * ps_oracle.c -- CPU restatement ("oracle") of the wudikua/ps hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ps_oracle.h).  PARITY UNPINNED: no reference
 * run and no reference golden vectors exist for this path; pinned by the
 * hand-derived KATs of SURVEY.md App. B and by the independent numpy twin.
 *
 * Build with -ffp-contract=off: every Java float op is individually rounded
 * (the JVM never fuses), and so is every op here.
 *
 * Citations: /root/reference/src/main/java/<file>:<line>.
 */
#include "ps_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== */
/* strings                                                               */
/* ===================================================================== */

int32_t orc_java_hashcode(const char *s) {
    /* String.hashCode: h = 31*h + c over UTF-16 units; keys here are ASCII. */
    uint32_t h = 0;
    for (; *s; ++s) h = 31u * h + (uint32_t)(unsigned char)*s;
    return (int32_t)h;
}

int orc_float_to_string(float v, char *buf, int cap) {
    /* Float.toString: shortest decimal digits that round-trip to the same
     * float; plain notation for 1e-3 <= |v| < 1e7 with at least one digit
     * after the point, otherwise computerized scientific notation. */
    if (v != v) return snprintf(buf, cap, "NaN");
    if (isinf(v)) return snprintf(buf, cap, v > 0 ? "Infinity" : "-Infinity");
    if (v == 0.0f) return snprintf(buf, cap, signbit(v) ? "-0.0" : "0.0");
    char digs[32];
    int prec;
    for (prec = 1; prec <= 9; ++prec) {
        snprintf(digs, sizeof digs, "%.*e", prec - 1, (double)v);
        if (strtof(digs, NULL) == v) break;
    }
    /* digs = [-]d.ddde[+-]xx */
    char mant[16];
    int nm = 0, neg = 0;
    const char *p = digs;
    if (*p == '-') { neg = 1; ++p; }
    for (; *p && *p != 'e'; ++p)
        if (*p != '.') mant[nm++] = *p;
    mant[nm] = 0;
    int ex = atoi(p + 1); /* value = 0.d1d2.. * 10^(ex+1) */
    while (nm > 1 && mant[nm - 1] == '0') mant[--nm] = 0;
    char out[64];
    int o = 0;
    if (neg) out[o++] = '-';
    float a = fabsf(v);
    if (a >= 1e-3f && a < 1e7f) {
        if (ex >= 0) {
            for (int i = 0; i <= ex; ++i) out[o++] = i < nm ? mant[i] : '0';
            out[o++] = '.';
            if (nm > ex + 1) for (int i = ex + 1; i < nm; ++i) out[o++] = mant[i];
            else out[o++] = '0';
        } else {
            out[o++] = '0'; out[o++] = '.';
            for (int i = 0; i < -ex - 1; ++i) out[o++] = '0';
            for (int i = 0; i < nm; ++i) out[o++] = mant[i];
        }
        out[o] = 0;
    } else {
        out[o++] = mant[0]; out[o++] = '.';
        if (nm > 1) for (int i = 1; i < nm; ++i) out[o++] = mant[i];
        else out[o++] = '0';
        o += snprintf(out + o, sizeof out - o, "E%d", ex);
    }
    return snprintf(buf, cap, "%s", out);
}

int orc_emb_key(int field, float id, char *buf, int cap) {
    /* layer/EmbeddingLayer.java:52 ("emF"+j) + layer/EmbeddingField.java:71 */
    char f[48];
    orc_float_to_string(id, f, sizeof f);
    return snprintf(buf, cap, "emF%d.%s", field, f);
}

int orc_wide_key(float id, char *buf, int cap) {
    /* layer/LRLayer.java:78: this.name + ".weights." + index */
    char f[48];
    orc_float_to_string(id, f, sizeof f);
    return snprintf(buf, cap, "wide.weights.%s", f);
}

int orc_mod_shard(const char *key, int n, int floor_mod) {
    /* net/Mod.java:13-15.  Java % truncates toward zero. */
    int32_t h = orc_java_hashcode(key);
    int r = (int)(h % n);
    if (floor_mod && r < 0) r += n;
    return r;
}

float orc_matrixutil_hash(float id, int size) {
    /* util/MatrixUtil.java:27-33: result.data[i] % size  (float % int -> fmodf) */
    return fmodf(id, (float)size);
}

/* ===================================================================== */
/* init                                                                  */
/* ===================================================================== */

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

float orc_init_value(uint64_t seed, uint64_t table, uint64_t row, uint64_t col, float scale) {
    /* +-RandomUtils.nextFloat(0, scale) with a fair-coin sign
     * (util/MatrixUtil.java:62-74), made a pure function of the key. */
    uint64_t h = splitmix64(seed + 0x9E3779B97F4A7C15ull * (table + 1));
    h = splitmix64(h ^ row);
    h = splitmix64(h ^ (col * 0xD6E8FEB86659FD93ull));
    float u = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
    float m = u * scale;
    return ((h >> 39) & 1) ? -m : m;
}

float orc_xavier_scale(int in_dims, int out_dims) {
    /* layer/EmbeddingField.java:40, layer/FcLayer.java:39,46 */
    return (float)(4 * (sqrt(6.0) / sqrt((double)(in_dims + out_dims))));
}

/* ===================================================================== */
/* arithmetic                                                            */
/* ===================================================================== */

float orc_sigmoid_clip(float x) {
    /* activations/Sigmoid.java:11 -- float constants, double exp, cast */
    return (float)(0.001f + (double)(.999f - 0.001f) / (1.0 + exp(-(double)x)));
}

static float sqrt_jf(float x) { return (float)sqrt((double)x); } /* MatrixFunctions.sqrt */

void orc_adam_update(float *w, const float *g, float *M, float *V, int n,
                     float alfa, float beta1, float beta2, float eps) {
    /* update/AdamUpdater.java:57-70 */
    const float c1 = 1 - beta1, c2 = 1 - beta2; /* float arithmetic, as in Java */
    const float na = -1 * alfa;
    for (int i = 0; i < n; ++i) {
        float m = g[i] * c1;             /* dw.mul(1-beta1)            :61 */
        float mo = M[i] * beta1;         /* M.get(key).muli(beta1)         */
        m = mo + m;                      /* addi  (saxpy 1.0)              */
        float v = g[i] * g[i];           /* dw.mul(dw)                 :62 */
        v = v * c2;                      /* muli(1-beta2)                  */
        float vo = V[i] * beta2;
        v = vo + v;
        M[i] = m; V[i] = v;
        float mm = m / c1;               /* M.div(1-beta1)             :63 */
        float vv = v / c2;               /* V.div(1-beta2)             :64 */
        float den = sqrt_jf(vv) + eps;   /* sqrt(Vv).addi(epsilon)     :69 */
        float q = mm / den;
        q = q * na;                      /* muli(-1*alfa)                  */
        w[i] = q + w[i];                 /* w.addi(...)                    */
    }
}

int orc_ftrl_update(float *w, const float *g, float *z, float *nn, int n,
                    float alfa, float beta, float l1, float l2) {
    /* update/FtrlUpdater.java:51-76 */
    if (g[0] == 0) return 0;                                      /* :52-54 */
    for (int i = 0; i < n; ++i) {                                 /* :64-71 */
        if (fabsf(z[i]) <= l1) {
            w[i] = 0;
        } else {
            float sign = z[i] >= 0 ? 1.f : -1.f;
            float den = (l2 + (beta + sqrt_jf(nn[i]))) / alfa;
            w[i] = -(z[i] - sign * l1) / den;
        }
    }
    for (int i = 0; i < n; ++i) {                                 /* :72-74 */
        float g2 = (float)((double)g[i] * (double)g[i]);          /* MatrixFunctions.pow(dw,2) */
        float s = sqrt_jf(nn[i] + g2) - sqrt_jf(nn[i] / alfa);
        float t = g[i] - s * w[i];
        z[i] = t + z[i];
        nn[i] = g2 + nn[i];
    }
    return 1;
}

/* Sum of the chunk partials [j0, j1) of a run, folded in chunk order (p_j0, then p + acc). */
static void fold_chunks(const float *gk, int n, int D, int chunk, int j0, int j1, float *acc) {
    for (int j = j0; j < j1; ++j) {
        int c0 = j * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
        for (int d = 0; d < D; ++d) {
            float p = gk[(size_t)c0 * D + d];
            for (int k = c0 + 1; k < c1; ++k) p = gk[(size_t)k * D + d] + p;
            acc[d] = (j == j0) ? p : p + acc[d];
        }
    }
}

/* Order of the f32 adds over a run of n per-sample gradients (SURVEY App. A.6 keeps the reference's strictly
 * sequential order only for n <= chunk -- a hot key's run cannot be summed by one thread at speed):
 *   n <= chunk                 : g_1, then g_k + acc, in batch order (the reference's order)
 *   chunk < n <= 128 * chunk   : chunk partials (each in batch order), folded in chunk order
 *   n > 128 * chunk            : chunk partials folded 32 at a time into super partials, those folded in order
 * The same folding is used for pass 1 (S) and for pass 2 (S/n + ...). */
#define ORC_SUPER 32
#define ORC_SUPER_MIN_CHUNKS 128
static void run_fold(const float *gk, int n, int D, int chunk, float *S, int have) {
    float *t = (float *)calloc((size_t)D, sizeof(float));
    if (chunk <= 0 || n <= chunk) {
        for (int k = 0; k < n; ++k)
            for (int d = 0; d < D; ++d) S[d] = (!have && k == 0) ? gk[d] : gk[(size_t)k * D + d] + S[d];   /* put :91 / addi :94 */
    } else {
        int nch = (n + chunk - 1) / chunk;
        int step = nch > ORC_SUPER_MIN_CHUNKS ? ORC_SUPER : 1;
        for (int j = 0; j < nch; j += step) {
            fold_chunks(gk, n, D, chunk, j, j + step < nch ? j + step : nch, t);
            for (int d = 0; d < D; ++d) S[d] = (!have && j == 0) ? t[d] : t[d] + S[d];
        }
    }
    free(t);
}

void orc_emb_geff(const float *gk, int n, int D, float *out, int mode, int chunk) {
    /* SURVEY App. A.6.  S = sum of the n per-sample gradients. */
    float *S = (float *)calloc((size_t)D, sizeof(float));
    run_fold(gk, n, D, chunk, S, 0);
    if (mode == ORC_GRAD_INTENDED) {
        for (int d = 0; d < D; ++d) out[d] = S[d] / (float)n;
        free(S);
        return;
    }
    /* pass 1 end: G = S / n (divi :100); sum.put(key, G) by reference, cnt=1 */
    for (int d = 0; d < D; ++d) S[d] = S[d] / (float)n;
    /* pass 2: G += g_k for every k again (:94), N = 2n */
    run_fold(gk, n, D, chunk, S, 1);
    /* G /= 2n (:100); sum[key].addi(G) on itself = 2G, cnt=2 (KVStore.java:197);
     * update divides by cnt (KVStore.java:253): (2G)/2 == G exactly. */
    for (int d = 0; d < D; ++d) {
        float G = S[d] / (float)(2 * n);
        G = G + G;
        out[d] = G / 2.0f;
    }
    free(S);
}

float orc_ce_forward(const float *p, const float *y, int B) {
    /* loss/CrossEntropy.java:10-18 (FastMath.log ~ log; double math, f32 sum) */
    float sum = 0;
    for (int i = 0; i < B; ++i) {
        float pi = p[i], l = y[i];
        sum += (float)(-l * log((double)pi) - ((1 - l) * log((double)(1 - pi))));
    }
    return sum / B;
}

void orc_ce_backward(const float *p, const float *y, int B, float *delta) {
    /* loss/CrossEntropy.java:20-28 */
    for (int i = 0; i < B; ++i) delta[i] = (p[i] - y[i]) / (p[i] * (1 - p[i]));
}

void orc_sgemm_nn(int M, int N, int K, const float *A, const float *Bm, float *C) {
    /* column-major C = A*B as jblas' sgemm('N','N',1,..,0,..); the native BLAS
     * summation order is unknowable -- sequential k is the restatement. */
    for (int j = 0; j < N; ++j) {
        float *c = C + (size_t)j * M;
        for (int i = 0; i < M; ++i) c[i] = 0;
        for (int k = 0; k < K; ++k) {
            const float b = Bm[(size_t)j * K + k];
            const float *a = A + (size_t)k * M;
            for (int i = 0; i < M; ++i) c[i] += a[i] * b;
        }
    }
}

/* ===================================================================== */
/* string-keyed map (the reference's HashMap<String, FloatMatrix>)       */
/* ===================================================================== */

typedef struct {
    char *key;
    float *data;
    int rows, cols;
    long cnt;
    int owned;
} sent;

typedef struct {
    sent *e;
    int *slot; /* open addressing -> index into e, -1 empty */
    int n, cap_e, cap_s;
} smap;

static void smap_init(smap *m) {
    m->n = 0; m->cap_e = 64; m->cap_s = 256;
    m->e = (sent *)malloc(sizeof(sent) * m->cap_e);
    m->slot = (int *)malloc(sizeof(int) * m->cap_s);
    for (int i = 0; i < m->cap_s; ++i) m->slot[i] = -1;
}
static uint32_t smap_h(const char *k) {
    uint32_t h = (uint32_t)orc_java_hashcode(k);
    return h ^ (h >> 16); /* HashMap.hash */
}
static int smap_find(const smap *m, const char *k) {
    uint32_t i = smap_h(k) & (uint32_t)(m->cap_s - 1);
    for (;;) {
        int s = m->slot[i];
        if (s < 0) return -1;
        if (strcmp(m->e[s].key, k) == 0) return s;
        i = (i + 1) & (uint32_t)(m->cap_s - 1);
    }
}
static void smap_rehash(smap *m) {
    m->cap_s *= 2;
    m->slot = (int *)realloc(m->slot, sizeof(int) * m->cap_s);
    for (int i = 0; i < m->cap_s; ++i) m->slot[i] = -1;
    for (int s = 0; s < m->n; ++s) {
        uint32_t i = smap_h(m->e[s].key) & (uint32_t)(m->cap_s - 1);
        while (m->slot[i] >= 0) i = (i + 1) & (uint32_t)(m->cap_s - 1);
        m->slot[i] = s;
    }
}
static sent *smap_put(smap *m, const char *k) {
    int s = smap_find(m, k);
    if (s >= 0) return &m->e[s];
    if (m->n * 2 >= m->cap_s) smap_rehash(m);
    if (m->n == m->cap_e) { m->cap_e *= 2; m->e = (sent *)realloc(m->e, sizeof(sent) * m->cap_e); }
    s = m->n++;
    memset(&m->e[s], 0, sizeof(sent));
    m->e[s].key = strdup(k);
    uint32_t i = smap_h(k) & (uint32_t)(m->cap_s - 1);
    while (m->slot[i] >= 0) i = (i + 1) & (uint32_t)(m->cap_s - 1);
    m->slot[i] = s;
    return &m->e[s];
}
static sent *smap_get(const smap *m, const char *k) {
    int s = smap_find(m, k);
    return s < 0 ? NULL : &m->e[s];
}
static void smap_clear(smap *m) {
    for (int s = 0; s < m->n; ++s) {
        free(m->e[s].key);
        if (m->e[s].owned) free(m->e[s].data);
    }
    m->n = 0;
    for (int i = 0; i < m->cap_s; ++i) m->slot[i] = -1;
}
static void smap_free(smap *m) { smap_clear(m); free(m->e); free(m->slot); }

static float *fdup(const float *src, int n) {
    float *d = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
    memcpy(d, src, sizeof(float) * n);
    return d;
}

/* ===================================================================== */
/* KVStore (standalone)  store/KVStore.java                              */
/* ===================================================================== */

struct orc_store {
    uint64_t seed;
    smap store;      /* :40  key -> weights (owned) */
    smap sum;        /* :51-52  key -> (ptr BY REFERENCE, cnt) */
    smap M, V, Z, N; /* per-key updater state (update/AdamUpdater.java:38-39, FtrlUpdater.java:32-33) */
    /* updater hyper-parameters (model/DNN.java:95, model/WideDeepNN.java:109-113) */
    float a_alfa, a_b1, a_b2, a_eps;
    float f_alfa, f_beta, f_l1, f_l2;
    int wide_ftrl; /* "wide.weights" / "wide.bias" -> ftrl (WideDeepNN only) */
};

orc_store *orc_store_new(uint64_t seed) {
    orc_store *s = (orc_store *)calloc(1, sizeof *s);
    s->seed = seed;
    smap_init(&s->store); smap_init(&s->sum);
    smap_init(&s->M); smap_init(&s->V); smap_init(&s->Z); smap_init(&s->N);
    s->a_alfa = (float)0.005; s->a_b1 = (float)0.9; s->a_b2 = (float)0.999;
    s->a_eps = (float)pow(10, -8);
    s->f_alfa = 0.005f; s->f_beta = 1.0f; s->f_l1 = 0.001f; s->f_l2 = 0.001f;
    return s;
}
void orc_store_free(orc_store *s) {
    if (!s) return;
    smap_free(&s->store); smap_free(&s->sum);
    smap_free(&s->M); smap_free(&s->V); smap_free(&s->Z); smap_free(&s->N);
    free(s);
}
float *orc_store_get(orc_store *s, const char *key, int *rows, int *cols) {
    sent *e = smap_get(&s->store, key);
    if (!e) return NULL;
    if (rows) *rows = e->rows;
    if (cols) *cols = e->cols;
    return e->data;
}
void orc_store_put(orc_store *s, const char *key, const float *data, int rows, int cols) {
    sent *e = smap_put(&s->store, key);
    if (e->owned) free(e->data);
    e->data = fdup(data, rows * cols);
    e->rows = rows; e->cols = cols; e->owned = 1;
}
int orc_store_size(orc_store *s) { return s->store.n; }
float *orc_store_state(orc_store *s, const char *key, int which, int *len) {
    smap *m = which == 0 ? &s->M : which == 1 ? &s->V : which == 2 ? &s->Z : &s->N;
    sent *e = smap_get(m, key);
    if (!e) return NULL;
    if (len) *len = e->rows * e->cols;
    return e->data;
}

/* kvStore.get(key, init) standalone branch (:151-158, create :168-189) */
static sent *store_get_init(orc_store *s, const char *key, int rows, int cols,
                            uint64_t table, uint64_t row0, float scale, int zero) {
    sent *e = smap_get(&s->store, key);
    if (e) return e;
    e = smap_put(&s->store, key);
    e->rows = rows; e->cols = cols; e->owned = 1;
    e->data = (float *)calloc((size_t)rows * cols, sizeof(float));
    if (!zero) {
        if (table < ORC_TABLE_WIDE) /* embedding row: (row=id, col=d) */
            for (int d = 0; d < rows * cols; ++d) e->data[d] = orc_init_value(s->seed, table, row0, (uint64_t)d, scale);
        else /* dense tensor: (row = flat index, col = 0) */
            for (int d = 0; d < rows * cols; ++d) e->data[d] = orc_init_value(s->seed, table, (uint64_t)d, 0, scale);
    }
    return e;
}

/* KVStore.sum :192-200 -- first touch keeps the caller's matrix BY REFERENCE */
static void store_sum(orc_store *s, const char *key, float *val, int len) {
    sent *e = smap_get(&s->sum, key);
    if (!e) {
        e = smap_put(&s->sum, key);
        e->data = val; e->rows = len; e->cols = 1; e->cnt = 1; e->owned = 0;
    } else {
        float *y = e->data;
        for (int i = 0; i < len; ++i) y[i] = val[i] + y[i]; /* saxpy(1.0, val, sum); val may alias */
        e->cnt++;
    }
}

static sent *state_get(smap *m, const char *key, int len) {
    sent *e = smap_get(m, key);
    if (!e) {
        e = smap_put(m, key);
        e->data = (float *)calloc((size_t)len, sizeof(float));
        e->rows = len; e->cols = 1; e->owned = 1;
    }
    return e;
}

/* updater.update(key, store.get(key), g) with the updater chosen as in
 * KVStore.update(Map) :241-252: exact key, else prefix, else "default". */
static int store_apply(orc_store *s, const char *key, const float *g, int len, int force_updater /*-1 auto,0 adam,1 ftrl*/) {
    sent *w = smap_get(&s->store, key);
    if (!w) return -1;
    int ftrl = 0;
    if (force_updater >= 0) ftrl = force_updater;
    else if (s->wide_ftrl && (strncmp(key, "wide.weights", 12) == 0 || strcmp(key, "wide.bias") == 0)) ftrl = 1;
    if (ftrl) {
        if (g[0] == 0) return 0;
        sent *z = state_get(&s->Z, key, len), *n = state_get(&s->N, key, len);
        orc_ftrl_update(w->data, g, z->data, n->data, len, s->f_alfa, s->f_beta, s->f_l1, s->f_l2);
    } else {
        sent *m = state_get(&s->M, key, len), *v = state_get(&s->V, key, len);
        orc_adam_update(w->data, g, m->data, v->data, len, s->a_alfa, s->a_b1, s->a_b2, s->a_eps);
    }
    return 1;
}

/* ===================================================================== */
/* layers / model                                                        */
/* ===================================================================== */

typedef struct {
    int in, out;
    int act;            /* 0 none, 1 relu, 2 sigmoid */
    char wkey[32], bkey[32];
    float *W, *b;       /* live references into the store (pullWeights) */
    float *A;           /* out x B, Z aliased (in-place activation) */
    float *delta;       /* in x B */
    float *dW, *db;
} fc_layer;

struct orc_model {
    orc_store *st;
    int kind, F, D, X, nfc, wide_size;
    fc_layer *fc;
    int emb_mode, wide_mode, chunk;
    /* per-field maps (layer/EmbeddingField.java:25-27) */
    smap *f_weights, *f_grad; /* f_grad: data = gradient buffer, cnt = N */
    smap wide_weights;        /* layer/LRLayer.java:26 -- never cleared by the reference */
    int B;
    float *E;                 /* copy of ids [B][F] */
    float *Wd;                /* copy of wide ids [B][F] */
    float *embA;              /* (F*D) x B */
    float *concatA;           /* (F*D+X) x B */
    float *wideZ, *addA, *P, *lossdelta, *wide_gbar;
    /* gradient snapshot of the last step */
    smap grads;
    int pending;
};

orc_model *orc_model_new(orc_store *st, int kind, int F, int D, int X,
                         int nfc, const int *fc_dims, int wide_size) {
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    m->st = st; m->kind = kind; m->F = F; m->D = D; m->X = X; m->nfc = nfc; m->wide_size = wide_size;
    m->fc = (fc_layer *)calloc((size_t)nfc, sizeof(fc_layer));
    int in = F * D + X;
    for (int i = 0; i < nfc; ++i) {
        /* layer/FcLayer.java:53-70: last layer sigmoid, others relu;
         * model/WideDeepNN.java:128: last deep layer activation = null */
        m->fc[i].in = in; m->fc[i].out = fc_dims[i];
        m->fc[i].act = (i == nfc - 1) ? (kind == ORC_WIDEDEEP ? 0 : 2) : 1;
        snprintf(m->fc[i].wkey, sizeof m->fc[i].wkey, "fc%d.weights", i);
        snprintf(m->fc[i].bkey, sizeof m->fc[i].bkey, "fc%d.bias", i);
        in = fc_dims[i];
    }
    m->f_weights = (smap *)malloc(sizeof(smap) * F);
    m->f_grad = (smap *)malloc(sizeof(smap) * F);
    for (int f = 0; f < F; ++f) { smap_init(&m->f_weights[f]); smap_init(&m->f_grad[f]); }
    smap_init(&m->wide_weights);
    smap_init(&m->grads);
    if (kind == ORC_WIDEDEEP) {
        st->wide_ftrl = 1;
        /* layer/LRLayer.java:37-52: bias fetched in the constructor, zeros(1) */
        store_get_init(st, "wide.bias", 1, 1, ORC_TABLE_WIDE_B, 0, 0.f, 1);
    }
    return m;
}

static void model_free_step(orc_model *m) {
    for (int i = 0; i < m->nfc; ++i) {
        free(m->fc[i].A); free(m->fc[i].delta); free(m->fc[i].dW); free(m->fc[i].db);
        m->fc[i].A = m->fc[i].delta = m->fc[i].dW = m->fc[i].db = NULL;
    }
    free(m->E); free(m->Wd); m->Wd = NULL; free(m->embA); free(m->concatA); free(m->wideZ); free(m->addA);
    free(m->lossdelta); free(m->wide_gbar);
    m->E = m->embA = m->concatA = m->wideZ = m->addA = m->lossdelta = m->wide_gbar = NULL;
    m->P = NULL;
}

void orc_model_free(orc_model *m) {
    if (!m) return;
    model_free_step(m);
    for (int f = 0; f < m->F; ++f) {
        smap_free(&m->f_weights[f]);
        for (int s = 0; s < m->f_grad[f].n; ++s) m->f_grad[f].e[s].owned = 1;
        smap_free(&m->f_grad[f]);
    }
    smap_free(&m->wide_weights);
    smap_free(&m->grads);
    free(m->f_weights); free(m->f_grad); free(m->fc); free(m);
}

void orc_model_set_grad_mode(orc_model *m, int emb_mode, int wide_mode, int chunk) {
    m->emb_mode = emb_mode; m->wide_mode = wide_mode; m->chunk = chunk;
}

static void relu_fwd(float *x, size_t n) {          /* activations/Relu.java:7-12 */
    for (size_t i = 0; i < n; ++i) x[i] = x[i] != x[i] ? x[i] : (x[i] > 0 ? x[i] : 0); /* Math.max(0, x): -0 -> +0, NaN stays */
}
static void relu_bwd(float *dy, const float *y, size_t n) { /* activations/Relu.java:14-19 */
    for (size_t i = 0; i < n; ++i) dy[i] *= y[i] > 0 ? 1.f : 0.f;
}
static void sigmoid_bwd(float *dy, const float *y, size_t n) { /* activations/Sigmoid.java:16-21 */
    for (size_t i = 0; i < n; ++i) dy[i] *= y[i] * (1 - y[i]);
}

/* Model.pullWeights (model/DNN.java:72-76) */
static void pull_weights(orc_model *m) {
    /* layer/EmbeddingLayer.java:71-75 -> EmbeddingField.clear :80-84 */
    for (int f = 0; f < m->F; ++f) {
        smap_clear(&m->f_weights[f]);
        for (int s = 0; s < m->f_grad[f].n; ++s) m->f_grad[f].e[s].owned = 1; /* buffers die with the map */
        smap_clear(&m->f_grad[f]);
    }
    /* layer/FcLayer.java:112-115 */
    for (int i = 0; i < m->nfc; ++i) {
        fc_layer *l = &m->fc[i];
        l->W = store_get_init(m->st, l->wkey, l->out, l->in, ORC_TABLE_FC(i), 0,
                              orc_xavier_scale(l->in, l->out), 0)->data;
        l->b = store_get_init(m->st, l->bkey, l->out, 1, ORC_TABLE_FC(i) + 1, 0,
                              orc_xavier_scale(l->in, 1), 0)->data;
    }
    /* layer/LRLayer.java:122-124: bias only; LRLayer.weights is NOT cleared */
}

static void forward(orc_model *m, const float *E, const float *Xd, const float *Wd, int B) {
    const int F = m->F, D = m->D, X = m->X, FD = F * D, C = FD + X;
    model_free_step(m);
    m->B = B;
    m->E = fdup(E, B * F);
    if (Wd) m->Wd = fdup(Wd, B * F);
    /* ---- EmbeddingLayer.forward  layer/EmbeddingLayer.java:25-48 ---- */
    m->embA = (float *)calloc((size_t)FD * B, sizeof(float));
    const float xav = orc_xavier_scale(1, D);
    char key[96];
    for (int f = 0; f < F; ++f) {
        /* EmbeddingField.forward  layer/EmbeddingField.java:66-78 */
        for (int i = 0; i < B; ++i) {
            float sample = E[(size_t)i * F + f];          /* E.getRow(f)[i] */
            orc_emb_key(f, sample, key, sizeof key);      /* :71 */
            sent *c = smap_get(&m->f_weights[f], key);    /* checkExists :49-54 */
            if (!c) {
                sent *w = store_get_init(m->st, key, D, 1, (uint64_t)f, (uint64_t)(int64_t)sample, xav, 0);
                c = smap_put(&m->f_weights[f], key);
                c->data = w->data; c->rows = D; c->cols = 1; c->owned = 0;
            }
            memcpy(&m->embA[(size_t)i * FD + (size_t)f * D], c->data, sizeof(float) * D); /* rcopy :73 */
        }
    }
    relu_fwd(m->embA, (size_t)FD * B);                    /* :76, per field; same thing */
    /* ---- ConcatLayer.forward  layer/ConcatLayer.java:30-37 ---- */
    m->concatA = (float *)malloc(sizeof(float) * (size_t)C * B);
    for (int i = 0; i < B; ++i) {
        memcpy(&m->concatA[(size_t)i * C], &m->embA[(size_t)i * FD], sizeof(float) * FD);
        memcpy(&m->concatA[(size_t)i * C + FD], &Xd[(size_t)i * X], sizeof(float) * X);
    }
    /* ---- FcLayer.forward  layer/FcLayer.java:74-91 ---- */
    const float *A = m->concatA;
    for (int l = 0; l < m->nfc; ++l) {
        fc_layer *L = &m->fc[l];
        L->A = (float *)malloc(sizeof(float) * (size_t)L->out * B);
        orc_sgemm_nn(L->out, B, L->in, L->W, A, L->A);                     /* weights.mmul(A) :76 */
        for (int c = 0; c < B; ++c)
            for (int r = 0; r < L->out; ++r) L->A[(size_t)c * L->out + r] += L->b[r]; /* addiColumnVector :77 */
        if (L->act == 1) relu_fwd(L->A, (size_t)L->out * B);
        else if (L->act == 2)
            for (size_t i = 0; i < (size_t)L->out * B; ++i) L->A[i] = orc_sigmoid_clip(L->A[i]);
        A = L->A;
    }
    m->P = m->fc[m->nfc - 1].A;
    if (m->kind == ORC_WIDEDEEP) {
        /* ---- LRLayer.forward  layer/LRLayer.java:62-98 ---- */
        /* the matrix itself is stable; entry pointers are not (the map grows below) */
        const float *biasd = smap_get(&m->st->store, "wide.bias")->data;
        m->wideZ = (float *)malloc(sizeof(float) * B);
        for (int i = 0; i < B; ++i) {
            float sumW = 0.f;
            for (int j = 0; j < F; ++j) {
                float index = Wd[(size_t)i * F + j];
                orc_wide_key(index, key, sizeof key);
                sent *wi = store_get_init(m->st, key, 1, 1, ORC_TABLE_WIDE, 0, 0.f, 1); /* zeros(1) :39-44 */
                if (!smap_get(&m->wide_weights, key)) {                          /* weights.put :79 */
                    sent *c = smap_put(&m->wide_weights, key);
                    c->data = wi->data; c->rows = 1; c->cols = 1; c->owned = 0;
                }
                sumW += wi->data[0];
            }
            m->wideZ[i] = sumW;
        }
        for (int i = 0; i < B; ++i) m->wideZ[i] += biasd[0];              /* addiColumnVector :84 */
        /* ---- AddLayer.forward  layer/AddLayer.java:33-48 ---- */
        m->addA = (float *)malloc(sizeof(float) * B);
        for (int i = 0; i < B; ++i) m->addA[i] = orc_sigmoid_clip(m->fc[m->nfc - 1].A[i] + m->wideZ[i]);
        m->P = m->addA;
    }
}

/* EmbeddingField.backward  layer/EmbeddingField.java:86-104, one pass over
 * all fields (EmbeddingLayer.backward  layer/EmbeddingLayer.java:59-69). */
static void emb_backward_pass(orc_model *m, const float *delta /* C x B */) {
    const int F = m->F, D = m->D, FD = F * D, C = FD + m->X, B = m->B;
    char key[96];
    float *g = (float *)malloc(sizeof(float) * D);
    for (int f = 0; f < F; ++f) {
        smap *G = &m->f_grad[f];
        const int off = f * D;
        for (int k = 0; k < B; ++k) {
            orc_emb_key(f, m->E[(size_t)k * F + f], key, sizeof key);  /* via double: same text for integer ids */
            for (int d = 0; d < D; ++d) g[d] = delta[(size_t)k * C + off + d];  /* getRange copy */
            relu_bwd(g, &m->embA[(size_t)k * FD + off], (size_t)D);             /* activation.backward(.., A.getColumn(k)) */
            sent *e = smap_get(G, key);
            if (!e) {
                e = smap_put(G, key);
                e->data = fdup(g, D); e->rows = D; e->cols = 1; e->cnt = 1; e->owned = 0; /* :91-92 */
            } else {
                for (int d = 0; d < D; ++d) e->data[d] = g[d] + e->data[d];     /* addi :94 */
                e->cnt++;
            }
        }
        for (int s = 0; s < G->n; ++s) {                                        /* :99-102 */
            sent *e = &G->e[s];
            const float n = (float)(int)e->cnt;
            for (int d = 0; d < D; ++d) e->data[d] = e->data[d] / n;            /* divi(N) */
            store_sum(m->st, e->key, e->data, D);                               /* BY REFERENCE */
        }
    }
    free(g);
}

/* "intended" variant: one pass, mean over occurrences (App. A.6 switch) */
static void emb_backward_modes(orc_model *m, const float *delta) {
    if (m->emb_mode == ORC_GRAD_COMPAT && m->chunk <= 0) {
        emb_backward_pass(m, delta);  /* ConcatLayer.backward -> embedding.backward()  layer/ConcatLayer.java:42-46 */
        emb_backward_pass(m, delta);  /* model loop, layers[0]  model/DNN.java:66-68 */
        return;
    }
    /* array form through orc_emb_geff (same arithmetic; used for the
     * intended / chunked-order switches) */
    const int F = m->F, D = m->D, FD = F * D, C = FD + m->X, B = m->B;
    char key[96];
    for (int f = 0; f < F; ++f) {
        smap seen; smap_init(&seen);
        for (int k = 0; k < B; ++k) {
            orc_emb_key(f, m->E[(size_t)k * F + f], key, sizeof key);
            if (smap_get(&seen, key)) continue;
            smap_put(&seen, key);
            float *gk = (float *)malloc(sizeof(float) * (size_t)D * B);
            int n = 0;
            for (int k2 = k; k2 < B; ++k2) {
                if (m->E[(size_t)k2 * F + f] != m->E[(size_t)k * F + f]) continue;
                for (int d = 0; d < D; ++d) gk[(size_t)n * D + d] = delta[(size_t)k2 * C + f * D + d];
                relu_bwd(&gk[(size_t)n * D], &m->embA[(size_t)k2 * FD + f * D], (size_t)D);
                ++n;
            }
            float *out = (float *)malloc(sizeof(float) * D);
            orc_emb_geff(gk, n, D, out, m->emb_mode, m->chunk);
            sent *e = smap_put(&m->f_grad[f], key);
            e->data = out; e->rows = D; e->cols = 1; e->cnt = n; e->owned = 0;
            store_sum(m->st, key, out, D);
            free(gk);
        }
        smap_clear(&seen); smap_free(&seen);
    }
}

static void backward(orc_model *m, const float *Y) {
    const int B = m->B, F = m->F;
    /* loss.backward  loss/CrossEntropy.java:20-28 */
    m->lossdelta = (float *)malloc(sizeof(float) * B);
    orc_ce_backward(m->P, Y, B, m->lossdelta);
    float *delta = m->lossdelta; /* 1 x B */
    if (m->kind == ORC_WIDEDEEP) {
        /* AddLayer.backward  layer/AddLayer.java:50-61 : sigmoid' in place */
        sigmoid_bwd(delta, m->addA, (size_t)B);
        /* LRLayer.backward  layer/LRLayer.java:100-120 */
        float s = 0.f;
        for (int c = 0; c < B; ++c) s = s + delta[c];    /* rowSums, sequential over columns */
        m->wide_gbar = (float *)malloc(sizeof(float));
        m->wide_gbar[0] = s / (float)B;                  /* rowMeans */
        store_sum(m->st, "wide.bias", m->wide_gbar, 1);
        if (m->wide_mode == ORC_GRAD_COMPAT) {
            /* :110-117: EVERY key this replica ever touched gets the same gbar */
            for (int k = 0; k < m->wide_weights.n; ++k) store_sum(m->st, m->wide_weights.e[k].key, m->wide_gbar, 1);
        } else {
            /* intended (not a reference path): g(key) = sum over the (sample, field)
             * occurrences of the key, in (sample, field) order, of delta / B */
            char key[96];
            for (int i = 0; i < B; ++i)
                for (int j = 0; j < F; ++j) {
                    orc_wide_key(m->Wd[(size_t)i * F + j], key, sizeof key);
                    sent *e = smap_get(&m->st->sum, key);
                    if (!e) {
                        float *z = (float *)calloc(1, sizeof(float));
                        store_sum(m->st, key, z, 1);
                        e = smap_get(&m->st->sum, key);
                        e->owned = 1;
                    }
                    e->data[0] = delta[i] + e->data[0];
                }
            for (int k = 0; k < m->st->sum.n; ++k)
                if (strncmp(m->st->sum.e[k].key, "wide.weights.", 13) == 0)
                    m->st->sum.e[k].data[0] = m->st->sum.e[k].data[0] / (float)B;
        }
    }
    /* FcLayer.backward  layer/FcLayer.java:93-110, last to first */
    for (int l = m->nfc - 1; l >= 0; --l) {
        fc_layer *L = &m->fc[l];
        const float *preA = l == 0 ? m->concatA : m->fc[l - 1].A;
        if (L->act == 1) relu_bwd(delta, L->A, (size_t)L->out * B);       /* activation.backward in place :100-102 */
        else if (L->act == 2) sigmoid_bwd(delta, L->A, (size_t)L->out * B);
        /* biasGradient = delta.rowMeans() :103 */
        L->db = (float *)calloc((size_t)L->out, sizeof(float));
        for (int c = 0; c < B; ++c)
            for (int r = 0; r < L->out; ++r) L->db[r] = L->db[r] + delta[(size_t)c * L->out + r];
        for (int r = 0; r < L->out; ++r) L->db[r] = L->db[r] / (float)B;
        store_sum(m->st, L->bkey, L->db, L->out);                           /* :104 */
        /* weightsGradient = delta.mmul(pre.A^T).divi(B) :105 ; dW[o,i] = sum_c delta[o,c]*preA[i,c] */
        L->dW = (float *)calloc((size_t)L->out * L->in, sizeof(float));
        for (int c = 0; c < B; ++c) {
            const float *dc = delta + (size_t)c * L->out;
            const float *ac = preA + (size_t)c * L->in;
            for (int i = 0; i < L->in; ++i) {
                const float a = ac[i];
                float *w = L->dW + (size_t)i * L->out;
                for (int o = 0; o < L->out; ++o) w[o] += dc[o] * a;
            }
        }
        for (size_t i = 0; i < (size_t)L->out * L->in; ++i) L->dW[i] = L->dW[i] / (float)B;
        store_sum(m->st, L->wkey, L->dW, L->out * L->in);                   /* :106 */
        /* this.delta = weights^T.mmul(delta) :108 ; [in x B] */
        L->delta = (float *)malloc(sizeof(float) * (size_t)L->in * B);
        for (int c = 0; c < B; ++c) {
            const float *dc = delta + (size_t)c * L->out;
            float *o = L->delta + (size_t)c * L->in;
            for (int i = 0; i < L->in; ++i) {
                const float *w = L->W + (size_t)i * L->out;
                float acc = 0;
                for (int k = 0; k < L->out; ++k) acc += w[k] * dc[k];
                o[i] = acc;
            }
        }
        delta = L->delta;
    }
    /* ConcatLayer.backward + EmbeddingLayer.backward (twice) */
    emb_backward_modes(m, m->fc[0].delta);
}

/* KVStore.update(Map) :240-261 + clear :270-277 */
static void snapshot_grads(orc_model *m) {
    orc_store *s = m->st;
    smap_clear(&m->grads);
    for (int k = 0; k < s->sum.n; ++k) {
        sent *e = &s->sum.e[k];
        const int len = e->rows * e->cols;
        const float cnt = (float)e->cnt;
        /* g = sum.get(key).divi(cnt) -- in place; aliased matrices (the shared
         * wide gbar) are divided once per key, harmless for cnt == 1 */
        for (int i = 0; i < len; ++i) e->data[i] = e->data[i] / cnt;
        sent *g = smap_put(&m->grads, e->key);
        if (g->owned) free(g->data);
        g->data = fdup(e->data, len); g->rows = len; g->cols = 1; g->owned = 1;
    }
}

void orc_model_apply_update(orc_model *m) {
    if (!m->pending) return;
    orc_store *s = m->st;
    for (int k = 0; k < m->grads.n; ++k) {
        sent *g = &m->grads.e[k];
        store_apply(s, g->key, g->data, g->rows, -1);
    }
    smap_clear(&s->sum);                                  /* KVStore.clear */
    m->pending = 0;
}

float orc_model_train(orc_model *m, const float *E, const float *Xd, const float *Wd,
                      const float *Y, int B, int do_update) {
    if (m->pending) { smap_clear(&m->st->sum); m->pending = 0; }
    pull_weights(m);                                      /* train/TrainerThread.java:34 */
    forward(m, E, Xd, Wd, B);                             /* model/DNN.java:44-46 */
    float loss = orc_ce_forward(m->P, Y, B);              /* :49 */
    smap_clear(&m->grads);
    m->pending = 0;
    if (loss <= (float)pow(10, -2) || loss != loss)       /* :58-63 slim / NaN: no backward */
        return loss;
    backward(m, Y);
    snapshot_grads(m);
    m->pending = 1;
    if (do_update) orc_model_apply_update(m);
    return loss;
}

void orc_model_predict(orc_model *m, const float *E, const float *Xd, const float *Wd, int B, float *P) {
    pull_weights(m);
    forward(m, E, Xd, Wd, B);
    memcpy(P, m->P, sizeof(float) * B);
}

const float *orc_model_act(orc_model *m, int layer, int *rows, int *cols) {
    const int FD = m->F * m->D;
    if (cols) *cols = m->B;
    if (layer == 0) { if (rows) *rows = FD; return m->embA; }
    if (layer == 1) { if (rows) *rows = FD + m->X; return m->concatA; }
    layer -= 2;
    if (layer < 0 || layer >= m->nfc) return NULL;
    if (rows) *rows = m->fc[layer].out;
    return m->fc[layer].A;
}
const float *orc_model_delta(orc_model *m, int layer, int *rows, int *cols) {
    layer -= 2;
    if (layer < 0 || layer >= m->nfc) return NULL;
    if (rows) *rows = m->fc[layer].in;
    if (cols) *cols = m->B;
    return m->fc[layer].delta;
}
const float *orc_model_wide_logit(orc_model *m, int *B) { if (B) *B = m->B; return m->wideZ; }
const float *orc_model_p(orc_model *m, int *B) { if (B) *B = m->B; return m->P; }
const float *orc_model_grad(orc_model *m, const char *key, int *len) {
    sent *e = smap_get(&m->grads, key);
    if (!e) return NULL;
    if (len) *len = e->rows;
    return e->data;
}
int orc_model_num_grad_keys(orc_model *m) { return m->grads.n; }
const char *orc_model_grad_key(orc_model *m, int i) { return m->grads.e[i].key; }

/* ===================================================================== */
/* PS semantics                                                          */
/* ===================================================================== */

struct orc_ps {
    int n, floor_mod;
    orc_store **shard;
    smap *upd;   /* per shard: updateKeys key -> updater kind (cnt: 0 adam, 1 ftrl)  net/PServer.java:44 */
    long global_step;
};

orc_ps *orc_ps_new(int nshards, uint64_t seed, int floor_mod) {
    orc_ps *p = (orc_ps *)calloc(1, sizeof *p);
    p->n = nshards; p->floor_mod = floor_mod;
    p->shard = (orc_store **)malloc(sizeof(orc_store *) * nshards);
    p->upd = (smap *)malloc(sizeof(smap) * nshards);
    for (int i = 0; i < nshards; ++i) { p->shard[i] = orc_store_new(seed); smap_init(&p->upd[i]); }
    return p;
}
void orc_ps_free(orc_ps *p) {
    if (!p) return;
    for (int i = 0; i < p->n; ++i) {
        /* pushed gradients are owned copies */
        for (int s = 0; s < p->shard[i]->sum.n; ++s) p->shard[i]->sum.e[s].owned = 1;
        orc_store_free(p->shard[i]);
        smap_free(&p->upd[i]);
    }
    free(p->shard); free(p->upd); free(p);
}
orc_store *orc_ps_shard(orc_ps *p, int s) { return p->shard[s]; }
int orc_ps_route(orc_ps *p, const char *key) { return orc_mod_shard(key, p->n, p->floor_mod); }

static int updater_kind_from_name(const char *name) {
    /* update/AdamUpdater.java:72-74 "adam@alfa:..@beta1:..", update/FtrlUpdater.java:78-80
     * "adam@alfa:..@beta:..@l1:.." (sic: also prefixed adam@).  -1 = unknown (Resp 500). */
    if (strncmp(name, "adam@", 5) != 0) return -1;
    if (strstr(name, "@beta1:")) return 0;
    if (strstr(name, "@l1:")) return 1;
    return -1;
}

int orc_ps_push(orc_ps *p, const char *key, const float *g, int len, const char *updater_name, int is_async) {
    /* net/PServer.java:164-195 */
    int kind = updater_kind_from_name(updater_name);
    if (kind < 0) return 500;
    int sh = orc_ps_route(p, key);
    orc_store *st = p->shard[sh];
    if (!smap_get(&st->store, key)) return 204;
    if (is_async) {
        /* :176-184 intended: apply this push's gradient alone, on arrival */
        store_apply(st, key, g, len, kind);
        return 200;
    }
    sent *e = smap_get(&st->sum, key);
    if (!e) {
        e = smap_put(&st->sum, key);
        e->data = fdup(g, len); e->rows = len; e->cols = 1; e->cnt = 1; e->owned = 1;
    } else {
        for (int i = 0; i < len; ++i) e->data[i] = g[i] + e->data[i];
        e->cnt++;
    }
    sent *u = smap_get(&p->upd[sh], key);
    if (!u) { u = smap_put(&p->upd[sh], key); u->cnt = kind; }   /* first updater name wins :187-189 */
    return 200;
}

void orc_ps_barrier_update(orc_ps *p) {
    /* net/PServer.java:197-214 psUpdate: for each pending key,
     * KVStore.update(updater,key) :202-218 -> g = sum/cnt ; then globalStep++ */
    for (int sh = 0; sh < p->n; ++sh) {
        orc_store *st = p->shard[sh];
        for (int k = 0; k < p->upd[sh].n; ++k) {
            const char *key = p->upd[sh].e[k].key;
            sent *e = smap_get(&st->sum, key);
            if (!e) continue;
            const int len = e->rows;
            const float cnt = (float)e->cnt;
            for (int i = 0; i < len; ++i) e->data[i] = e->data[i] / cnt;
            store_apply(st, key, e->data, len, (int)p->upd[sh].e[k].cnt);
        }
        smap_clear(&p->upd[sh]);
        smap_clear(&st->sum);   /* intended semantics: the reference never clears (App. A.9 bug, not copied) */
    }
    p->global_step++;
}
long orc_ps_global_step(orc_ps *p) { return p->global_step; }
