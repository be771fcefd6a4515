"""ctypes binding of oracle/libps_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under ps_amd/.
PARITY UNPINNED (see ps_oracle.h): no reference run / golden vectors exist.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libps_oracle.so")

DNN, WIDEDEEP = 0, 1
GRAD_COMPAT, GRAD_INTENDED = 0, 1
TABLE_WIDE = 1 << 20
TABLE_WIDE_B = (1 << 20) + 1


def TABLE_FC(i):
    return (2 << 20) + 2 * i


def build(force=False):
    src = os.path.join(_HERE, "ps_oracle.c")
    hdr = os.path.join(_HERE, "ps_oracle.h")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    def sig(name, res, *args):
        f = getattr(L, name); f.restype = res; f.argtypes = list(args)
    sig("orc_java_hashcode", C.c_int32, C.c_char_p)
    sig("orc_float_to_string", C.c_int, C.c_float, C.c_char_p, C.c_int)
    sig("orc_emb_key", C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_int)
    sig("orc_wide_key", C.c_int, C.c_float, C.c_char_p, C.c_int)
    sig("orc_mod_shard", C.c_int, C.c_char_p, C.c_int, C.c_int)
    sig("orc_matrixutil_hash", C.c_float, C.c_float, C.c_int)
    sig("orc_init_value", C.c_float, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float)
    sig("orc_xavier_scale", C.c_float, C.c_int, C.c_int)
    sig("orc_sigmoid_clip", C.c_float, C.c_float)
    sig("orc_adam_update", None, _fp, _fp, _fp, _fp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float)
    sig("orc_ftrl_update", C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float)
    sig("orc_emb_geff", None, _fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int)
    sig("orc_ce_forward", C.c_float, _fp, _fp, C.c_int)
    sig("orc_ce_backward", None, _fp, _fp, C.c_int, _fp)
    sig("orc_sgemm_nn", None, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp)
    sig("orc_store_new", C.c_void_p, C.c_uint64)
    sig("orc_store_free", None, C.c_void_p)
    sig("orc_store_get", _fp, C.c_void_p, C.c_char_p, _ip, _ip)
    sig("orc_store_put", None, C.c_void_p, C.c_char_p, _fp, C.c_int, C.c_int)
    sig("orc_store_size", C.c_int, C.c_void_p)
    sig("orc_store_state", _fp, C.c_void_p, C.c_char_p, C.c_int, _ip)
    sig("orc_model_new", C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int)
    sig("orc_model_free", None, C.c_void_p)
    sig("orc_model_set_grad_mode", None, C.c_void_p, C.c_int, C.c_int, C.c_int)
    sig("orc_model_train", C.c_float, C.c_void_p, _fp, _fp, _fp, _fp, C.c_int, C.c_int)
    sig("orc_model_predict", None, C.c_void_p, _fp, _fp, _fp, C.c_int, _fp)
    sig("orc_model_apply_update", None, C.c_void_p)
    sig("orc_model_act", _fp, C.c_void_p, C.c_int, _ip, _ip)
    sig("orc_model_delta", _fp, C.c_void_p, C.c_int, _ip, _ip)
    sig("orc_model_wide_logit", _fp, C.c_void_p, _ip)
    sig("orc_model_p", _fp, C.c_void_p, _ip)
    sig("orc_model_grad", _fp, C.c_void_p, C.c_char_p, _ip)
    sig("orc_model_num_grad_keys", C.c_int, C.c_void_p)
    sig("orc_model_grad_key", C.c_char_p, C.c_void_p, C.c_int)
    sig("orc_ps_new", C.c_void_p, C.c_int, C.c_uint64, C.c_int)
    sig("orc_ps_free", None, C.c_void_p)
    sig("orc_ps_shard", C.c_void_p, C.c_void_p, C.c_int)
    sig("orc_ps_route", C.c_int, C.c_void_p, C.c_char_p)
    sig("orc_ps_push", C.c_int, C.c_void_p, C.c_char_p, _fp, C.c_int, C.c_char_p, C.c_int)
    sig("orc_ps_barrier_update", None, C.c_void_p)
    sig("orc_ps_global_step", C.c_long, C.c_void_p)
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def _np(ptr, n):
    if not ptr:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


# ---- scalar helpers ------------------------------------------------------
def java_hashcode(s):
    return lib().orc_java_hashcode(s.encode())


def float_to_string(v):
    b = C.create_string_buffer(64)
    lib().orc_float_to_string(C.c_float(v), b, 64)
    return b.value.decode()


def emb_key(field, idv):
    b = C.create_string_buffer(96)
    lib().orc_emb_key(field, C.c_float(idv), b, 96)
    return b.value.decode()


def wide_key(idv):
    b = C.create_string_buffer(96)
    lib().orc_wide_key(C.c_float(idv), b, 96)
    return b.value.decode()


def mod_shard(key, n, floor_mod=False):
    return lib().orc_mod_shard(key.encode(), n, int(floor_mod))


def matrixutil_hash(idv, size):
    return lib().orc_matrixutil_hash(C.c_float(idv), size)


def init_value(seed, table, row, col, scale):
    return lib().orc_init_value(seed, table, row, col, C.c_float(scale))


def xavier_scale(i, o):
    return lib().orc_xavier_scale(i, o)


def sigmoid_clip(x):
    return lib().orc_sigmoid_clip(C.c_float(x))


def init_rows(seed, table, rows, D, scale):
    """[len(rows), D] embedding rows exactly as the lazy init would make them."""
    out = np.empty((len(rows), D), np.float32)
    L = lib()
    for i, r in enumerate(rows):
        for d in range(D):
            out[i, d] = L.orc_init_value(seed, table, int(r), d, C.c_float(scale))
    return out


def init_dense(seed, table, n, scale):
    L = lib()
    return np.array([L.orc_init_value(seed, table, i, 0, C.c_float(scale)) for i in range(n)], np.float32)


def adam_update(w, g, M, V, alfa=0.005, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place on copies; returns (w, M, V).  Hyper-parameters are cast to
    float32 as update/AdamUpdater.java:43-48 does."""
    w, wp = _f(np.array(w, np.float32).ravel().copy()); g, gp = _f(np.ravel(g))
    M, Mp = _f(np.array(M, np.float32).ravel().copy()); V, Vp = _f(np.array(V, np.float32).ravel().copy())
    lib().orc_adam_update(wp, gp, Mp, Vp, w.size, np.float32(alfa), np.float32(beta1), np.float32(beta2), np.float32(eps))
    return w, M, V


def ftrl_update(w, g, z, n, alfa=0.005, beta=1.0, l1=0.001, l2=0.001):
    w, wp = _f(np.array(w, np.float32).ravel().copy()); g, gp = _f(np.ravel(g))
    z, zp = _f(np.array(z, np.float32).ravel().copy()); n, np_ = _f(np.array(n, np.float32).ravel().copy())
    did = lib().orc_ftrl_update(wp, gp, zp, np_, w.size, np.float32(alfa), np.float32(beta), np.float32(l1), np.float32(l2))
    return w, z, n, did


def emb_geff(gk, mode=GRAD_COMPAT, chunk=0):
    gk, gp = _f(gk)
    n, D = gk.shape
    out = np.empty(D, np.float32)
    lib().orc_emb_geff(gp, n, D, out.ctypes.data_as(_fp), mode, chunk)
    return out


def ce_forward(p, y):
    p, pp = _f(p); y, yp = _f(y)
    return lib().orc_ce_forward(pp, yp, p.size)


def ce_backward(p, y):
    p, pp = _f(p); y, yp = _f(y)
    d = np.empty_like(p)
    lib().orc_ce_backward(pp, yp, p.size, d.ctypes.data_as(_fp))
    return d


# ---- store / model -------------------------------------------------------
class Store:
    def __init__(self, seed=0, _handle=None, _owner=None):
        self._owner = _owner
        self.h = _handle if _handle is not None else lib().orc_store_new(seed)

    def __del__(self):
        if self._owner is None and getattr(self, "h", None):
            lib().orc_store_free(self.h); self.h = None

    def get(self, key):
        r, c = C.c_int(), C.c_int()
        p = lib().orc_store_get(self.h, key.encode(), C.byref(r), C.byref(c))
        return _np(p, r.value * c.value)

    def put(self, key, data, rows=None, cols=1):
        data, dp = _f(np.ravel(data))
        rows = data.size // cols if rows is None else rows
        lib().orc_store_put(self.h, key.encode(), dp, rows, cols)

    def size(self):
        return lib().orc_store_size(self.h)

    def state(self, key, which):
        n = C.c_int()
        p = lib().orc_store_state(self.h, key.encode(), which, C.byref(n))
        return _np(p, n.value)


class Model:
    """model/DNN.java / model/WideDeepNN.java restated; thread = 1."""

    def __init__(self, store, kind, F, D, X, fc_dims, wide_size=100000):
        self.store, self.kind, self.F, self.D, self.X = store, kind, F, D, X
        self.fc_dims = list(fc_dims)
        arr = (C.c_int * len(fc_dims))(*fc_dims)
        self.h = lib().orc_model_new(store.h, kind, F, D, X, len(fc_dims), arr, wide_size)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_free(self.h); self.h = None

    def set_grad_mode(self, emb_mode=GRAD_COMPAT, wide_mode=GRAD_COMPAT, chunk=0):
        lib().orc_model_set_grad_mode(self.h, emb_mode, wide_mode, chunk)

    def train(self, E, Xd, Y, Wd=None, do_update=True):
        E, Ep = _f(E); Xd, Xp = _f(Xd); Y, Yp = _f(Y)
        B = Y.size
        Wp = None
        if Wd is not None:
            Wd, Wp = _f(Wd)
        return lib().orc_model_train(self.h, Ep, Xp, Wp, Yp, B, int(do_update))

    def apply_update(self):
        lib().orc_model_apply_update(self.h)

    def predict(self, E, Xd, Wd=None):
        E, Ep = _f(E); Xd, Xp = _f(Xd)
        B = E.shape[0]
        Wp = None
        if Wd is not None:
            Wd, Wp = _f(Wd)
        P = np.empty(B, np.float32)
        lib().orc_model_predict(self.h, Ep, Xp, Wp, B, P.ctypes.data_as(_fp))
        return P

    def act(self, layer):
        r, c = C.c_int(), C.c_int()
        p = lib().orc_model_act(self.h, layer, C.byref(r), C.byref(c))
        a = _np(p, r.value * c.value)
        return None if a is None else a.reshape(c.value, r.value)   # [B][features]

    def delta(self, layer):
        r, c = C.c_int(), C.c_int()
        p = lib().orc_model_delta(self.h, layer, C.byref(r), C.byref(c))
        a = _np(p, r.value * c.value)
        return None if a is None else a.reshape(c.value, r.value)

    def wide_logit(self):
        n = C.c_int(); p = lib().orc_model_wide_logit(self.h, C.byref(n)); return _np(p, n.value)

    def p(self):
        n = C.c_int(); p = lib().orc_model_p(self.h, C.byref(n)); return _np(p, n.value)

    def grad(self, key):
        n = C.c_int(); p = lib().orc_model_grad(self.h, key.encode(), C.byref(n)); return _np(p, n.value)

    def grad_keys(self):
        L = lib()
        return [L.orc_model_grad_key(self.h, i).decode() for i in range(L.orc_model_num_grad_keys(self.h))]


ADAM_NAME = "adam@alfa:0.005@beta1:0.9@beta2:0.999@epsilon:1.0E-8@"
FTRL_NAME = "adam@alfa:0.005@beta:1.0@l1:0.001@l2:0.001@"


class PS:
    """net/PServer.java + net/PSRouterClient.java semantics (BSP / async)."""

    def __init__(self, nshards, seed=0, floor_mod=True):
        self.n = nshards
        self.h = lib().orc_ps_new(nshards, seed, int(floor_mod))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ps_free(self.h); self.h = None

    def shard(self, i):
        return Store(_handle=lib().orc_ps_shard(self.h, i), _owner=self)

    def route(self, key):
        return lib().orc_ps_route(self.h, key.encode())

    def push(self, key, g, updater_name=ADAM_NAME, is_async=False):
        g, gp = _f(np.ravel(g))
        return lib().orc_ps_push(self.h, key.encode(), gp, g.size, updater_name.encode(), int(is_async))

    def barrier_update(self):
        lib().orc_ps_barrier_update(self.h)

    def global_step(self):
        return lib().orc_ps_global_step(self.h)


# ---------------------------------------------------------------------------
# data/LibsvmParser.java:13-25 + CTR.parseFeature (CTR.java:47-68) + DataSource.readLine
# (data/DataSource.java:25-46), restated in plain Python (test infrastructure only).
# ---------------------------------------------------------------------------
def parse_libsvm(text, F, X, wide_size=0, offset=0, step=1):
    """Returns E [n][F] float32 (ids AS FLOATS, as the reference keeps them), X [n][X], Y [n], W [n][F] float32."""
    import numpy as _np
    raw = text.split("\n")
    if raw and raw[-1] == "":
        raw.pop()                                                           # BufferedReader.readLine: no line after the last newline
    lines = raw[offset::step]                                               # DataSource.readLine counts RAW lines (DataSource.java:25-46)
    lines = [ln for ln in lines if ln.strip() != ""]                        # a selected blank line parses to an empty list: dropped
    n = len(lines)
    E = _np.zeros((n, F), _np.float32); Xd = _np.zeros((n, X), _np.float32); Y = _np.zeros(n, _np.float32)
    for i, ln in enumerate(lines):
        cols = ln.rstrip("\r ").split(" ")
        cols = [c for c in cols if c != ""]
        Y[i] = _np.float32(cols[0])                                         # Float.parseFloat: nearest float
        for j in range(1, 1 + F):
            E[i, j - 1] = _np.float32(int(cols[j].split(":")[0]))          # long -> float (CTR.java:57)
        for j in range(1 + F, 1 + F + X):
            Xd[i, j - 1 - F] = _np.float32(cols[j].split(":")[1])
    W = _np.fmod(E, _np.float32(wide_size)).astype(_np.float32) if wide_size > 0 else None   # MatrixUtil.hash
    return E, Xd, Y, W


# ---------------------------------------------------------------------------
# evaluate/AUC.java:32-82, op for op in double (test infrastructure only)
# ---------------------------------------------------------------------------
def auc(p, y):
    import numpy as _np
    pd = _np.asarray(p, _np.float32).astype(_np.float64)            # new MutablePair((double) p[i], (double) y[i])
    yd = _np.asarray(y, _np.float32).astype(_np.float64)
    # Arrays.sort(Object[], Comparator) is a stable merge sort; the comparator is Double.compareTo: a TOTAL order
    # (-0.0 < 0.0, every NaN equal and above +Infinity), i.e. the order of these keys
    bits = pd.view(_np.int64)
    key = _np.where(bits < 0, ~bits.view(_np.uint64), bits.view(_np.uint64) | _np.uint64(1 << 63))
    key = _np.where(_np.isnan(pd), _np.uint64(0xFFFFFFFFFFFFFFFF), key)
    order = _np.argsort(key, kind="stable")
    pos = float((yd > 0.0).sum()); neg = float(len(yd)) - pos        # sampleCount
    tp = fp = 0.0
    prev = 0.0
    total = 0.0
    with _np.errstate(divide="ignore", invalid="ignore"):
        for i in order[::-1]:                                        # getCoordinatePoint: from the highest p down
            if yd[i] > 0.0:
                fp += 1.0
            else:
                tp += 1.0
            x = _np.float64(tp) / _np.float64(pos)                   # (tp / posNum, fp / negNum)
            yy = _np.float64(fp) / _np.float64(neg)
            if x != prev:                                            # calculate()
                total = total + (x - prev) * yy
                prev = x
    return float(total)
