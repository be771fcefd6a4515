"""Host-side mirror of the reference's interface for the hot path, over the
C ABI (include/ps_native.h).  Same names, argument meaning and error
behaviour as the Java types so the parity tests read like the reference:

    update.AdamUpdater / FtrlUpdater / SimpleUpdater   update/*.java
    net.Mod (Router)                                   net/Mod.java:13-15
    store.KVStore                                      store/KVStore.java
    model.DNN / model.WideDeepNN (buildModel, train,
        predict, pullWeights, getUpdater)              model/DNN.java, model/WideDeepNN.java
    train.Trainer (thread = 1)                         train/Trainer.java:70-101

Arrays use the reference's byte layout: a FloatMatrix "features x B"
(column-major) is a numpy [B, features] array.  The HIP library is the only
compute path; nothing here falls back to numpy.
"""
import ctypes as C

import numpy as np

from . import native as N


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


# ---------------------------------------------------------------------------
# update.Updater
# ---------------------------------------------------------------------------
class Updater:
    def __init__(self, st):
        self._s = st

    def getName(self):
        buf = C.create_string_buffer(160)
        N.check(N.lib().ps_updater_name(C.byref(self._s), buf, 160))
        return buf.value.decode()

    @staticmethod
    def fromName(name):
        """AdamUpdater(String) / FtrlUpdater(String) / SimpleUpdater(String);
        an unknown name is Resp 500 (net/PServer.java:169-175)."""
        st = N.ps_updater_t()
        N.check(N.lib().ps_updater_from_name(name.encode(), C.byref(st)))
        return Updater(st)

    @property
    def kind(self):
        return self._s.kind


class AdamUpdater(Updater):
    def __init__(self, alfa=0.005, beta1=0.9, beta2=0.999, epsilon=1e-8):
        st = N.ps_updater_t()
        st.kind = N.PS_UPD_ADAM
        st.alfa, st.beta1, st.beta2, st.epsilon = alfa, beta1, beta2, epsilon   # (float) casts, AdamUpdater.java:43-48
        super().__init__(st)


class FtrlUpdater(Updater):
    def __init__(self, alfa=0.005, beta=1.0, l1=0.001, l2=0.001):
        st = N.ps_updater_t()
        st.kind = N.PS_UPD_FTRL
        st.alfa, st.beta, st.l1, st.l2 = alfa, beta, l1, l2
        super().__init__(st)


class SimpleUpdater(Updater):
    def __init__(self, eta):
        st = N.ps_updater_t()
        st.kind = N.PS_UPD_SIMPLE
        st.eta = eta
        super().__init__(st)


# ---------------------------------------------------------------------------
# net.Router
# ---------------------------------------------------------------------------
class Mod:
    """net/Mod.java with the floorMod fix (Java % goes negative)."""

    def __init__(self, n):
        self.n = n

    def shard(self, key):
        return N.lib().ps_router_shard_key(key.encode(), self.n)


def java_string_hash(key):
    return N.lib().ps_java_string_hash(key.encode())


# ---------------------------------------------------------------------------
# store.KVStore
# ---------------------------------------------------------------------------
class KVStore:
    """One GPU-resident shard.  `KVStore.ins(device)` mirrors the singleton."""

    _ins = {}

    def __init__(self, device=0, seed=0):
        h = C.c_void_p()
        N.check(N.lib().ps_store_create(device, seed, C.byref(h)))
        self.h = h
        self.device = device
        self._dependents = []      # weak references to models / datasets built on this store: closed before it is

    def _adopt(self, obj):
        import weakref
        self._dependents.append(weakref.ref(obj))
        # a model's native handle, shared with the model: when both are collected together (a reference cycle, interpreter
        # exit) the weak reference above is already dead, and the model must still be destroyed BEFORE its store
        if hasattr(obj, "_hcell"):
            self._model_cells = getattr(self, "_model_cells", []) + [obj._hcell]

    @classmethod
    def ins(cls, device=0, seed=0):
        if device not in cls._ins:
            cls._ins[device] = cls(device, seed)
        return cls._ins[device]

    def close(self):
        if getattr(self, "h", None):
            for ref in getattr(self, "_dependents", []):       # a model keeps pointers into the store
                obj = ref()
                if obj is not None:
                    obj.close()
            self._dependents = []
            for cell in getattr(self, "_model_cells", []):
                if cell[0]:
                    N.lib().ps_model_destroy(cell[0])
                    cell[0] = None
            self._model_cells = []
            N.lib().ps_store_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- tables
    def create_embedding(self, rows, D, state_slots=2, shard=0, nshards=1, route_mode=N.PS_ROUTE_ID_MOD):
        rows = np.ascontiguousarray(rows, np.int64)
        N.check(N.lib().ps_store_create_embedding(self.h, len(rows), _ip(rows), D, state_slots, shard, nshards, route_mode))
        self.F, self.D = len(rows), D

    def create_wide(self, wide_size):
        N.check(N.lib().ps_store_create_wide(self.h, wide_size))

    def create_fc(self, layer, in_dims, out_dims):
        N.check(N.lib().ps_store_create_fc(self.h, layer, in_dims, out_dims))

    def set_updater(self, key, updater):
        N.check(N.lib().ps_store_set_updater(self.h, key.encode(), C.byref(updater._s)))

    # -- KVStore.get / put
    def get(self, key, cap=1 << 22):
        """KVStore.get(key): None when the key is absent (store/KVStore.java:129-134)."""
        out = np.empty(cap, np.float32)
        n = C.c_int()
        rc = N.lib().ps_store_get(self.h, key.encode(), _fp(out), cap, C.byref(n))
        if rc == N.PS_MISSING:
            return None
        N.check(rc)
        return out[:n.value].copy()

    def put(self, key, val):
        val = np.ascontiguousarray(val, np.float32).ravel()
        N.check(N.lib().ps_store_put(self.h, key.encode(), _fp(val), val.size))

    def get_rows(self, field, ids, which=0, D=None):
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty((len(ids), D or self.D), np.float32)
        N.check(N.lib().ps_store_get_rows(self.h, field, _ip(ids), len(ids), which, _fp(out)))
        return out

    def put_rows(self, field, ids, val, which=0):
        ids = np.ascontiguousarray(ids, np.int64)
        val = np.ascontiguousarray(val, np.float32)
        N.check(N.lib().ps_store_put_rows(self.h, field, _ip(ids), len(ids), which, _fp(val)))

    def get_wide(self, ids, which=0):
        ids = np.ascontiguousarray(ids, np.int64)
        out = np.empty(len(ids), np.float32)
        N.check(N.lib().ps_store_get_wide(self.h, _ip(ids), len(ids), which, _fp(out)))
        return out

    def put_wide(self, ids, val, which=0):
        ids = np.ascontiguousarray(ids, np.int64)
        val = np.ascontiguousarray(val, np.float32)
        N.check(N.lib().ps_store_put_wide(self.h, _ip(ids), len(ids), which, _fp(val)))

    def save(self, path):
        """Checkpoint of this shard (tables, updater state, FC tensors, updaters, globalStep)."""
        N.check(N.lib().ps_store_save(self.h, str(path).encode()))

    def load(self, path):
        N.check(N.lib().ps_store_load(self.h, str(path).encode()))

    def global_step(self):
        return N.lib().ps_store_global_step(self.h)

    def advance_global_step(self, by=1):
        N.check(N.lib().ps_store_advance_global_step(self.h, by))

    def key_length(self, key):
        """floats a push of `key` must carry on this shard (raises PsError PS_MISSING for a key the store does not hold)"""
        n = C.c_int()
        N.check(N.lib().ps_store_key_length(self.h, key.encode(), C.byref(n)))
        return n.value

    def push_update(self, messages, is_async=False):
        """PServer.push x n + psUpdate (net/PServer.java:164-214) by string key: messages = [(key, gradient), ...] in
        ARRIVAL order.  BSP: every key's pushes summed in arrival order, / count, one updater step; async: one step per
        message.  All arithmetic on the device."""
        n = len(messages)
        if n == 0:
            return
        arrs = [np.ascontiguousarray(g, np.float32).ravel() for _, g in messages]
        keys = (C.c_char_p * n)(*[k.encode() for k, _ in messages])
        ptrs = (C.POINTER(C.c_float) * n)(*[_fp(a) for a in arrs])
        lens = (C.c_int * n)(*[a.size for a in arrs])
        N.check(N.lib().ps_store_push_update(self.h, n, keys, ptrs, lens, 1 if is_async else 0))

    def bytes(self):
        return N.lib().ps_store_bytes(self.h)

    def sync(self):
        N.check(N.lib().ps_store_sync(self.h))


# ---------------------------------------------------------------------------
# model.DNN / model.WideDeepNN
# ---------------------------------------------------------------------------
class Batch:
    """Keeps the numpy arrays alive behind a ps_batch_t."""

    def __init__(self, E, X, Y=None, W=None, offsets=None):
        self.E = np.ascontiguousarray(E, np.int64)
        self.X = None if X is None else np.ascontiguousarray(X, np.float32)
        self.Y = None if Y is None else np.ascontiguousarray(Y, np.float32).ravel()
        self.W = None if W is None else np.ascontiguousarray(W, np.int64)
        self.offsets = None if offsets is None else np.ascontiguousarray(offsets, np.int64)
        b = N.ps_batch_t()
        if self.offsets is not None:
            b.B = int(self.X.shape[0]) if self.X is not None else int(self.Y.size)
        else:
            b.B = int(self.E.shape[0])
        b.ids = self.E.ctypes.data
        b.offsets = None if self.offsets is None else self.offsets.ctypes.data
        b.dense = None if self.X is None else self.X.ctypes.data
        b.labels = None if self.Y is None else self.Y.ctypes.data
        b.wide_ids = None if self.W is None else self.W.ctypes.data
        b.on_device = 0
        self.c = b


class DeviceBatch:
    """A minibatch resident in HBM (ps_batch_t.on_device = 1): what the bench
    times against, so that `value` excludes the PCIe hand-over."""

    def __init__(self, store, E, X, Y=None, W=None, offsets=None):
        self.store = store
        self._bufs = []
        host = Batch(E, X, Y, W, offsets)

        def up(a):
            if a is None:
                return None
            p = C.c_void_p()
            N.check(N.lib().ps_dev_alloc(store.h, a.nbytes, C.byref(p)))
            N.check(N.lib().ps_dev_upload(store.h, p, a.ctypes.data, a.nbytes))
            self._bufs.append(p)
            return p.value

        b = N.ps_batch_t()
        b.B = host.c.B
        b.ids, b.offsets, b.dense = up(host.E), up(host.offsets), up(host.X)
        b.labels, b.wide_ids = up(host.Y), up(host.W)
        b.on_device = 1
        b.nnz = int(host.E.size) if host.offsets is not None else 0     # spares ps_model_train the read-back of offsets[B*F]
        self.c = b
        self.nnz = int(host.E.size)
        store._adopt(self)

    def close(self):
        for p in self._bufs:
            N.lib().ps_dev_free(self.store.h, p)
        self._bufs = []

    def __del__(self):
        try:
            if getattr(self.store, "h", None):
                self.close()
        except Exception:
            pass


class _Model:
    KIND = N.PS_MODEL_DNN

    def __init__(self, store, F, D, X, fc_dims, wide_size=0, max_batch=4096, max_nnz=0,
                 emb_grad_mode=N.PS_GRAD_COMPAT, wide_grad_mode=N.PS_GRAD_COMPAT, use_graph=0, emb_sum_order=N.PS_SUM_AUTO, keep_grads=False):
        self.store = store
        cfg = N.ps_model_config_t()
        cfg.kind = self.KIND
        cfg.F, cfg.D, cfg.X, cfg.nfc = F, D, X, len(fc_dims)
        for i, d in enumerate(fc_dims):
            cfg.fc_dims[i] = d
        cfg.wide_size = wide_size
        cfg.max_batch, cfg.max_nnz = max_batch, max_nnz
        cfg.emb_grad_mode, cfg.wide_grad_mode, cfg.use_graph = emb_grad_mode, wide_grad_mode, use_graph
        cfg.emb_sum_order = emb_sum_order
        self.cfg = cfg
        self.F, self.D, self.X, self.fc_dims = F, D, X, list(fc_dims)
        h = C.c_void_p()
        N.check(N.lib().ps_model_create(store.h, C.byref(cfg), C.byref(h)))
        self._hcell = [h]               # (self.h reads it: the store destroys its models' handles through this cell too)
        store._adopt(self)
        if keep_grads:                  # emb_grads() after a fused train(): parity tests (ps_native.h ps_model_set_keep_grads)
            self.keep_grads(True)
        self._updater = {"default": AdamUpdater()}

    def close(self):
        cell = getattr(self, "_hcell", None)
        if cell and cell[0]:
            N.lib().ps_model_destroy(cell[0])
            cell[0] = None

    @property
    def h(self):
        cell = self.__dict__.get("_hcell")
        return cell[0] if cell else None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def getUpdater(self):
        return self._updater

    def pullWeights(self):
        """Model.pullWeights: parameters live in HBM behind the store; nothing to fetch."""

    @staticmethod
    def _batch(datas):
        if isinstance(datas, (Batch, DeviceBatch)) or hasattr(datas, "c"):      # incl. DataSet's device batches
            return datas
        return Batch(datas["E"], datas.get("X"), datas.get("Y"), datas.get("W"), datas.get("offsets"))

    def train(self, datas):
        """TrainerThread.call + the KVStore.update/clear tail of Trainer.train; returns the loss."""
        b = self._batch(datas)
        loss = C.c_float()
        N.check(N.lib().ps_model_train(self.h, C.byref(b.c), C.byref(loss)))
        return loss.value

    def train_async(self, batch):
        N.check(N.lib().ps_model_train(self.h, C.byref(batch.c), None))

    def forward(self, datas):
        b = self._batch(datas)
        self._last = b
        loss = C.c_float()
        N.check(N.lib().ps_model_forward(self.h, C.byref(b.c), C.byref(loss)))
        return loss.value

    def backward(self):
        N.check(N.lib().ps_model_backward(self.h))

    def update(self):
        N.check(N.lib().ps_model_update(self.h))

    def predict(self, datas):
        b = self._batch(datas)
        out = np.empty(b.c.B, np.float32)
        N.check(N.lib().ps_model_predict(self.h, C.byref(b.c), _fp(out)))
        return out

    def labels(self, datas):
        """The labels "Y" of a batch as a host array (device batches: copied back)."""
        b = self._batch(datas)
        if not b.c.on_device:
            return b.Y
        out = np.empty(b.c.B, np.float32)
        N.check(N.lib().ps_dev_download(self.store.h, out.ctypes.data, b.c.labels, out.nbytes))
        return out

    def sync(self):
        N.check(N.lib().ps_model_sync(self.h))

    # -- intermediates (Layer.A / Layer.delta) for parity tests
    def _get2d(self, fn, layer):
        r, c = C.c_int(), C.c_int()
        N.check(fn(self.h, layer, None, 0, C.byref(r), C.byref(c)))
        out = np.empty((c.value, r.value), np.float32)       # [B][features]
        N.check(fn(self.h, layer, _fp(out), out.size, C.byref(r), C.byref(c)))
        return out

    def act(self, layer):
        return self._get2d(N.lib().ps_model_get_act, layer)

    def delta(self, layer):
        return self._get2d(N.lib().ps_model_get_delta, layer)

    def p(self, B):
        out = np.empty(B, np.float32)
        N.check(N.lib().ps_model_get_p(self.h, _fp(out), B))
        return out

    def keep_grads(self, on=True):
        N.check(N.lib().ps_model_set_keep_grads(self.h, int(on)))

    def emb_grads(self, field):
        n = C.c_int64()
        N.check(N.lib().ps_model_get_emb_grads(self.h, field, None, None, 0, C.byref(n)))
        ids = np.empty(n.value, np.int64)
        g = np.empty((n.value, self.D), np.float32)
        N.check(N.lib().ps_model_get_emb_grads(self.h, field, _ip(ids), _fp(g), n.value, C.byref(n)))
        return ids, g

    def fc_grad(self, layer, bias=False):
        inn = (self.F * self.D + self.X) if layer == 0 else self.fc_dims[layer - 1]
        n = self.fc_dims[layer] if bias else inn * self.fc_dims[layer]
        out = np.empty(n, np.float32)
        N.check(N.lib().ps_model_get_fc_grad(self.h, layer, int(bias), _fp(out), n))
        return out

    # -- measurement
    def time_steps(self, batch, steps):
        ms = C.c_double()
        N.check(N.lib().ps_model_time_steps(self.h, C.byref(batch.c), steps, C.byref(ms)))
        return ms.value

    def set_profile(self, on, only=None):
        if on and only:
            N.check(N.lib().ps_model_set_profile_filter(self.h, only.encode()))
        else:
            N.check(N.lib().ps_model_set_profile(self.h, int(on)))

    def profile_report(self):
        buf = C.create_string_buffer(1 << 16)
        N.check(N.lib().ps_model_profile_report(self.h, buf, 1 << 16))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                name, cnt, ms = item.split(":")
                out[name] = (int(cnt), float(ms))
        return out


class DNN(_Model):
    KIND = N.PS_MODEL_DNN

    @staticmethod
    def buildModel(embeddingFieldNum, embeddingSize, numberFieldNum, fcLayerDims, store=None, rows=None, **kw):
        """DNN.buildModel (model/DNN.java:92-128).  `rows`: vocabulary per field
        (the reference grows its maps lazily; HBM tables are sized up front)."""
        store = store or KVStore.ins()
        if not hasattr(store, "F"):
            store.create_embedding(rows if rows is not None else [100000] * embeddingFieldNum, embeddingSize)
        return DNN(store, embeddingFieldNum, embeddingSize, numberFieldNum, fcLayerDims, **kw)


class WideDeepNN(_Model):
    KIND = N.PS_MODEL_WIDEDEEP

    @staticmethod
    def buildModel(embeddingFieldNum, embeddingSize, numberFieldNum, fcLayerDims, wideSize, store=None, rows=None, **kw):
        """WideDeepNN.buildModel (model/WideDeepNN.java:105-161): Ftrl on wide.*, Adam default."""
        store = store or KVStore.ins()
        if not hasattr(store, "F"):
            store.create_embedding(rows if rows is not None else [100000] * embeddingFieldNum, embeddingSize)
        m = WideDeepNN(store, embeddingFieldNum, embeddingSize, numberFieldNum, fcLayerDims, wide_size=wideSize, **kw)
        ftrl = FtrlUpdater(0.005, 1.0, 0.001, 0.001)
        m._updater["wide.weights"] = ftrl
        m._updater["wide.bias"] = ftrl
        return m


class Trainer:
    """train/Trainer.java for thread = 1 (what CTR.java:72 forces)."""

    def __init__(self, nThreads, modelCallable):
        if nThreads != 1:
            raise ValueError("one replica per GPU: thread-DP is replaced by one process per GPU")
        self.models = [modelCallable()]

    def train(self, dataList):
        if len(dataList) > len(self.models):
            raise RuntimeError("dataList size > thread size")     # Trainer.java:72-74
        return self.models[0].train(dataList[0])

    def predict(self, dataList):
        return [self.models[0].predict(d) for d in dataList]

    def getTrainResult(self):
        return self.models[0]


# ---------------------------------------------------------------------------
# data.LibsvmParser / data.FileSource / CTR (DataSet): libsvm text -> batches in HBM
# ---------------------------------------------------------------------------
def _ingest_cfg(F, X, batch, wide_size, threads, offset, step, ids_via_float):
    c = N.ps_ingest_config_t()
    c.F, c.X, c.batch, c.threads, c.offset, c.step = int(F), int(X), int(batch), int(threads), int(offset), int(step)
    c.ids_via_float, c.wide_size = int(bool(ids_via_float)), int(wide_size)
    return c


class LibsvmParser:
    """data/LibsvmParser.java + CTR.parseFeature (CTR.java:47-68) on the host: text -> {"E","X","Y","W"}.
    E/W come back as int64 [n][F] (the bytes of the reference's F x n float matrices for ids < 2^24)."""

    def __init__(self, F, X, wide_size=0, threads=1, ids_via_float=True):
        self.F, self.X, self.wide_size, self.threads, self.ids_via_float = F, X, wide_size, threads, ids_via_float

    def parse(self, text, offset=0, step=1, first_line=0, max_lines=None):
        if isinstance(text, str):
            text = text.encode()
        n = C.c_int64()
        N.check(N.lib().ps_libsvm_count(text, len(text), offset, step, C.byref(n)))
        n = max(n.value - first_line, 0)
        if max_lines is not None:
            n = min(n, max_lines)
        cfg = _ingest_cfg(self.F, self.X, max(n, 1), self.wide_size, self.threads, offset, step, self.ids_via_float)
        E = np.zeros((n, self.F), np.int64); W = np.zeros((n, self.F), np.int64)
        X = np.zeros((n, self.X), np.float32); Y = np.zeros(n, np.float32)
        got = C.c_int64()
        N.check(N.lib().ps_libsvm_parse(text, len(text), C.byref(cfg), first_line, n, _ip(E), _fp(X), _fp(Y), _ip(W), C.byref(got)))
        assert got.value == n
        out = {"E": E, "X": X, "Y": Y}
        if self.wide_size > 0:
            out["W"] = W
        return out


class _IngestBatch:
    def __init__(self, c):
        self.c = c
        self.B = c.B


class DataSet:
    """The reader side of data/DataSet.java for CTR-shaped libsvm: host threads parse batch k+1 into pinned
    memory and copy it to HBM while batch k trains.  `source`: a file path (FileSource) or bytes."""

    def __init__(self, store, source, F, X, batch, wide_size=0, threads=4, offset=0, step=1, ids_via_float=True):
        self.store = store
        self.h = C.c_void_p()
        cfg = _ingest_cfg(F, X, batch, wide_size, threads, offset, step, ids_via_float)
        N.check(N.lib().ps_ingest_create(store.h, C.byref(cfg), C.byref(self.h)))
        store._adopt(self)
        if isinstance(source, (bytes, bytearray)):
            N.check(N.lib().ps_ingest_open_memory(self.h, bytes(source), len(source)))
        else:
            N.check(N.lib().ps_ingest_open_file(self.h, str(source).encode()))

    def lines(self):
        n = C.c_int64()
        N.check(N.lib().ps_ingest_lines(self.h, C.byref(n)))
        return n.value

    def next(self):
        """The next batch (device-resident), or None at the end of the data."""
        b = N.ps_batch_t()
        rc = N.lib().ps_ingest_next(self.h, C.byref(b))
        if rc == N.PS_MISSING:
            return None
        N.check(rc)
        return _IngestBatch(b)

    def __iter__(self):
        while True:
            b = self.next()
            if b is None:
                return
            yield b

    def reset(self):
        N.check(N.lib().ps_ingest_reset(self.h))

    def train(self, model, max_batches=-1):
        """CTR.java:84-100's loop in C (ps_ingest_train): the batches trained as they arrive; returns how many."""
        k = C.c_int64()
        N.check(N.lib().ps_ingest_train(self.h, model.h, int(max_batches), C.byref(k)))
        return k.value

    def stats(self):
        s, l, b = C.c_double(), C.c_int64(), C.c_int64()
        N.check(N.lib().ps_ingest_stats(self.h, C.byref(s), C.byref(l), C.byref(b)))
        return {"parse_seconds": s.value, "lines": l.value, "bytes": b.value}

    def close(self):
        if self.h:
            N.lib().ps_ingest_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            if getattr(self.store, "h", None):
                self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# evaluate.AUC
# ---------------------------------------------------------------------------
class AUC:
    """evaluate/AUC.java: AUC(p, y).calculate().  p, y: host arrays (or device pointers with n and on_device)."""

    def __init__(self, p, y, store=None, n=None, on_device=False):
        self.store = store if store is not None else KVStore.ins()
        self.on_device = bool(on_device)
        if self.on_device:
            self.p, self.y, self.n = p, y, int(n)
        else:
            self.p = np.ascontiguousarray(p, np.float32).ravel()
            self.y = np.ascontiguousarray(y, np.float32).ravel()
            self.n = int(self.p.size)
            assert self.y.size == self.n
        self.posNum = self.negNum = None

    def calculate(self):
        auc, pos, neg = C.c_double(), C.c_int64(), C.c_int64()
        pp = self.p if self.on_device else self.p.ctypes.data
        yy = self.y if self.on_device else self.y.ctypes.data
        N.check(N.lib().ps_auc_compute(self.store.h, pp, yy, self.n, int(self.on_device), C.byref(auc), C.byref(pos), C.byref(neg)))
        self.posNum, self.negNum = pos.value, neg.value
        return auc.value
