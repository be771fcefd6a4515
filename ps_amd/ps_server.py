"""The reference's wire, served from a GPU shard.

`service PS` of src/main/resources/proto/ps.proto -- get, getList, upsert, upsertList, push, barrier -- with the
behaviour of net/PServer.java, backed by one `ps_store_t` (HBM-resident tables) through the C ABI.  A reference worker
(`-Dmode=dist`, net/PSClient.java / PSRouterClient.java) can point its channel at this server unchanged: same service
name (`net.PS`), same messages (field numbers and types below are ps.proto's), same `Resp.ec` codes (200 / 204 / 500),
same BSP / async semantics:

  push     (PServer.java:164-195)  BSP: KVStore.sum(key, g) -- here: the message is queued in arrival order;
                                   async: sum + update(updater, key) at once -- here: one device updater step
  barrier  (PServer.java:236-283)  the last of `worker_num` arrivals wakes the update thread; psUpdate (:197-214) runs
                                   Updater.update on every pushed key with sum / count, globalStep++, everybody returns.
                                   Here psUpdate is ONE call, ps_store_push_update: sums in arrival order, / count and the
                                   updater on the device (the hot path's kernels).
  async barrier: globalStep++ and return (PServer.java:240-246)

There is no protoc in this image: the message classes are built from a FileDescriptorProto written out below, which is
ps.proto field for field (tests/test_ps_proto.py pins the wire bytes).  Not reproduced: the reference's bug that the PS
never clears `sum` between rounds (SURVEY App. A.9 -- every round's gradient would include all earlier ones).

Serving is not a hot path: one Python thread per call, the store behind one lock."""
import threading
from concurrent import futures

import numpy as np

try:
    import grpc
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
except ImportError as e:        # pragma: no cover
    raise ImportError("ps_amd.ps_server needs grpcio and protobuf") from e

from . import native as N
from .api import Updater

SERVICE = "net.PS"
EC_OK, EC_NULL, EC_ERR = 200, 204, 500


# ---------------------------------------------------------------------------
# ps.proto, as a descriptor
# ---------------------------------------------------------------------------
def _build_messages():
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "ps.proto"
    fd.package = "net"
    fd.syntax = "proto3"

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for fname, num, ftype, label, tname in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = ".net." + tname
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("Matrix", [("key", 1, F.TYPE_STRING, OPT, None), ("row", 2, F.TYPE_INT32, OPT, None), ("cols", 3, F.TYPE_INT32, OPT, None),
                   ("data", 4, F.TYPE_FLOAT, REP, None), ("update", 5, F.TYPE_BOOL, OPT, None)])
    msg("Resp", [("ec", 1, F.TYPE_INT32, OPT, None), ("em", 2, F.TYPE_STRING, OPT, None)])
    msg("RequestMeta", [("host", 1, F.TYPE_STRING, OPT, None)])
    msg("GetListMessage", [("meta", 1, F.TYPE_MESSAGE, OPT, "RequestMeta"), ("weights", 2, F.TYPE_MESSAGE, REP, "Matrix"),
                           ("resp", 3, F.TYPE_MESSAGE, OPT, "Resp")])
    msg("GetMessage", [("meta", 1, F.TYPE_MESSAGE, OPT, "RequestMeta"), ("weights", 2, F.TYPE_MESSAGE, OPT, "Matrix"),
                       ("resp", 4, F.TYPE_MESSAGE, OPT, "Resp")])
    msg("UpdateMessage", [("meta", 1, F.TYPE_MESSAGE, OPT, "RequestMeta"), ("weights", 2, F.TYPE_MESSAGE, OPT, "Matrix"),
                          ("resp", 3, F.TYPE_MESSAGE, OPT, "Resp"), ("replace", 4, F.TYPE_BOOL, OPT, None)])
    msg("UpdateListMessage", [("meta", 1, F.TYPE_MESSAGE, OPT, "RequestMeta"), ("weights", 2, F.TYPE_MESSAGE, REP, "Matrix"),
                              ("resp", 3, F.TYPE_MESSAGE, OPT, "Resp"), ("replace", 4, F.TYPE_BOOL, OPT, None)])
    msg("GradientMessage", [("meta", 1, F.TYPE_MESSAGE, OPT, "RequestMeta"), ("gradient", 2, F.TYPE_MESSAGE, OPT, "Matrix"),
                            ("isAsync", 3, F.TYPE_BOOL, OPT, None), ("updaterKey", 4, F.TYPE_STRING, OPT, None),
                            ("resp", 5, F.TYPE_MESSAGE, OPT, "Resp")])
    msg("BarrierMessage", [("meta", 1, F.TYPE_MESSAGE, OPT, "RequestMeta"), ("resp", 2, F.TYPE_MESSAGE, OPT, "Resp")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {m.name: message_factory.GetMessageClass(pool.FindMessageTypeByName("net." + m.name)) for m in fd.message_type}


M = _build_messages()
Matrix, Resp, RequestMeta = M["Matrix"], M["Resp"], M["RequestMeta"]
GetListMessage, GetMessage, UpdateMessage = M["GetListMessage"], M["GetMessage"], M["UpdateMessage"]
UpdateListMessage, GradientMessage, BarrierMessage = M["UpdateListMessage"], M["GradientMessage"], M["BarrierMessage"]
# rpc name -> (request class, response class), ps.proto:7-14
RPCS = {"get": (GetMessage, GetMessage), "getList": (GetListMessage, GetListMessage), "upsert": (UpdateMessage, UpdateMessage),
        "upsertList": (UpdateListMessage, UpdateListMessage), "push": (GradientMessage, GradientMessage),
        "barrier": (BarrierMessage, BarrierMessage)}


def _ok():
    return Resp(ec=EC_OK, em="")


def updater_group(key):
    """The key (or prefix) under which ps_store_set_updater keeps the updater of `key`'s table."""
    if key.startswith("emF"):
        return "emF"
    if key.startswith("wide."):
        return "wide"
    return key            # fc<i>.weights / fc<i>.bias


# ---------------------------------------------------------------------------
# net/PServer.java over a ps_amd.KVStore
# ---------------------------------------------------------------------------
class PsServicer:
    def __init__(self, store, worker_num=1, is_async=False):
        self.store = store
        self.worker_num = int(worker_num)
        self.is_async = bool(is_async)
        self.lock = threading.RLock()              # the store's C ABI is single-threaded
        self.round = threading.Condition(self.lock)
        self.pending = []                          # BSP: (key, gradient) in arrival order
        self.update_keys = {}                      # key -> updaterKey of its first push (PServer.java:187-190)
        self.updaters = {}                         # updaterKey -> parsed Updater
        self.arrived = 0                           # barrier arrivals of the current round
        self.generation = 0

    # ---- FloatMatrix <-> Matrix (util/MatrixUtil.java:84-109) ----
    def _shape(self, key, n):
        """rows x columns of the FloatMatrix behind `key` (data is column-major, i.e. our [in][out] bytes)."""
        if key.startswith("fc") and key.endswith(".weights"):
            b = self.store.get(key[:-len("weights")] + "bias")
            out = len(b) if b is not None else n
            return out, n // max(out, 1)
        return n, 1                                # embedding row D x 1, fc bias out x 1, wide 1 x 1

    def _to_matrix(self, key, val):
        m = Matrix(key=key)
        if val is None:
            return m                               # FloatMatrix_2_ProtoMatrix(key, null): the key alone
        val = np.asarray(val, np.float32).ravel()
        m.row, m.cols = self._shape(key, val.size)
        m.data.extend(val.tolist())
        return m

    # ---- rpcs ----
    def get(self, req, ctx=None):
        with self.lock:
            val = self.store.get(req.weights.key)
        if val is None:
            return GetMessage(resp=Resp(ec=EC_NULL, em="null weights"))
        return GetMessage(weights=self._to_matrix(req.weights.key, val), resp=_ok())

    def getList(self, req, ctx=None):
        out = GetListMessage(resp=_ok())
        with self.lock:
            for w in req.weights:
                out.weights.append(self._to_matrix(w.key, self.store.get(w.key)))
        return out

    def _upsert_one(self, w, replace):
        exists = self.store.get(w.key)
        update = True
        if exists is None or replace:
            update = False
            exists = np.asarray(w.data, np.float32)
            self.store.put(w.key, exists)          # unknown key kinds raise: answered with ec 500 by the caller
        m = self._to_matrix(w.key, exists)
        m.update = update
        return m

    def upsert(self, req, ctx=None):
        with self.lock:
            try:
                return UpdateMessage(weights=self._upsert_one(req.weights, req.replace), resp=_ok())
            except N.PsError as e:
                return UpdateMessage(resp=Resp(ec=EC_ERR, em=str(e)))

    def upsertList(self, req, ctx=None):
        out = UpdateListMessage(resp=_ok())
        with self.lock:
            try:
                for w in req.weights:
                    out.weights.append(self._upsert_one(w, req.replace))
            except N.PsError as e:
                return UpdateListMessage(resp=Resp(ec=EC_ERR, em=str(e)))
        return out

    def _updater(self, name, key):
        """updaterMap.get(updaterKey) (PServer.java:169): parsed once, installed for the key's table."""
        u = self.updaters.get(name)
        if u is None:
            u = Updater.fromName(name)             # raises PsError (PS_NO_UPDATER) for an unknown name
            self.updaters[name] = u
        grp = updater_group(key)
        if self.update_keys.get("@" + grp) != name:
            self.store.set_updater(grp, u)
            self.update_keys["@" + grp] = name
        return u

    def push(self, req, ctx=None):
        key = req.gradient.key
        g = np.asarray(req.gradient.data, np.float32)
        with self.lock:
            try:
                self._updater(req.updaterKey, key)
            except N.PsError:
                return GradientMessage(resp=Resp(ec=EC_ERR, em="updater is null"))
            try:
                if req.isAsync:
                    self.store.push_update([(key, g)], is_async=True)       # sum + update at once (:176-184)
                else:
                    # validated when it ARRIVES: a bad key or length is this worker's 500, not a failure of the whole
                    # round inside the last worker's barrier
                    want = self.store.key_length(key)
                    if g.size != want:
                        return GradientMessage(resp=Resp(ec=EC_ERR, em="%s wants %d floats, got %d" % (key, want, g.size)))
                    self.pending.append((key, g))                            # KVStore.sum; applied by psUpdate
            except N.PsError as e:
                return GradientMessage(resp=Resp(ec=EC_ERR, em=str(e)))
        return GradientMessage()                   # the reference answers with an empty message (:181, :192)

    def _ps_update(self):
        """PServer.psUpdate (:197-214): every pushed key, sum / count, its updater; globalStep++."""
        pend, self.pending = self.pending, []
        self.store.push_update(pend, is_async=False)
        self.store.advance_global_step(1)

    def barrier(self, req, ctx=None):
        with self.round:
            if self.is_async:                      # :240-246
                self.store.advance_global_step(1)
                return BarrierMessage(resp=_ok())
            gen = self.generation
            self.arrived += 1
            if self.arrived >= self.worker_num:    # the last worker of the round wakes the update thread: here it IS it
                try:
                    self._ps_update()
                finally:
                    self.arrived = 0
                    self.generation += 1
                    self.round.notify_all()
            else:
                while self.generation == gen:
                    self.round.wait(0.1)
        return BarrierMessage(resp=_ok())


def serve(store, port=0, worker_num=1, is_async=False, host="127.0.0.1", max_workers=16):
    """Start `service PS` on host:port (0: an ephemeral port).  Returns (grpc server, bound port, servicer)."""
    sv = PsServicer(store, worker_num, is_async)
    handlers = {name: grpc.unary_unary_rpc_method_handler(getattr(sv, name), request_deserializer=rq.FromString,
                                                          response_serializer=rs.SerializeToString)
                for name, (rq, rs) in RPCS.items()}
    opts = [("grpc.max_send_message_length", -1), ("grpc.max_receive_message_length", -1)]
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers), options=opts)
    server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler(SERVICE, handlers),))
    bound = server.add_insecure_port("%s:%d" % (host, port))
    server.start()
    return server, bound, sv


class PsClient:
    """net/PSClient.java's calls over the same wire (for tests and Python hosts)."""

    def __init__(self, target, host_name="worker"):
        opts = [("grpc.max_send_message_length", -1), ("grpc.max_receive_message_length", -1)]
        self.channel = grpc.insecure_channel(target, options=opts)
        self.meta = RequestMeta(host=host_name)
        self.call = {name: self.channel.unary_unary("/%s/%s" % (SERVICE, name), request_serializer=rq.SerializeToString,
                                                    response_deserializer=rs.FromString) for name, (rq, rs) in RPCS.items()}

    @staticmethod
    def _arr(m):
        return np.asarray(m.data, np.float32) if len(m.data) else None

    def get(self, key):
        r = self.call["get"](GetMessage(meta=self.meta, weights=Matrix(key=key)))
        return (self._arr(r.weights) if r.resp.ec == EC_OK else None), r.resp.ec

    def getList(self, keys):
        r = self.call["getList"](GetListMessage(meta=self.meta, weights=[Matrix(key=k) for k in keys]))
        return {w.key: self._arr(w) for w in r.weights}

    def upsertList(self, kv, replace=False):
        ws = [Matrix(key=k, row=len(v), cols=1, data=np.asarray(v, np.float32).ravel().tolist()) for k, v in kv.items()]
        r = self.call["upsertList"](UpdateListMessage(meta=self.meta, weights=ws, replace=replace))
        return {w.key: (self._arr(w), w.update) for w in r.weights}, r.resp.ec

    def push(self, key, grad, updater_name, is_async=False):
        g = np.asarray(grad, np.float32).ravel()
        r = self.call["push"](GradientMessage(meta=self.meta, gradient=Matrix(key=key, row=g.size, cols=1, data=g.tolist()),
                                              isAsync=is_async, updaterKey=updater_name))
        return r.resp.ec           # 0 (empty Resp) on success, exactly as the reference answers

    def barrier(self):
        return self.call["barrier"](BarrierMessage(meta=self.meta)).resp.ec

    def close(self):
        self.channel.close()
