"""The N>1 HIP path on ONE GPU: N worker/owner ranks run as N threads of this process, each with
its own sharded ps_store (rows id mod N == rank) and model, driving the real orchestration
(ps_amd/sharded.py ShardedWorker + HipBackend -> the C ABI).  The collectives are an in-process
stand-in that moves the device buffers through the host, so everything except the RCCL wire is the
product code: composite owner|row sort with nshards > 1, per-owner counts, serve_pull on every shard,
the sort-free push from N workers, the flat dense/wide reduction.  Expected values: the key-addressed
single-process simulation of the PS semantics used by the gloo test (net/PServer.java:164-214)."""
import ctypes as C
import threading

import numpy as np
import pytest

from test_sharded_gloo import CFG, SEED, STEPS, expected, make_batches

pytestmark = pytest.mark.gpu
ISOLATE_IN_SUBPROCESS = True      # tests/conftest.py: one python process per test, hard limit, one reported retry
f32 = np.float32


class Shared:
    def __init__(self, world):
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world, timeout=60)


_WEDGED = []          # a rank thread that never came back owns HIP-runtime state: later tests here would hang behind it


def run_ranks(target, argsets, deadline_s=90):
    """Start one daemon thread per rank, join them against ONE deadline, fail fast when a thread is stuck."""
    import time
    if _WEDGED:
        pytest.fail("skipped: rank threads of %s never returned" % _WEDGED[0])
    th = [threading.Thread(target=target, args=a, daemon=True) for a in argsets]
    for t in th:
        t.start()
    t_end = time.monotonic() + deadline_s
    for t in th:
        t.join(max(0.0, t_end - time.monotonic()))
    stuck = [i for i, t in enumerate(th) if t.is_alive()]
    if stuck:
        import faulthandler
        faulthandler.dump_traceback(all_threads=True)
        _WEDGED.append(getattr(target, "__name__", "rank threads"))
        pytest.fail("rank threads %s still running after %d s (stacks on stderr)" % (stuck, deadline_s))


class ThreadComm:
    """Blocking collectives between the rank threads; buffers are (device pointer, shape) pairs."""
    side_stream_handle = None

    def __init__(self, rank, shared, kv):
        from ps_amd import native as N
        self.rank, self.world, self.sh, self.kv, self.N = rank, shared.world, shared, kv, N
        self.keep = []

    # -- the parts of the comm contract that are no-ops without streams
    def side(self):
        from ps_amd.sharded import _NullCtx
        return _NullCtx()

    def side_wait_for(self, ev):
        pass

    def join_side(self, ready=None, tensors=()):
        pass

    def record_side(self):
        return None

    def record_done(self):
        return None

    def bind_thread(self):
        pass

    # -- host staging
    def _down(self, buf, dtype):
        ptr, shape = buf
        a = np.empty(shape, dtype)
        if a.size:
            self.N.check(self.N.lib().ps_dev_download(self.kv.h, a.ctypes.data, ptr, a.nbytes))
        return a

    def _up(self, a):
        p = C.c_void_p()
        self.N.check(self.N.lib().ps_dev_alloc(self.kv.h, max(a.nbytes, 4), C.byref(p)))
        if a.size:
            a = np.ascontiguousarray(a)
            self.N.check(self.N.lib().ps_dev_upload(self.kv.h, p, a.ctypes.data, a.nbytes))
        self.keep.append(p)
        return (p.value, a.shape)

    def _swap(self, mine):
        self.sh.slots[self.rank] = mine
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        self.sh.barrier.wait()
        return got

    def exchange_counts(self, counts, side=False):
        return [int(c[self.rank]) for c in self._swap(list(counts))]

    def exchange_counts_launch(self, counts):
        return self.exchange_counts(counts)

    def exchange_counts_complete(self, h):
        return h

    def all_to_all_v(self, send, send_counts, recv_counts, width, side=False):
        host = self._down(send, np.uint32 if width == 1 else f32)
        offs = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        parts = self._swap((host, offs))
        out = np.concatenate([h[o[self.rank]:o[self.rank + 1]] for h, o in parts])
        assert [len(h[o[self.rank]:o[self.rank + 1]]) for h, o in parts] == list(recv_counts)
        return self._up(out)

    def all_reduce_sum_async(self, buf):
        from ps_amd.sharded import _Done
        parts = self._swap(self._down(buf, f32))
        tot = parts[0].copy()
        for p in parts[1:]:
            tot = (tot + p).astype(f32)          # rank order (RCCL's order is its own; the test allows for it)
        self.N.check(self.N.lib().ps_dev_upload(self.kv.h, buf[0], tot.ctypes.data, tot.nbytes))
        return _Done()

    def barrier(self):
        self.sh.barrier.wait()

    def free(self):
        for p in self.keep:
            self.N.lib().ps_dev_free(self.kv.h, p)
        self.keep = []


def rank_main(rank, world, shared, is_async, pipelined, out, errs):
    try:
        import ps_amd
        from ps_amd.sharded import HipBackend, ShardedWorker
        F, D, V = CFG["F"], CFG["D"], CFG["V"]
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D, shard=rank, nshards=world)
        nctx = 3 if pipelined else 1
        gms = [ps_amd.WideDeepNN.buildModel(F, D, CFG["X"], CFG["fc"], CFG["wide"], store=kv, max_batch=CFG["B"]) for _ in range(nctx)]
        comm = ThreadComm(rank, shared, kv)
        wk = ShardedWorker(HipBackend(gms), comm, is_async=is_async)
        bs = [ps_amd.Batch(b["E"], b["X"], b["Y"], b["W"]) for b in make_batches(rank, STEPS)]
        if pipelined:
            wk.run(bs, STEPS)
        else:
            for b in bs:
                wk.step(b)
        kv.sync()
        rows = {}
        for f in range(F):
            ids = np.arange(rank, V, world)
            w = kv.get_rows(f, ids)
            for i, idv in enumerate(ids):
                rows[(f, int(idv))] = w[i]
        out[rank] = (rows, [kv.get("fc%d.weights" % l) for l in range(3)], [kv.get("fc%d.bias" % l) for l in range(3)],
                     kv.get_wide(np.arange(CFG["wide"])), kv.get("wide.bias"), kv.global_step())
        comm.free()
        for g in gms:
            g.close()
        kv.close()
    except BaseException:       # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        shared.barrier.abort()


@pytest.mark.parametrize("world,is_async,pipelined", [(2, False, False), (4, False, True), (4, True, False), (3, False, False)])
def test_n_ranks_on_one_gpu(orc, world, is_async, pipelined):
    shared = Shared(world)
    out, errs = [None] * world, []
    run_ranks(rank_main, [(r, world, shared, is_async, pipelined, out, errs) for r in range(world)])
    assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
    emb, fcW, fcb, ww, wb = expected(world, is_async)
    xav = orc.xavier_scale(1, CFG["D"])
    tol = 2e-5 * STEPS                   # same bound as the single-GPU step parity (FP32 GEMM order differs from the oracle's)
    touched = 0
    for r in range(world):
        rows, W, b, wide, wbias, gstep = out[r][:6]
        assert gstep == STEPS
        for (f, i), got in rows.items():
            assert i % world == r
            if (f, i) in emb:
                assert np.abs(got - emb[(f, i)][0]).max() <= tol, "rank %d emF%d.%d" % (r, f, i)
                touched += 1
            else:                         # never pulled by any worker: still the initial row, bit for bit
                np.testing.assert_array_equal(got, orc.init_rows(SEED, f, [i], CFG["D"], xav)[0])
        for l in range(3):
            assert np.abs(W[l] - fcW[l]).max() <= tol and np.abs(b[l] - fcb[l]).max() <= tol
        assert np.abs(wide - ww).max() <= tol and abs(wbias[0] - wb[0]) <= tol
        # replicated tensors are identical on every rank, bit for bit (same reduced gradient, same updater)
        for l in range(3):
            np.testing.assert_array_equal(W[l], out[0][1][l]); np.testing.assert_array_equal(b[l], out[0][2][l])
        np.testing.assert_array_equal(wide, out[0][3]); np.testing.assert_array_equal(wbias, out[0][4])
    assert touched > 0


# ---------------------------------------------------------------------------
# the library-driven step (ps_shard_step): same ranks-as-threads set-up, the collectives plugged into the
# C callback table (ps_comm_ops_t) -- so the C++ orchestration (count matrix, split sizes, buffer order) is
# exercised at N > 1 on one GPU; only the RCCL calls themselves are replaced.
# ---------------------------------------------------------------------------
class CallbackComm:
    def __init__(self, rank, shared, kv):
        from ps_amd import native as N
        self.rank, self.world, self.sh, self.kv, self.N = rank, shared.world, shared, kv, N
        self.ops = N.ps_comm_ops_t()
        self.ops.ctx, self.ops.nranks, self.ops.rank = None, self.world, rank
        self._ag = N.ALL_GATHER_FN(self.all_gather); self._a2a = N.ALL_TO_ALL_V_FN(self.all_to_all_v); self._ar = N.ALL_REDUCE_FN(self.all_reduce)
        self.ops.all_gather, self.ops.all_to_all_v, self.ops.all_reduce_sum_f32 = self._ag, self._a2a, self._ar
        self.err = None

    def _down(self, ptr, nbytes):
        a = np.empty(nbytes, np.uint8)
        if nbytes:
            self.N.check(self.N.lib().ps_dev_download(self.kv.h, a.ctypes.data, ptr, nbytes))
        return a

    def _up(self, ptr, a):
        if a.size:
            a = np.ascontiguousarray(a)
            self.N.check(self.N.lib().ps_dev_upload(self.kv.h, ptr, a.ctypes.data, a.nbytes))

    def _swap(self, mine):
        self.sh.slots[self.rank] = mine
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        self.sh.barrier.wait()
        return got

    def _guard(self, fn, stream):
        try:
            self.N.check(self.N.lib().ps_stream_sync(self.kv.h, stream))    # "enqueue on stream": here the stream is drained first
            fn()
            return 0
        except BaseException as e:             # noqa: BLE001 -- reported through the status code
            self.err = e
            self.sh.barrier.abort()
            return 500

    def all_gather(self, ctx, send, recv, nbytes, stream):
        def f():
            parts = self._swap(self._down(send, nbytes))
            self._up(recv, np.concatenate(parts))
        return self._guard(f, stream)

    def all_to_all_v(self, ctx, send, sc, recv, rc, eb, stream):
        def f():
            scl = [int(sc[i]) for i in range(self.world)]; rcl = [int(rc[i]) for i in range(self.world)]
            host = self._down(send, sum(scl) * eb)
            offs = np.concatenate([[0], np.cumsum(scl)]).astype(np.int64) * eb
            parts = self._swap((host, offs))
            pieces = [h[o[self.rank]:o[self.rank + 1]] for h, o in parts]
            assert [len(x) // eb for x in pieces] == rcl
            self._up(recv, np.concatenate(pieces) if pieces else np.zeros(0, np.uint8))
        return self._guard(f, stream)

    def all_reduce(self, ctx, buf, n, stream):
        def f():
            parts = self._swap(self._down(buf, n * 4).view(f32))
            tot = parts[0].copy()
            for p in parts[1:]:
                tot = (tot + p).astype(f32)
            self._up(buf, tot)
        return self._guard(f, stream)


def native_rank_main(rank, world, shared, is_async, out, errs, pipelined=False, wide_mode=None):
    try:
        import ps_amd
        from ps_amd.sharded import NativeWorker
        F, D, V = CFG["F"], CFG["D"], CFG["V"]
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D, shard=rank, nshards=world)
        # pipelined: True = two models, begin(t+1) on the prefetch stream before finish(t); "one" = ONE model, the next
        # step's begin slipped in before this step's push (ps_shard_step_finish_begin: what bench.py --gpus N runs)
        kw = {} if wide_mode is None else {"wide_grad_mode": wide_mode}
        gms = [ps_amd.WideDeepNN.buildModel(F, D, CFG["X"], CFG["fc"], CFG["wide"], store=kv, max_batch=CFG["B"], **kw) for _ in range(2 if pipelined is True else 1)]
        gm = gms[0]
        comm = CallbackComm(rank, shared, kv)
        wk = NativeWorker(gms, world, rank, ops=comm.ops, is_async=is_async)
        wk.selfcheck()                          # ps_comm_selfcheck over the plugged-in collectives (N > 1: real patterns)
        if pipelined == "one-dev":
            # device-resident batches: the next step's plan then runs on side chain 0 WHILE the step trains (early plan,
            # ps_shard.hip shard_plan_enqueue), released and joined by device-side flags
            bs = [ps_amd.DeviceBatch(kv, b["E"], b["X"], b["Y"], b["W"]) for b in make_batches(rank, STEPS)]
        else:
            bs = [ps_amd.Batch(b["E"], b["X"], b["Y"], b["W"]) for b in make_batches(rank, STEPS)]
        if pipelined:
            wk.run(bs, STEPS)                   # begin(t+1) on the prefetch stream before finish(t)
        else:
            for b in bs:
                wk.step(b)
        kv.sync()
        if comm.err is not None:
            raise comm.err
        rows = {}
        for f in range(F):
            ids = np.arange(rank, V, world)
            w = kv.get_rows(f, ids)
            for i, idv in enumerate(ids):
                rows[(f, int(idv))] = w[i]
        import ctypes as C
        from ps_amd import native as N
        st10 = (C.c_int64 * 10)(); N.check(N.lib().ps_shard_exchange_stats(gm.h, st10, 10))
        out[rank] = (rows, [kv.get("fc%d.weights" % l) for l in range(3)], [kv.get("fc%d.bias" % l) for l in range(3)],
                     kv.get_wide(np.arange(CFG["wide"])), kv.get("wide.bias"), kv.global_step(), int(st10[4]))
        for g in gms:
            g.close()
        kv.close()
    except BaseException:       # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        shared.barrier.abort()


@pytest.mark.parametrize("world,is_async", [(3, False), (4, True)])
def test_wide_part_as_worker_slots_equals_dense_vectors(world, is_async):
    """The all-reduced buffer's wide part as per-worker slots [gbar_w | touched_w, 24 bits to a float] (kernels_emb.h
    WideUpdArgs.slots, the default) against rounds 2-4's dense G | C vectors (ps_tune_set("wide_slots", 0)): the sum of slots of which
    all but one are zero is exact in any order, and every rank rebuilds G[k] in rank order -- the order this test's all-reduce adds
    in -- so every table is bit-identical, with fewer floats on the wire (net/PServer.java:164-214: mean over the workers that
    pushed the key)."""
    from ps_amd import native as N
    res = []
    for slots in (0, 1):
        N.lib().ps_tune_set(b"wide_slots", slots)
        try:
            shared = Shared(world)
            out, errs = [None] * world, []
            run_ranks(native_rank_main, [(r, world, shared, is_async, out, errs, "one") for r in range(world)])
            assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
            res.append(out)
        finally:
            N.lib().ps_tune_set(b"wide_slots", 1)
    for r in range(world):
        a, b = res[0][r], res[1][r]
        for k in a[0]:
            np.testing.assert_array_equal(a[0][k], b[0][k])
        for i in (1, 2):
            for x, y in zip(a[i], b[i]):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a[3], b[3]); np.testing.assert_array_equal(a[4], b[4])
        dense_elems = sum(x.size for x in a[1]) + sum(x.size for x in a[2])
        assert a[6] == STEPS * 4 * (dense_elems + 2 * CFG["wide"] + 1), (a[6], dense_elems)
        assert b[6] == STEPS * 4 * (dense_elems + 1 + world * (1 + (CFG["wide"] + 23) // 24)), (b[6], dense_elems)
        assert b[6] < a[6]


@pytest.mark.parametrize("world,is_async,pipelined", [(2, False, False), (4, False, True), (3, True, False), (2, True, True), (3, False, "one"), (2, True, "one"),
                                                        (3, False, "one-dev"), (2, True, "one-dev"), (4, False, "one-dev")])
def test_library_driven_step_n_ranks_on_one_gpu(orc, world, is_async, pipelined):
    shared = Shared(world)
    out, errs = [None] * world, []
    run_ranks(native_rank_main, [(r, world, shared, is_async, out, errs, pipelined) for r in range(world)])
    assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
    emb, fcW, fcb, ww, wb = expected(world, is_async)
    xav = orc.xavier_scale(1, CFG["D"])
    tol = 2e-5 * STEPS
    touched = 0
    for r in range(world):
        rows, W, b, wide, wbias, gstep = out[r][:6]
        assert gstep == STEPS
        for (f, i), got in rows.items():
            if (f, i) in emb:
                assert np.abs(got - emb[(f, i)][0]).max() <= tol, "rank %d emF%d.%d" % (r, f, i)
                touched += 1
            else:
                np.testing.assert_array_equal(got, orc.init_rows(SEED, f, [i], CFG["D"], xav)[0])
        for l in range(3):
            assert np.abs(W[l] - fcW[l]).max() <= tol and np.abs(b[l] - fcb[l]).max() <= tol
            np.testing.assert_array_equal(W[l], out[0][1][l]); np.testing.assert_array_equal(b[l], out[0][2][l])
        assert np.abs(wide - ww).max() <= tol and abs(wbias[0] - wb[0]) <= tol
        np.testing.assert_array_equal(wide, out[0][3])
    assert touched > 0


def test_library_driven_step_equals_python_driven_step():
    """N = 2 ranks on one GPU: ps_shard_step (C++ orchestration) == ShardedWorker (Python orchestration), bit for bit."""
    world = 2
    res = []
    for native in (False, True):
        shared = Shared(world)
        out, errs = [None] * world, []
        tgt = native_rank_main if native else rank_main
        args = (lambda r: (r, world, shared, False, out, errs)) if native else (lambda r: (r, world, shared, False, False, out, errs))
        run_ranks(tgt, [args(r) for r in range(world)])
        assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
        res.append(out)
    for r in range(world):
        a, b = res[0][r], res[1][r]
        for k in a[0]:
            np.testing.assert_array_equal(a[0][k], b[0][k])
        for i in (1, 2):
            for x, y in zip(a[i], b[i]):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a[3], b[3]); np.testing.assert_array_equal(a[4], b[4])


# ---------------------------------------------------------------------------
# BASELINE configs[4] semantics at small scale: multi-hot bags (incl. empty ones), FTRL on every embedding
# row, async push, N ranks -- through ps_shard_step.  Expected: the PS semantics re-played on the host with the
# single-GPU split form as each worker's gradient engine (forward + backward(apply = false) on a full-table
# store loaded with the step's global weights), then mean / per-push updates with the oracle's updaters.
# ---------------------------------------------------------------------------
BF, BV, BD, BX, BFC, BB = 3, 23, 4, 2, [6, 4, 1], 10


def bag_batches(rank, steps):
    rng = np.random.default_rng(500 + rank)
    out = []
    for _ in range(steps):
        lens = rng.integers(0, 5, size=BB * BF)
        lens[2] = 0
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = rng.integers(0, BV, size=int(offsets[-1])).astype(np.int64)
        ids[:3] = [1, 2, 3][:len(ids[:3])]                       # keys several workers push in the same step
        out.append({"E": ids, "offsets": offsets, "X": rng.standard_normal((BB, BX)).astype(f32), "Y": (rng.random(BB) < 0.4).astype(f32)})
    return out


def bag_rank_main(rank, world, shared, is_async, out, errs):
    try:
        import ps_amd
        from ps_amd.sharded import NativeWorker
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([BV] * BF, BD, shard=rank, nshards=world)
        kv.set_updater("emF", ps_amd.FtrlUpdater())
        gm = ps_amd.DNN.buildModel(BF, BD, BX, BFC, store=kv, max_batch=BB, max_nnz=BB * BF * 5)
        comm = CallbackComm(rank, shared, kv)
        wk = NativeWorker(gm, world, rank, ops=comm.ops, is_async=is_async)
        for b in bag_batches(rank, STEPS):
            wk.step(ps_amd.Batch(b["E"], b["X"], b["Y"], None, b["offsets"]))
        kv.sync()
        if comm.err is not None:
            raise comm.err
        rows = {}
        for f in range(BF):
            ids = np.arange(rank, BV, world)
            w, z, n = kv.get_rows(f, ids), kv.get_rows(f, ids, 1), kv.get_rows(f, ids, 2)
            for i, idv in enumerate(ids):
                rows[(f, int(idv))] = (w[i], z[i], n[i])
        out[rank] = (rows, [kv.get("fc%d.weights" % l) for l in range(3)], [kv.get("fc%d.bias" % l) for l in range(3)])
        gm.close(); kv.close()
    except BaseException:       # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        shared.barrier.abort()


def bag_expected(orc, world, is_async):
    import ps_amd
    engines = []
    for w in range(world):                      # one full-table store + model per worker: its gradient engine
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([BV] * BF, BD)
        engines.append((kv, ps_amd.DNN.buildModel(BF, BD, BX, BFC, store=kv, max_batch=BB, max_nnz=BB * BF * 5)))
    kv0 = engines[0][0]
    allid = np.arange(BV)
    W = [kv0.get_rows(f, allid) for f in range(BF)]
    Z = [np.zeros_like(x) for x in W]; Nn = [np.zeros_like(x) for x in W]
    fcW = [kv0.get("fc%d.weights" % l) for l in range(3)]; fcb = [kv0.get("fc%d.bias" % l) for l in range(3)]
    fcS = [[np.zeros_like(w), np.zeros_like(w), np.zeros_like(b), np.zeros_like(b)] for w, b in zip(fcW, fcb)]
    data = [bag_batches(w, STEPS) for w in range(world)]
    for step in range(STEPS):
        pushes, dense = {}, None
        for w, (kv, gm) in enumerate(engines):
            for f in range(BF):
                kv.put_rows(f, allid, W[f])
            for l in range(3):
                kv.put("fc%d.weights" % l, fcW[l]); kv.put("fc%d.bias" % l, fcb[l])
            b = data[w][step]
            gm.forward({"E": b["E"], "X": b["X"], "Y": b["Y"], "offsets": b["offsets"]})
            gm.backward()
            for f in range(BF):
                ids, g = gm.emb_grads(f)
                for i, idv in enumerate(ids):
                    pushes.setdefault((f, int(idv)), []).append(g[i])           # worker order
            d = np.concatenate([np.concatenate([gm.fc_grad(l), gm.fc_grad(l, True)]) for l in range(3)])
            dense = d if dense is None else (dense + d).astype(f32)
        for (f, i), gs in pushes.items():
            if is_async:
                for g in gs:
                    W[f][i], Z[f][i], Nn[f][i], _ = orc.ftrl_update(W[f][i], g, Z[f][i], Nn[f][i])
            else:
                S = gs[0].copy()
                for g in gs[1:]:
                    S = (g + S).astype(f32)
                W[f][i], Z[f][i], Nn[f][i], _ = orc.ftrl_update(W[f][i], (S / f32(len(gs))).astype(f32), Z[f][i], Nn[f][i])
        off = 0
        for l in range(3):
            nw, nb = fcW[l].size, fcb[l].size
            gw = (dense[off:off + nw] / f32(world)).astype(f32); off += nw
            gb = (dense[off:off + nb] / f32(world)).astype(f32); off += nb
            fcW[l], fcS[l][0], fcS[l][1] = orc.adam_update(fcW[l], gw, fcS[l][0], fcS[l][1])
            fcb[l], fcS[l][2], fcS[l][3] = orc.adam_update(fcb[l], gb, fcS[l][2], fcS[l][3])
    for kv, gm in engines:
        gm.close(); kv.close()
    return W, Z, Nn, fcW, fcb


@pytest.mark.parametrize("world,is_async", [(3, True), (2, False)])
def test_multi_hot_ftrl_ranks(orc, world, is_async):
    shared = Shared(world)
    out, errs = [None] * world, []
    run_ranks(bag_rank_main, [(r, world, shared, is_async, out, errs) for r in range(world)])
    assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
    W, Z, Nn, fcW, fcb = bag_expected(orc, world, is_async)
    moved = 0
    for r in range(world):
        rows, gW, gb = out[r]
        for (f, i), (w, z, n) in rows.items():
            assert i % world == r
            np.testing.assert_array_equal(w, W[f][i], err_msg="rank %d emF%d.%d w" % (r, f, i))
            np.testing.assert_array_equal(z, Z[f][i]); np.testing.assert_array_equal(n, Nn[f][i])
            moved += int(np.any(n != 0))
        for l in range(3):
            np.testing.assert_array_equal(gW[l], fcW[l]); np.testing.assert_array_equal(gb[l], fcb[l])
    assert moved > 10


def test_intended_wide_mode_n_ranks():
    """wide_grad_mode = intended at N = 3 ranks through ps_shard_step: every rank ends with the SAME wide table (the owner side
    applies the mean over the workers that pushed a key, from the all-reduced G / C pair), it differs from the compat run's,
    and the embedding / FC results are those of the compat run only up to the wide part's influence (not compared)."""
    from ps_amd import native as N
    world = 3
    res = []
    for mode in (N.PS_GRAD_INTENDED, None):
        shared = Shared(world)
        out, errs = [None] * world, []
        run_ranks(native_rank_main, [(r, world, shared, False, out, errs, "one", mode) for r in range(world)])
        assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
        for r in range(1, world):
            np.testing.assert_array_equal(out[r][3], out[0][3]); np.testing.assert_array_equal(out[r][4], out[0][4])
            for l in range(3):
                np.testing.assert_array_equal(out[r][1][l], out[0][1][l])
        res.append(out[0])
    assert np.abs(res[0][3]).max() > 0 and not np.array_equal(res[0][3], res[1][3])
