"""Key-sharded parameter server over N GPUs: the host-side wire of
net/PSRouterClient.java + net/PServer.java, with RCCL collectives in place of
one gRPC per key.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).
Every rank is both a worker (its own minibatch) and the owner of the
embedding rows with id mod N == rank (net/Mod.java routing).  Per step:

    getList   counts all-to-all, row-id all-to-all-v, rows all-to-all-v back
    train     forward/backward on the pulled rows (HIP, ps_shard_forward_backward)
    push      per-key gradients all-to-all-v to their owners
    psUpdate  owner: mean over the pushing workers + fused Adam/Ftrl (BSP), or
              one update per push in worker order (async, -DisPsAsync=1)
    dense     ONE all-reduce of [fc weights+biases | wide G | wide C | wide.bias]
    barrier   completion of the collectives; globalStep++

The orchestration (`ShardedWorker.step`) is backend-agnostic: the product
backend is `HipBackend` (the C ABI of include/ps_native.h); the world_size-2
gloo tests drive the same orchestration with a CPU backend built on the
oracle (tests/ only).
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

from . import native as N


# ---------------------------------------------------------------------------
# communication
# ---------------------------------------------------------------------------
class _Done:
    def wait(self):
        pass


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class LocalComm:
    """world_size 1: every collective is the identity (used by single-GPU tests)."""
    rank, world = 0, 1
    side_stream_handle = None

    def side(self):
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

    def exchange_counts(self, counts, side=False):
        return list(counts)

    def exchange_counts_launch(self, counts):
        return list(counts)

    def exchange_counts_complete(self, h):
        return h

    def all_to_all_v(self, send, send_counts, recv_counts, width, side=False):
        return send

    def all_reduce_sum_async(self, buf):
        return _Done()

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed (nccl = RCCL on ROCm, gloo on CPU).  Buffers are torch tensors.

    Three communicators so that independent exchanges do not queue behind each other:
    the main one (rows / gradients, on the training stream), a `side` one for the
    weight-independent prefetch of the next step's key lists (on its own stream), and one
    for the dense all-reduce, which runs beside the gradient all-to-all."""

    def __init__(self, dist, torch, device, overlap=True):
        self.dist, self.torch, self.device = dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.cuda = device.type == "cuda"
        self.side_group = dist.new_group(list(range(self.world))) if overlap else None
        self.ar_group = dist.new_group(list(range(self.world))) if overlap else None
        # high priority: the prefetch is a chain of tiny kernels that must not queue behind the training step's GEMMs
        self.side_stream = torch.cuda.Stream(device, priority=-1) if (overlap and self.cuda) else None
        self.side_stream_handle = self.side_stream.cuda_stream if self.side_stream is not None else None

    def side(self):
        """Context of the prefetch: torch's current stream becomes the side stream."""
        return self.torch.cuda.stream(self.side_stream) if self.side_stream is not None else _NullCtx()

    def side_wait_for(self, ev):
        if ev is not None and self.side_stream is not None:
            self.side_stream.wait_event(ev)

    def record_side(self):
        """Event at the current tail of the side stream (call inside side())."""
        if self.side_stream is None:
            return None
        ev = self.torch.cuda.Event()
        ev.record(self.side_stream)
        return ev

    def join_side(self, ready=None, tensors=()):
        """The training stream continues after the prefetch of THIS step (not whatever later
        prefetch is already queued behind it on the side stream)."""
        if self.side_stream is None:
            return
        main = self.torch.cuda.current_stream(self.device)
        if ready is not None:
            main.wait_event(ready)
        else:
            main.wait_stream(self.side_stream)
        for t in tensors:
            if hasattr(t, "record_stream"):
                t.record_stream(main)

    def bind_thread(self):
        if self.cuda:
            self.torch.cuda.set_device(self.device)

    def record_done(self):
        if self.side_stream is None:
            return None
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    def exchange_counts(self, counts, side=False):
        t = self.torch
        send = t.tensor(counts, dtype=t.int64, device=self.device)
        recv = t.empty_like(send)
        self.dist.all_to_all_single(recv, send, group=self.side_group if side else None)
        return [int(x) for x in recv.tolist()]

    def exchange_counts_launch(self, counts):
        """Counts exchange on the side communicator without a host wait: returns a handle for
        exchange_counts_complete.  (CPU/gloo: the exchange itself is synchronous.)"""
        if self.side_stream is None:
            return self.exchange_counts(counts, side=True)
        t = self.torch
        if not hasattr(self, "_cnt_ring"):
            self._cnt_ring = [(t.empty(self.world, dtype=t.int64).pin_memory(), t.empty(self.world, dtype=t.int64).pin_memory(),
                               t.cuda.Event()) for _ in range(4)]
            self._cnt_next = 0
        sp, rp, ev = self._cnt_ring[self._cnt_next % len(self._cnt_ring)]
        self._cnt_next += 1
        for i, c in enumerate(counts):
            sp[i] = int(c)
        send = sp.to(self.device, non_blocking=True)
        recv = t.empty_like(send)
        self.dist.all_to_all_single(recv, send, group=self.side_group)
        rp.copy_(recv, non_blocking=True)
        ev.record(t.cuda.current_stream(self.device))
        return (rp, ev, send, recv)

    def exchange_counts_complete(self, h):
        if isinstance(h, list):
            return h
        rp, ev = h[0], h[1]
        ev.synchronize()
        return [int(x) for x in rp.tolist()]

    def all_to_all_v(self, send, send_counts, recv_counts, width, side=False):
        """send: [sum(send_counts), width] (or 1-D when width == 1), grouped by destination rank."""
        t = self.torch
        n_recv = int(sum(recv_counts))
        shape = (n_recv,) if send.dim() == 1 else (n_recv, width)
        recv = t.empty(shape, dtype=send.dtype, device=self.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                                    group=self.side_group if side else None)
        return recv

    def all_reduce_sum_async(self, buf):
        return self.dist.all_reduce(buf, group=self.ar_group, async_op=True)     # sum

    def barrier(self):
        self.dist.barrier()


# ---------------------------------------------------------------------------
# the orchestration: one worker/owner step
# ---------------------------------------------------------------------------
class Prepared:
    """What the weight-independent half of a step produced (PSRouterClient.getList fan-out)."""
    __slots__ = ("ctx", "counts", "rcounts", "recv_rows", "n_recv", "ready")

    def __init__(self, ctx, counts, rcounts, recv_rows, ready=None):
        self.ctx, self.counts, self.rcounts, self.recv_rows, self.ready = ctx, counts, rcounts, recv_rows, ready
        self.n_recv = int(sum(rcounts))


class ShardedWorker:
    """prepare(batch) -> finish(prepared) is one BSP step.  `run` software-pipelines them: the key
    lists of step t+1 (sort, unique, counts and row-id exchange: nothing that reads a weight) are
    prepared on the side stream/communicator while step t trains, so the two host round trips of
    all-to-all-v (split sizes) are off the critical path.  The backend keeps `nctx` plan contexts."""

    def __init__(self, backend, comm, is_async=False):
        self.be, self.comm, self.is_async = backend, comm, is_async
        self.nctx = getattr(backend, "nctx", 1)
        self.done = [None] * self.nctx
        self.turn = 0

    def prepare_launch(self, batch):
        """P0: enqueue the key-list kernels of `batch` on the side stream; no host wait."""
        be, comm = self.be, self.comm
        ctx = self.turn % self.nctx
        self.turn += 1
        with comm.side():
            comm.side_wait_for(self.done[ctx])            # the context's previous step no longer reads its buffers
            be.plan_launch(batch, comm.world, ctx, comm.side_stream_handle)
        return ctx

    def prepare_counts(self, ctx):
        """P1: per-owner counts back to the host (the plan's event), counts exchange enqueued."""
        be, comm = self.be, self.comm
        with comm.side():
            counts, send_rows = be.plan_finish(ctx)
            h = comm.exchange_counts_launch(counts)
        return (ctx, counts, send_rows, h)

    def prepare_complete(self, pc):
        """P2: split sizes known -> the row-id all-to-all-v (side communicator)."""
        comm = self.comm
        if not isinstance(pc, tuple):
            pc = self.prepare_counts(pc)
        ctx, counts, send_rows, h = pc
        with comm.side():
            rcounts = comm.exchange_counts_complete(h)
            recv_rows = comm.all_to_all_v(send_rows, counts, rcounts, 1, side=True)
            ready = comm.record_side()
        return Prepared(ctx, counts, rcounts, recv_rows, ready)

    def prepare(self, batch):
        return self.prepare_complete(self.prepare_counts(self.prepare_launch(batch)))

    def finish(self, p, want_loss=True):
        be, comm = self.be, self.comm
        comm.join_side(p.ready, (p.recv_rows,))
        rows_out = be.serve_pull(p.recv_rows, p.n_recv)                          # PServer.getList
        cache = comm.all_to_all_v(rows_out, p.rcounts, p.counts, be.D)           # worker cache
        loss = be.forward_backward(p.ctx, cache, want_loss)                      # Model.train on the cached rows
        ar = comm.all_reduce_sum_async(be.flat_grad(p.ctx))                      # dense tensors + wide keys, beside the push
        grads = be.grads(p.ctx)                                                  # what PSClient.push sends
        recv_grads = comm.all_to_all_v(grads, p.counts, p.rcounts, be.D)
        be.apply_push(p.recv_rows, recv_grads, p.n_recv, p.rcounts, self.is_async)          # PServer.push + psUpdate
        ar.wait()
        be.apply_flat(p.ctx, comm.world)
        self.done[p.ctx] = comm.record_done()
        return loss

    def step(self, batch, want_loss=True):
        return self.finish(self.prepare(batch), want_loss)

    def run(self, batches, steps, want_loss=False, threaded=False):
        """`steps` software-pipelined steps over batches[i % len]; returns the last loss (None unless
        want_loss).  With 3 plan contexts the host never waits: while step i is enqueued, step i+2's
        key lists are sorted (P0), then its counts exchanged (P1), and step i+1's row ids travel (P2):

            iteration i:   P0(i+2) | finish(i) | P1(i+2)  P2(i+1)

        With 2 contexts P1+P2 of step i+1 follow finish(i) (one blocking counts round trip); with 1 the
        steps run back to back.  threaded: prepare() runs one step ahead in its own host thread (each
        communicator still driven by exactly one thread); measured slower than the in-line pipeline on
        MI355X/ROCm 7 (the two threads contend inside the HIP runtime), kept for hosts where it is not."""
        if steps <= 0:
            return None
        if threaded and self.nctx > 1:
            return self._run_threaded(batches, steps, want_loss)
        nb = len(batches)
        loss = None
        if self.nctx >= 3:
            cur = self.prepare(batches[0])
            nxt = self.prepare_counts(self.prepare_launch(batches[1 % nb])) if steps > 1 else None
            for i in range(steps):
                far = self.prepare_launch(batches[(i + 2) % nb]) if i + 2 < steps else None      # P0(i+2)
                loss = self.finish(cur, want_loss)                                                # (enqueue only unless want_loss)
                far = self.prepare_counts(far) if far is not None else None                       # P1(i+2)
                cur = self.prepare_complete(nxt) if nxt is not None else None                     # P2(i+1)
                nxt = far
            return loss
        p = self.prepare(batches[0])
        for i in range(steps):
            more = i + 1 < steps and self.nctx > 1
            if more:
                nxt = self.prepare_launch(batches[(i + 1) % nb])      # sort/unique of step i+1 beside step i
            loss = self.finish(p, want_loss)
            if more:
                p = self.prepare_complete(nxt)
            elif i + 1 < steps:
                p = self.prepare(batches[(i + 1) % nb])
        return loss

    def _run_threaded(self, batches, steps, want_loss):
        import queue
        import threading
        ready, free = queue.Queue(), queue.Queue()

        def producer():
            try:
                self.comm.bind_thread()
                for i in range(steps):
                    if i >= self.nctx and free.get(timeout=300) is None:     # a context is free once its step is enqueued
                        return
                    ready.put(self.prepare(batches[i % len(batches)]))
            except BaseException as e:      # noqa: BLE001 -- handed to the training thread
                ready.put(e)

        th = threading.Thread(target=producer, name="ps-prefetch", daemon=True)
        th.start()
        loss = None
        try:
            for _ in range(steps):
                p = ready.get(timeout=300)
                if isinstance(p, BaseException):
                    raise p
                loss = self.finish(p, want_loss)
                free.put(p.ctx)
        finally:
            free.put(None)
            th.join(timeout=300)
        return loss

    def step_timed(self, batch, sync, acc):
        """The same step with a host-side stopwatch around every phase (sync() between phases):
        measurement only -- it serialises what the real step overlaps."""
        be, comm = self.be, self.comm
        t = [time.perf_counter()]

        def lap(name):
            sync()
            t.append(time.perf_counter())
            acc[name] = acc.get(name, 0.0) + (t[-1] - t[-2])

        ctx = self.turn % self.nctx
        self.turn += 1
        counts, send_rows = be.plan(batch, comm.world, ctx, None); lap("plan")
        rcounts = comm.exchange_counts(counts); lap("a2a_counts")
        recv_rows = comm.all_to_all_v(send_rows, counts, rcounts, 1); lap("a2a_ids")
        n = int(sum(rcounts))
        rows_out = be.serve_pull(recv_rows, n); lap("serve_pull")
        cache = comm.all_to_all_v(rows_out, rcounts, counts, be.D); lap("a2a_rows")
        be.forward_backward(ctx, cache, False); lap("forward_backward")
        grads = be.grads(ctx)
        recv_grads = comm.all_to_all_v(grads, counts, rcounts, be.D); lap("a2a_grads")
        be.apply_push(recv_rows, recv_grads, n, rcounts, self.is_async); lap("apply_push")
        comm.all_reduce_sum_async(be.flat_grad(ctx)).wait(); lap("allreduce_flat")
        be.apply_flat(ctx, comm.world); lap("apply_flat")
        self.done[ctx] = comm.record_done()


# ---------------------------------------------------------------------------
# product backend: the HIP library
# ---------------------------------------------------------------------------
class _DevView:
    """Zero-copy view of a device buffer owned by libps_amd for torch.as_tensor."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipBackend:
    """Device-side halves through the C ABI.  With torch: buffers cross as torch tensors
    (zero-copy views of the library's device memory); without (LocalComm): raw pointers.
    `models`: one ps_model per plan context, all on the same store (they share every weight;
    each has its own batch staging, key lists and activations)."""

    def __init__(self, models, torch=None, device=None):
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        self.nctx = len(self.models)
        self.kv = self.models[0].store
        self.D = self.models[0].D
        self.torch, self.device = torch, device
        self._pull_buf, self._pull_cap = None, 0

    def _tensor(self, ptr, shape, typestr, dtype):
        if self.torch is None:
            return (ptr, shape)
        if int(np.prod(shape)) == 0:
            return self.torch.empty(shape, dtype=dtype, device=self.device)
        return self.torch.as_tensor(_DevView(ptr, shape, typestr), device=self.device)

    @staticmethod
    def _ptr(buf):
        return buf[0] if isinstance(buf, tuple) else buf.data_ptr()

    def plan_launch(self, batch, world, ctx=0, stream=None):
        self._world = world
        N.check(N.lib().ps_shard_plan_launch(self.models[ctx].h, C.byref(batch.c), world, stream))

    def plan_finish(self, ctx=0):
        counts = (C.c_int64 * self._world)()
        rows = C.c_void_p()
        nu = C.c_int64()
        N.check(N.lib().ps_shard_plan_finish(self.models[ctx].h, counts, C.byref(rows), C.byref(nu)))
        t = self.torch
        return list(counts), self._tensor(rows.value, (nu.value,), "<i4", None if t is None else t.int32)

    def plan(self, batch, world, ctx=0, stream=None):
        self.plan_launch(batch, world, ctx, stream)
        return self.plan_finish(ctx)

    def serve_pull(self, recv_rows, n):
        t = self.torch
        if t is None:
            if n > self._pull_cap:           # grow-only: no device allocation inside a steady-state step
                if self._pull_buf is not None:
                    N.lib().ps_dev_free(self.kv.h, self._pull_buf)
                self._pull_cap = max(n + n // 2, 1024)
                self._pull_buf = C.c_void_p()
                N.check(N.lib().ps_dev_alloc(self.kv.h, self._pull_cap * self.D * 4, C.byref(self._pull_buf)))
            buf = (self._pull_buf.value, (n, self.D))
        else:
            buf = t.empty((n, self.D), dtype=t.float32, device=self.device)
        N.check(N.lib().ps_shard_serve_pull(self.kv.h, self._ptr(recv_rows), n, self._ptr(buf)))
        return buf

    def forward_backward(self, ctx, cache, want_loss=True):
        loss = C.c_float()
        N.check(N.lib().ps_shard_forward_backward(self.models[ctx].h, self._ptr(cache), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def grads(self, ctx):
        g = C.c_void_p()
        nu = C.c_int64()
        N.check(N.lib().ps_shard_grads(self.models[ctx].h, C.byref(g), C.byref(nu)))
        t = self.torch
        return self._tensor(g.value, (nu.value, self.D), "<f4", None if t is None else t.float32)

    def apply_push(self, recv_rows, recv_grads, n, peer_counts, is_async):
        pc = (C.c_int64 * len(peer_counts))(*peer_counts)
        N.check(N.lib().ps_shard_apply_push(self.kv.h, self._ptr(recv_rows), self._ptr(recv_grads), n, pc, len(peer_counts), int(is_async)))

    def flat_grad(self, ctx):
        f = C.c_void_p()
        n = C.c_int64()
        N.check(N.lib().ps_shard_flat_grad(self.models[ctx].h, C.byref(f), C.byref(n)))
        t = self.torch
        return self._tensor(f.value, (n.value,), "<f4", None if t is None else t.float32)

    def apply_flat(self, ctx, world):
        N.check(N.lib().ps_shard_apply_flat(self.models[ctx].h, world))


# ---------------------------------------------------------------------------
# the library-driven step: ONE C call (ps_shard_step) per step, RCCL bound inside libps_amd.so
# ---------------------------------------------------------------------------
class NativeWorker:
    """ps_shard_step over a ps_comm_ops_t.  `ops` None: the RCCL communicators (3) are created from `id256` (384 bytes)
    (made by rank 0 with NativeWorker.unique_id() and handed to every rank by the host).
    `models`: one ps_model, or two on the same store -- then `run` begins step t+1 (its key lists and the
    counts all-gather, on the prefetch stream) before it finishes step t."""

    def __init__(self, models, nranks, rank, id256=None, ops=None, is_async=False):
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        self.kv, self.is_async = self.models[0].store, is_async
        self._own = ops is None
        if ops is None:
            ops = N.ps_comm_ops_t()
            N.check(N.lib().ps_comm_rccl_create(self.kv.h, nranks, rank, id256, C.byref(ops)))
        self.ops = ops

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(384)
        N.check(N.lib().ps_comm_rccl_unique_id(buf))
        return buf.raw

    def selfcheck(self):
        """Every collective of the table once on known patterns (collective call: all ranks)."""
        N.check(N.lib().ps_comm_selfcheck(self.kv.h, C.byref(self.ops)))

    def step(self, batch, want_loss=True):
        loss = C.c_float()
        N.check(N.lib().ps_shard_step(self.models[0].h, C.byref(batch.c), C.byref(self.ops), int(self.is_async), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def begin(self, k, batch, side=True):
        N.check(N.lib().ps_shard_step_begin(self.models[k].h, C.byref(batch.c), C.byref(self.ops), int(side)))

    def finish(self, k, want_loss=False):
        loss = C.c_float()
        N.check(N.lib().ps_shard_step_finish(self.models[k].h, C.byref(self.ops), int(self.is_async), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def finish_begin(self, k, next_batch, want_loss=False):
        """finish of the running step with the next step's begin slipped in before its push (one model)."""
        loss = C.c_float()
        N.check(N.lib().ps_shard_step_finish_begin(self.models[k].h, C.byref(self.ops), int(self.is_async),
                                                   C.byref(next_batch.c) if next_batch is not None else None,
                                                   C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def run(self, batches, steps, want_loss=False):
        nb, nm = len(batches), len(self.models)
        loss = None
        if nm < 2:
            # software pipeline on ONE stream and ONE model: the plan of step i+1 (no weights read) is enqueued between
            # step i's backward and its push, so the host's wait for the counts overlaps GPU work
            if steps > 0:
                self.begin(0, batches[0], side=False)
            trace = [] if (os.environ.get("PS_BENCH_DEBUG") and steps <= 64) else None
            for i in range(steps):
                if trace is not None:
                    trace.append(time.perf_counter())
                loss = self.finish_begin(0, batches[(i + 1) % nb] if i + 1 < steps else None, want_loss and i + 1 == steps)
            if trace:
                trace.append(time.perf_counter())
                print("[ps_shard_step calls, us] " + " ".join("%.0f" % (1e6 * (b - a)) for a, b in zip(trace, trace[1:])), file=sys.stderr, flush=True)
            return loss
        if steps > 0:
            self.begin(0, batches[0])
        for i in range(steps):
            if i + 1 < steps:
                self.begin((i + 1) % 2, batches[(i + 1) % nb])      # step i+1's key lists beside step i's training
            loss = self.finish(i % 2, want_loss)
        return loss

    def close(self):
        if self._own and self.ops is not None:
            N.lib().ps_comm_rccl_destroy(C.byref(self.ops))
        self.ops = None


# ---------------------------------------------------------------------------
# bench.py --gpus N (N > 1): BASELINE configs[2]
# ---------------------------------------------------------------------------
# A multi-GPU run that wedges costs the driver its whole timeout and yields nothing.  The bench therefore runs in STAGES,
# each more conservative than the one before, and a watchdog thread moves on when a stage makes no progress:
#   stage 0  rows and gradients over MAPPED PEER MEMORY (round 6: one launch of stores + flags per exchange instead of a grouped
#            ncclSend / ncclRecv; ps_native.h ps_shard_mapped_info), everything else as stage 1.  Its set-up checks the wire on
#            patterns and every rank falls back to RCCL's all-to-all-v together when any mapping or any word fails (the line says so)
#   stage 1  rounds 3-5's default: every collective through RCCL, device-side joins, key lists + all-reduce on side streams with
#            their own communicators
#   stage 2  events only (dev_wait = 0, end_wait = 0), everything on the training stream with ONE communicator
#   stage 3  the torch.distributed wire (ShardedWorker), the library only runs the device-side halves
# "Moving on" = the process replaces itself (os.execve) with the same command line and PS_BENCH_STAGE + 1: a hung
# collective or a spinning stream cannot be recovered from inside the process.  Every rank does this on its own
# watchdog; they meet again in init_process_group on the next stage's rendezvous port.  The stage a line was measured
# on is in config["stage"], the reason for leaving the earlier ones in config["stage_history"].
STAGES = ["rows + gradients over mapped peer memory (hipIpc; stores + flags), id blocks + all-reduce through RCCL (device-side joins, 3 communicators)",
          "every collective through RCCL (device-side joins, 3 communicators, key lists + all-reduce on side streams)",
          "events only, one communicator, one stream (dev_wait=0, end_wait=0, shard_overlap=0)",
          "torch.distributed wire (ShardedWorker)"]


class Watchdog:
    """kick() marks progress; if none for `limit` seconds the process re-executes itself on the next stage."""

    def __init__(self, stage, limit, rank):
        import threading
        self.stage, self.limit, self.rank = stage, limit, rank
        self.last = time.monotonic()
        self.what = "start"
        self.off = False
        self.t = threading.Thread(target=self._run, name="ps-bench-watchdog", daemon=True)
        self.t.start()

    def kick(self, what, limit=None):
        self.last = time.monotonic()
        self.what = what
        if limit is not None:
            self.limit = limit

    def stop(self):
        self.off = True

    def _run(self):
        while not self.off:
            time.sleep(0.5)
            if not self.off and time.monotonic() - self.last > self.limit:
                next_stage(self.stage, "no progress for %d s in '%s'" % (self.limit, self.what), self.rank)


def next_stage(stage, why, rank):
    import sys
    hist = os.environ.get("PS_BENCH_STAGE_HISTORY", "")
    hist = (hist + " | " if hist else "") + "stage %d left: %s" % (stage, why)
    sys.stderr.write("[bench rank %d] %s\n" % (rank, hist))
    sys.stderr.flush()
    if stage + 1 >= len(STAGES):
        if rank == 0:
            os.write(int(os.environ.get("PS_BENCH_STDOUT_FD", "1")),
                     (json.dumps({"metric": "Wide&Deep training examples/sec", "value": None, "unit": "examples/s", "error": hist}) + "\n").encode())
        os._exit(3)
    env = dict(os.environ, PS_BENCH_STAGE=str(stage + 1), PS_BENCH_STAGE_HISTORY=hist)
    os.execve(sys.executable, [sys.executable] + sys.argv, env)


def run_bench(args, cfg, synth_batch):
    """One rank per GPU.  Weak scaling: every rank trains its own batch of cfg['B']; `value`
    is the whole-job examples/s over the max-over-ranks time of exactly `steps` steps."""
    rank = int(os.environ.get("RANK", "0"))
    stage = int(os.environ.get("PS_BENCH_STAGE", "0"))
    wd = Watchdog(stage, float(os.environ.get("PS_BENCH_WATCHDOG_S", "240")), rank)     # (first import of torch + RCCL init on a cold box: minutes)
    try:
        return _run_bench(args, cfg, synth_batch, stage, wd)
    except N.PsError as e:
        if e.code == N.PS_E_STATE and stage + 1 < len(STAGES):       # a bounded device-side wait timed out, the exchange's counts never arrived, ...
            next_stage(stage, "PsError: %s" % e, rank)
        raise
    finally:
        wd.stop()


def _run_bench(args, cfg, synth_batch, stage, wd):
    import torch                     # first: this process must share ONE HIP runtime with libps_amd
    import torch.distributed as dist
    import ps_amd

    L = N.lib()
    for kv_ in os.environ.get("PS_TUNE", "").split(","):      # measurement knobs (bench.py applies them itself on the fused path)
        if "=" in kv_:
            L.ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
    if stage >= 2:
        for k in (b"dev_wait", b"end_wait", b"shard_overlap"):
            L.ps_tune_set(k, 0)
    # (PS_MAPPED_PEER=2: a one-GPU run moves its own part through the mapped path too -- timelines of the mode on a one-GPU box)
    L.ps_tune_set(b"mapped_peer", int(os.environ.get("PS_MAPPED_PEER", "1")) if (stage == 0 and not os.environ.get("PS_BENCH_NO_MAPPED")) else 0)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if stage:       # (the previous stage's process image may have left its rendezvous port bound)
        base = int(os.environ.get("PS_BENCH_BASE_PORT", os.environ["MASTER_PORT"]))
        os.environ["PS_BENCH_BASE_PORT"] = str(base)
        os.environ["MASTER_PORT"] = str(base + 13 * stage)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wd.kick("init_process_group")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = dict(cfg)
    cfg["idgen"] = getattr(args, "idgen", cfg.get("idgen", "zipf_truncated"))
    kv = ps_amd.KVStore(local, cfg["seed"])
    kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"], shard=rank, nshards=world)
    overlap = bool(getattr(args, "overlap", 1))
    native = bool(getattr(args, "native", 1)) and stage < 3
    threaded = False
    wire_check = None
    rccl = None
    if native:
        # the library drives the exchange (ps_shard_step: one C call per step, RCCL bound inside libps_amd.so);
        # torch.distributed only hands the RCCL ids round and keeps the bench's barriers
        gms = [ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
               for _ in range(2 if overlap else 1)]      # two plan contexts: step t+1 begins before step t finishes
        idt = torch.zeros(384, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(NativeWorker.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        wd.kick("ncclCommInitRank x3")
        force = int(getattr(args, "rccl_force", 0)) if world == 1 else 0
        try:
            if os.environ.get("PS_AMD_FORCE_TORCH_WIRE"):       # exercise the fallback
                raise RuntimeError("PS_AMD_FORCE_TORCH_WIRE is set")
            if force:            # one rank, and the wire anyway: every collective of the step through RCCL (ps_native.h)
                L.ps_tune_set(b"rccl_force", force)
            try:
                worker = NativeWorker(gms, world, rank, id256=bytes(idt.cpu().numpy().tobytes()), is_async=bool(getattr(args, "is_async", 0)))
            finally:             # (a process-global knob: a constructor that raised must not leave later 1-rank tables on the wire)
                L.ps_tune_set(b"rccl_force", 0)
        except Exception as e:      # noqa: BLE001 -- decided collectively below
            worker = None
            ok.zero_()
            print("rank %d: RCCL communicator inside libps_amd failed (%s); falling back to the torch.distributed wire" % (rank, e), flush=True)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank takes the same path
        if int(ok.item()) != 0:
            # the RCCL calls inside libps_amd (ncclSend/Recv groups, all-gather, all-reduce) have never run at N > 1 on
            # the development box (one GPU): verify the wire on known patterns before trusting a single step
            wd.kick("wire self-check")
            try:
                worker.selfcheck()
                wire_check = "ok"
            except Exception as e:      # noqa: BLE001 -- decided collectively below
                ok.zero_()
                wire_check = "FAILED: %s" % e
                print("rank %d: RCCL wire self-check failed (%s); falling back to the torch.distributed wire" % (rank, e), flush=True)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if worker is not None:
                worker.close()
            for g in gms:
                g.close()
            native = False
        else:
            # every rank must join its streams the same way (the communicators a step uses depend on it)
            why = C.create_string_buffer(256)
            jm = torch.tensor([L.ps_store_join_mode(kv.h, why, 256)], dtype=torch.int32, device=dev)
            dist.all_reduce(jm, op=dist.ReduceOp.MIN)
            if int(jm.item()) == 0:
                for k in (b"dev_wait", b"end_wait", b"shard_overlap"):
                    L.ps_tune_set(k, 0)
            if world > 1 or force:
                rccl = rccl_info(worker)
            worker_run = lambda n: worker.run(batches, n)                              # noqa: E731
    if not native:
        overlap = bool(getattr(args, "overlap_torch", 0))     # measured: no gain on this wire (host-bound), keep it simple
        gms = [ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
               for _ in range(4 if overlap else 1)]       # plan contexts: steps t+1, t+2 are planned while step t trains and t-1 drains
        torch.cuda.set_stream(torch.cuda.Stream(dev))      # not the legacy null stream (implicit syncs with blocking streams)
        N.check(L.ps_store_set_stream(kv.h, torch.cuda.current_stream().cuda_stream))
        comm = TorchComm(dist, torch, dev, overlap=overlap)
        threaded = overlap and bool(getattr(args, "prefetch_thread", 0))
        worker = ShardedWorker(HipBackend(gms, torch, dev), comm, is_async=bool(getattr(args, "is_async", 0)))
        worker_run = lambda n: worker.run(batches, n, threaded=threaded)               # noqa: E731
    rng = np.random.default_rng(cfg["seed"] + 1000 * rank)     # every worker reads its own slice of the data
    nb = 32        # enough distinct samples that the model cannot memorise them within the run (see bench.py)
    batches = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(nb)]
    # the FIRST steps under the watchdog's short leash: a wedged exchange shows here
    wd.kick("first 3 steps", float(os.environ.get("PS_BENCH_FIRST_STEPS_S", "60")))
    if os.environ.get("PS_BENCH_FAKE_HANG") == str(stage):      # (test hook: this stage never gets past its first steps)
        time.sleep(10 ** 6)
    worker_run(3)
    kv.sync(); torch.cuda.synchronize()
    dist.barrier()
    mapped = None
    if native:
        mp5 = (C.c_int64 * 5)()
        N.check(L.ps_shard_mapped_info(gms[0].h, mp5))
        mapped = {"active": mp5[0] == 1, "wire_check": "failed on some rank: every rank went back to RCCL's all-to-all-v" if mp5[0] < 0 else ("ok" if mp5[0] == 1 else "not run"),
                  "flag_words_fine_grained": bool(mp5[4])}
    # priming (untimed, on top of --warmup): the first few hundred steps run ~30% slower while the communicators' buffers
    # and the host's clocks settle; keep that out of the timed region
    prim = max(args.warmup, 1) + int(getattr(args, "priming", 300))
    wd.kick("priming", 60 + 0.02 * prim)
    tp = time.perf_counter()
    # (in two pieces with a wait in between: the first ~18 calls behind a wait that follows a LONG asynchronous run take the host
    #  180 us instead of 142 -- the HIP runtime recycling what 300 steps of launches left behind; behind a short run they do not.
    #  tools/shard_short_run.py, PS_BENCH_DEBUG=1: 64-step regions behind 305 and behind 5 steps)
    tail = min(32, prim // 2)
    worker_run(prim - tail)
    kv.sync(); torch.cuda.synchronize()
    worker_run(tail)
    kv.sync(); torch.cuda.synchronize()
    dist.barrier()
    dtp_local = time.perf_counter() - tp               # (reduced over the ranks AFTER the timed region: nothing but the barrier in front of it)
    wd.kick("timed region", 60 + 0.02 * args.steps)
    t0 = time.perf_counter()
    worker_run(args.steps)
    t_enq = time.perf_counter()
    kv.sync(); torch.cuda.synchronize()
    t_sync = time.perf_counter()
    dist.barrier()
    if os.environ.get("PS_BENCH_DEBUG") and rank == 0:
        print("[timed region] enqueue %.0f us, sync %.0f us, barrier %.0f us" % (1e6 * (t_enq - t0), 1e6 * (t_sync - t_enq), 1e6 * (time.perf_counter() - t_sync)), file=sys.stderr, flush=True)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    dtp = torch.tensor([dtp_local], dtype=torch.float64, device=dev)
    dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
    unprimed_ms = 1e3 * float(dtp.item()) / prim       # what a job pays while the communicators' buffers and the host's clocks settle
    wd.kick("after the timed region", 300)
    loss = worker.step(batches[0], want_loss=True)
    stats = None
    if native and rccl is not None:
        rccl = dict(rccl, **rccl_info(worker))          # (with the call counts of the whole run)
    if native:
        st = (C.c_int64 * 10)()
        N.check(L.ps_shard_exchange_stats(gms[0].h, st, 10))
        n_st = max(int(st[0]), 1)
        per = {"id_blocks_sent": st[1] / n_st, "rows_received": st[2] / n_st, "gradients_sent": st[3] / n_st,
               "allreduce_payload": st[4] / n_st}
        wire = per["id_blocks_sent"] + per["rows_received"] + per["gradients_sent"] + (2.0 * (world - 1) / world) * per["allreduce_payload"]
        stats = {"bytes_per_step_per_rank": {k: int(v) for k, v in per.items()},
                 "wire_bytes_per_step_per_rank": int(wire),
                 # the step's average: bytes this rank moves over xGMI per step / the step's duration (the links idle most of a step)
                 "avg_xgmi_GBs_per_rank": wire / (dt / args.steps) / 1e9,
                 "unique_keys_requested_per_step": st[5] / n_st, "keys_served_per_step": st[6] / n_st, "id_block_words": int(st[7]),
                 "full_id_block_words": int(st[9]), "steps_with_the_full_size_id_exchange": int(st[8])}
        ct = collective_times(worker, worker_run, kv, gms[0])
        if ct:
            stats["collective_device_us"] = ct
    if os.environ.get("PS_STAMPS") and rank == 0:
        # measurement: the sharded step as the GPU ran it (in-kernel time stamps, tools/gpu_timeline.py's mechanism)
        L.ps_tune_set(b"stamps", 1)
        worker_run(200)
        kv.sync()
        fn = L.ps_dbg_stamps
        fn.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
        names = C.create_string_buffer(1 << 18)
        vals = (C.c_ulonglong * (2 * 8192))()
        n = fn(names, len(names), vals, 8192)
        L.ps_tune_set(b"stamps", 0)
        with open(os.environ["PS_STAMPS"], "w") as f:
            json.dump({"names": names.value.decode().split("\n")[:n], "vals": list(vals[:2 * n])}, f)
    phases = {}
    if getattr(args, "phases", 0) and not native:
        for i in range(50):
            worker.step_timed(batches[i % nb], torch.cuda.synchronize, phases)
        phases = {k: round(1e6 * v / 50, 1) for k, v in phases.items()}
        for g in gms:
            g.set_profile(True)
        worker.run(batches, 20)
        torch.cuda.synchronize()
        groups = {}
        for g in gms:
            for k, v in g.profile_report().items():
                c0, m0 = groups.get(k, (0, 0.0))
                groups[k] = (c0 + v[0], m0 + v[1])
            g.set_profile(False)
        phases["kernel_groups_us"] = {k: round(1e3 * v[1] / max(v[0], 1), 2) for k, v in groups.items()}
    why = C.create_string_buffer(256)
    join_mode = L.ps_store_join_mode(kv.h, why, 256)
    timeouts = int(L.ps_store_wait_timeouts(kv.h))
    for b in batches:
        b.close()
    dist.barrier()
    if native:
        worker.close()
    for g in gms:
        g.close()
    kv.close()
    # the 1-GPU reference values of the same run (rank 0, while the others wait): the fused step bench.py --gpus 1 times,
    # and the sharded step with a 1-rank communicator -- what the N-GPU value has to be read against
    n1 = {}
    if rank == 0 and world > 1 and int(getattr(args, "n1_reference", 1)):
        wd.kick("1-GPU reference", 300)
        n1 = n1_reference(cfg, synth_batch, local, min(args.steps, 1000))
    elif world == 1 and native and int(getattr(args, "wire_cost", 1)):
        # one GPU: what the wire costs -- the same sharded step with device copies, with every collective through RCCL
        # (a rank's own keys off the wire), and with the wire running under own-keys-in-place (the N > 1 data path)
        wd.kick("wire cost", 300)
        n1 = {"wire_cost_ms_per_step": sharded_n1_modes(cfg, synth_batch, local, min(args.steps, 1000))}
    dist.barrier()
    out = None
    if rank == 0:
        out = {
            "metric": "Wide&Deep training examples/sec", "value": cfg["B"] * world * args.steps / dt, "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Wide&Deep synthetic (26 x 100k x 16, FC[512,256,1]), batch 4096 per GPU, "
                                   "embedding rows sharded id mod N (PSRouterClient routing -> RCCL all-to-all-v), dense + wide all-reduce, BSP",
                       "global_batch": cfg["B"] * world, "parallelism": "ps-shard%d" % world, "resident_inputs": True,
                       "id_generator": "%s(alpha=%g, V=%d)" % (cfg.get("idgen", "zipf_truncated"), cfg["zipf"], cfg["V"]),
                       "exchange_driver": "libps_amd (ps_shard_step, RCCL via dlopen)" if native else "torch.distributed",
                       "stage": stage, "stage_name": STAGES[stage], "stage_history": os.environ.get("PS_BENCH_STAGE_HISTORY", ""),
                       "world_size": world, "rccl": rccl, "wire_selfcheck": wire_check,
                       # rows and gradients: stores into the peers' mapped memory (stage 0) or RCCL's grouped send / recv
                       "rows_and_gradients": ("mapped peer memory (hipIpc, stores + flags)" if (mapped and mapped["active"]) else "RCCL all-to-all-v") if native else "torch.distributed",
                       "mapped_peer": mapped,
                       "stream_joins": "device-side flags" if join_mode == 1 else "events (%s)" % why.value.decode(),
                       "device_wait_timeouts": timeouts,
                       "prefetch_next_key_lists": overlap, "prefetch_thread": threaded,
                       "priming_steps_untimed": int(getattr(args, "priming", 300)) + 3},
            "final_loss": loss,
            # the first steps of a job (untimed above): priming_steps_untimed steps from cold communicators
            "unprimed_ms_per_step": unprimed_ms,
        }
        if stats:
            out["exchange"] = stats
        if n1:
            out.update(n1)
            if n1.get("n1_fused_ms"):
                out["scaling_vs_n1_fused"] = out["value"] / (cfg["B"] / (1e-3 * n1["n1_fused_ms"]))
        if phases:
            out["phase_us_serialised"] = phases
    dist.destroy_process_group()
    return out


def rccl_info(worker):
    """RCCL's own view of the table a NativeWorker made (evidence of the wire a run used) + the operations issued so far."""
    cc, ur, hs = C.c_int(), C.c_int(), C.c_int()
    L = N.lib()
    N.check(L.ps_comm_rccl_info(C.byref(worker.ops), C.byref(cc), C.byref(ur), C.byref(hs)))
    calls = (C.c_int64 * 5)()
    N.check(L.ps_comm_rccl_calls(C.byref(worker.ops), calls))
    return {"ncclCommCount": cc.value, "ncclCommUserRank": ur.value, "extra_communicators": hs.value,
            "calls": {"ncclAllGather": int(calls[0]), "send_recv_groups": int(calls[1]), "ncclAllReduce": int(calls[2]),
                      "ncclSend_plus_ncclRecv": int(calls[3])}, "on_the_wire": bool(calls[4])}


def collective_times(worker, worker_run, kv, gm, steps=200):
    """Device time of every collective of the step, by kind (HIP events around each call on the stream it is enqueued on, in a
    pass of its own after the timed region: the events cost their streams a few microseconds each)."""
    L = N.lib()
    if L.ps_tune_set(b"comm_timing", 1) != 0:
        return None
    try:
        worker_run(steps)
        kv.sync()
        out = (C.c_double * 8)()
        N.check(L.ps_shard_collective_times(gm.h, out))
    finally:
        L.ps_tune_set(b"comm_timing", 0)
    names = ["id_blocks", "rows", "gradients", "allreduce"]
    return {n: {"avg_us": round(1e3 * out[2 * i + 1] / max(out[2 * i], 1), 2), "calls": int(out[2 * i])} for i, n in enumerate(names)}


def sharded_n1_modes(cfg, synth_batch, device, steps, modes=(0, 1, 2), with_info=False, coll_times=None, regions=None):
    """ps_shard_step on ONE GPU with a 1-rank table: rccl_force 0 (device copies), 1 (everything through RCCL), 2 (RCCL
    running, own keys in place).  ms per step of `steps` steps after 300 priming steps, each mode on a fresh store.
    The priming runs in two pieces with a wait in between, like run_bench's (a short region right behind a LONG asynchronous
    run pays the HIP runtime's housekeeping: tools/shard_short_run.py).  coll_times (a dict): filled with the device time of
    every collective by kind and mode (a pass of its own after the timed region)."""
    import ps_amd
    L = N.lib()
    for kv_ in os.environ.get("PS_TUNE", "").split(","):      # measurement knobs for A/B runs of this leg
        if "=" in kv_:
            L.ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
    res, info = {}, None
    names = {0: "device_copies", 1: "rccl_everything_off_the_wire", 2: "rccl_with_own_keys_in_place", 3: "mapped_peer"}
    for force in modes:
        # mode 3 (round 6): rows and gradients through the mapped-peer launch (this rank's own part stored by the same kernel that would
        # store the peers', own flag raised and awaited) with the id blocks and the all-reduce through RCCL as in mode 2
        mapped = force == 3
        if mapped:
            force = 2
            L.ps_tune_set(b"mapped_peer", 2)
        rng = np.random.default_rng(cfg["seed"])
        kv = ps_amd.KVStore(device, cfg["seed"])
        kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
        gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
        bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(32)]
        L.ps_tune_set(b"rccl_force", force)
        try:
            wk = NativeWorker([gm], 1, 0)
            if force:
                wk.selfcheck()
            wk.run(bs, 4)               # (the model's first begin decides whether its exchanges are mapped: while the knob is set)
        finally:
            L.ps_tune_set(b"rccl_force", 0)
            L.ps_tune_set(b"mapped_peer", 0)
        if mapped:
            mp5 = (C.c_int64 * 5)()
            N.check(L.ps_shard_mapped_info(gm.h, mp5))
            if not mp5[0]:
                raise RuntimeError("the mapped-peer mode did not come up on a 1-rank table")
        wk.run(bs, 264)
        kv.sync()
        wk.run(bs, 32)
        kv.sync()
        # a short region (the driver's 20 steps = 3 ms) carries whatever hiccup the host has in it (profiles/r05_shard_20step_repeats.txt:
        # 0.151 / 0.183 / 0.170 on one box): five regions of `steps` steps, the MEDIAN reported, all five in `regions_ms`
        reps = 5 if steps < 200 else 1
        regs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            wk.run(bs, steps)
            kv.sync()
            regs.append(1e3 * (time.perf_counter() - t0) / steps)
        name = names[3] if mapped else names[force]
        res[name] = sorted(regs)[len(regs) // 2]
        if regions is not None:
            regions[name] = [round(x, 5) for x in regs]
        if force and with_info:
            info = rccl_info(wk)
        if coll_times is not None:
            ct = collective_times(wk, lambda n: wk.run(bs, n), kv, gm, steps=100)
            if ct:
                coll_times[name] = {k: v["avg_us"] for k, v in ct.items()}
        wk.close()
        for b in bs:
            b.close()
        gm.close(); kv.close()
    return (res, info) if with_info else res


def n1_reference(cfg, synth_batch, device, steps):
    """The fused single-GPU step and the sharded step on a 1-rank communicator, same config, same generator, on this GPU."""
    import ps_amd
    res = {}
    rng = np.random.default_rng(cfg["seed"])
    kv = ps_amd.KVStore(device, cfg["seed"])
    kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(32)]
    for i in range(100):
        gm.train_async(bs[i % 32])
    gm.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        gm.train_async(bs[i % 32])
    gm.sync()
    res["n1_fused_ms"] = 1e3 * (time.perf_counter() - t0) / steps
    wk = NativeWorker([gm], 1, 0)
    wk.run(bs, 300)
    kv.sync()
    t0 = time.perf_counter()
    wk.run(bs, steps)
    kv.sync()
    res["n1_sharded_ms"] = 1e3 * (time.perf_counter() - t0) / steps
    wk.close()
    for b in bs:
        b.close()
    gm.close(); kv.close()
    return res
