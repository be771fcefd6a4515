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
import os
import time

import numpy as np

from . import native as N


# ---------------------------------------------------------------------------
# communication
# ---------------------------------------------------------------------------
class LocalComm:
    """world_size 1: every collective is the identity (used by single-GPU tests)."""
    rank, world = 0, 1

    def exchange_counts(self, counts):
        return list(counts)

    def all_to_all_v(self, send, send_counts, recv_counts, width, dtype):
        return send

    def all_reduce_sum(self, buf):
        return buf

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed (nccl = RCCL on ROCm, gloo on CPU).  Buffers are torch tensors."""

    def __init__(self, dist, torch, device):
        self.dist, self.torch, self.device = dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange_counts(self, counts):
        t = self.torch
        send = t.tensor(counts, dtype=t.int64, device=self.device)
        recv = t.empty_like(send)
        self.dist.all_to_all_single(recv, send)
        return [int(x) for x in recv.tolist()]

    def all_to_all_v(self, send, send_counts, recv_counts, width, dtype):
        """send: [sum(send_counts), width] (or 1-D when width == 1), grouped by destination rank."""
        t = self.torch
        n_recv = int(sum(recv_counts))
        shape = (n_recv,) if send.dim() == 1 else (n_recv, width)
        recv = t.empty(shape, dtype=send.dtype, device=self.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
        return recv

    def all_reduce_sum(self, buf):
        self.dist.all_reduce(buf)     # sum
        return buf

    def barrier(self):
        self.dist.barrier()


# ---------------------------------------------------------------------------
# the orchestration: one worker/owner step
# ---------------------------------------------------------------------------
class ShardedWorker:
    def __init__(self, backend, comm, is_async=False):
        self.be, self.comm, self.is_async = backend, comm, is_async

    def step_timed(self, batch, sync, acc):
        """The same step with a host-side stopwatch around every phase (sync() between phases):
        measurement only -- it serialises what the real step overlaps."""
        be, comm = self.be, self.comm
        t = [time.perf_counter()]

        def lap(name):
            sync()
            t.append(time.perf_counter())
            acc[name] = acc.get(name, 0.0) + (t[-1] - t[-2])

        counts, send_rows = be.plan(batch, comm.world); lap("plan")
        rcounts = comm.exchange_counts(counts); lap("a2a_counts")
        recv_rows = comm.all_to_all_v(send_rows, counts, rcounts, 1, "u32"); lap("a2a_ids")
        rows_out = be.serve_pull(recv_rows, int(sum(rcounts))); lap("serve_pull")
        cache = comm.all_to_all_v(rows_out, rcounts, counts, be.D, "f32"); lap("a2a_rows")
        be.forward_backward(cache, False); lap("forward_backward")
        grads = be.grads()
        recv_grads = comm.all_to_all_v(grads, counts, rcounts, be.D, "f32"); lap("a2a_grads")
        be.apply_push(recv_rows, recv_grads, int(sum(rcounts)), self.is_async); lap("apply_push")
        flat = be.flat_grad()
        comm.all_reduce_sum(flat); lap("allreduce_flat")
        be.apply_flat(comm.world); lap("apply_flat")

    def step(self, batch, want_loss=True):
        be, comm = self.be, self.comm
        counts, send_rows = be.plan(batch, comm.world)                     # PSRouterClient.getList fan-out
        rcounts = comm.exchange_counts(counts)
        recv_rows = comm.all_to_all_v(send_rows, counts, rcounts, 1, "u32")
        rows_out = be.serve_pull(recv_rows, int(sum(rcounts)))             # PServer.getList
        cache = comm.all_to_all_v(rows_out, rcounts, counts, be.D, "f32")  # worker cache
        loss = be.forward_backward(cache, want_loss)                       # Model.train on the cached rows
        grads = be.grads()                                                 # what PSClient.push sends
        recv_grads = comm.all_to_all_v(grads, counts, rcounts, be.D, "f32")
        be.apply_push(recv_rows, recv_grads, int(sum(rcounts)), self.is_async)   # PServer.push + psUpdate
        flat = be.flat_grad()
        comm.all_reduce_sum(flat)                                          # dense tensors + wide keys
        be.apply_flat(comm.world)
        return loss


# ---------------------------------------------------------------------------
# product backend: the HIP library
# ---------------------------------------------------------------------------
class _DevView:
    """Zero-copy view of a device buffer owned by libps_amd for torch.as_tensor."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipBackend:
    """Device-side halves through the C ABI.  With torch: buffers cross as torch tensors
    (zero-copy views of the library's device memory); without (LocalComm): raw pointers."""

    def __init__(self, model, torch=None, device=None):
        self.m, self.kv = model, model.store
        self.D = model.D
        self.torch, self.device = torch, device
        self._keep = []

    def _tensor(self, ptr, shape, typestr, dtype):
        if self.torch is None:
            return (ptr, shape)
        if int(np.prod(shape)) == 0:
            return self.torch.empty(shape, dtype=dtype, device=self.device)
        return self.torch.as_tensor(_DevView(ptr, shape, typestr), device=self.device)

    @staticmethod
    def _ptr(buf):
        return buf[0] if isinstance(buf, tuple) else buf.data_ptr()

    def plan(self, batch, world):
        counts = (C.c_int64 * world)()
        rows = C.c_void_p()
        nu = C.c_int64()
        N.check(N.lib().ps_shard_plan(self.m.h, C.byref(batch.c), world, counts, C.byref(rows), C.byref(nu)))
        self.U = nu.value
        t = self.torch
        return list(counts), self._tensor(rows.value, (self.U,), "<i4", None if t is None else t.int32)

    def serve_pull(self, recv_rows, n):
        t = self.torch
        if t is None:
            out = C.c_void_p()
            N.check(N.lib().ps_dev_alloc(self.kv.h, max(n, 1) * self.D * 4, C.byref(out)))
            self._keep.append(out)
            buf = (out.value, (n, self.D))
        else:
            buf = t.empty((n, self.D), dtype=t.float32, device=self.device)
        N.check(N.lib().ps_shard_serve_pull(self.kv.h, self._ptr(recv_rows), n, self._ptr(buf)))
        return buf

    def forward_backward(self, cache, want_loss=True):
        loss = C.c_float()
        N.check(N.lib().ps_shard_forward_backward(self.m.h, self._ptr(cache), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def grads(self):
        g = C.c_void_p()
        nu = C.c_int64()
        N.check(N.lib().ps_shard_grads(self.m.h, C.byref(g), C.byref(nu)))
        t = self.torch
        return self._tensor(g.value, (nu.value, self.D), "<f4", None if t is None else t.float32)

    def apply_push(self, recv_rows, recv_grads, n, is_async):
        N.check(N.lib().ps_shard_apply_push(self.kv.h, self._ptr(recv_rows), self._ptr(recv_grads), n, int(is_async)))
        for p in self._keep:
            N.lib().ps_dev_free(self.kv.h, p)
        self._keep = []

    def flat_grad(self):
        f = C.c_void_p()
        n = C.c_int64()
        N.check(N.lib().ps_shard_flat_grad(self.m.h, C.byref(f), C.byref(n)))
        t = self.torch
        return self._tensor(f.value, (n.value,), "<f4", None if t is None else t.float32)

    def apply_flat(self, world):
        N.check(N.lib().ps_shard_apply_flat(self.m.h, world))


# ---------------------------------------------------------------------------
# bench.py --gpus N (N > 1): BASELINE configs[2]
# ---------------------------------------------------------------------------
def run_bench(args, cfg, synth_batch):
    """One rank per GPU.  Weak scaling: every rank trains its own batch of cfg['B']; `value`
    is the whole-job examples/s over the max-over-ranks time of exactly `steps` steps."""
    import torch                     # first: this process must share ONE HIP runtime with libps_amd
    import torch.distributed as dist
    import ps_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = dict(cfg)
    kv = ps_amd.KVStore(local, cfg["seed"])
    kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"], shard=rank, nshards=world)
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    N.check(N.lib().ps_store_set_stream(kv.h, torch.cuda.current_stream().cuda_stream))
    comm = TorchComm(dist, torch, dev)
    worker = ShardedWorker(HipBackend(gm, torch, dev), comm, is_async=bool(getattr(args, "is_async", 0)))
    rng = np.random.default_rng(cfg["seed"] + 1000 * rank)     # every worker reads its own slice of the data
    nb = 8
    batches = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(nb)]
    for i in range(max(args.warmup, 1)):
        worker.step(batches[i % nb], want_loss=False)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        worker.step(batches[i % nb], want_loss=False)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    loss = worker.step(batches[0], want_loss=True)
    phases = {}
    if getattr(args, "phases", 0):
        for i in range(50):
            worker.step_timed(batches[i % nb], torch.cuda.synchronize, phases)
        phases = {k: round(1e6 * v / 50, 1) for k, v in phases.items()}
    out = None
    if rank == 0:
        out = {
            "metric": "Wide&Deep training examples/sec", "value": cfg["B"] * world * args.steps / dt, "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Wide&Deep synthetic (26 x 100k x 16, FC[512,256,1]), batch 4096 per GPU, "
                                   "embedding rows sharded id mod N (PSRouterClient routing -> RCCL all-to-all-v), dense + wide all-reduce, BSP",
                       "global_batch": cfg["B"] * world, "parallelism": "ps-shard%d" % world, "resident_inputs": True},
            "final_loss": loss,
        }
        if phases:
            out["phase_us_serialised"] = phases
    for b in batches:
        b.close()
    dist.barrier()
    gm.close(); kv.close()
    dist.destroy_process_group()
    return out
