"""Rehearsal of the 8-GPU run on ONE GPU: N rank PROCESSES at the full configs[2] size (26 x 100k x 16, FC[512,256,1], batch 4096
per rank, truncated Zipf ids), ps_shard_step over gloo collectives with host staging (tests/test_gpu_multiproc.py's table).
Everything a real node runs except the RCCL calls: block sizes, exchange buffers, 8 peers in the push, overlap mode, the pipeline.
Checks: no error / timeout on any rank, the replicated tensors (FC weights, wide table) bit-identical across ranks after K steps,
exchange statistics.     python tools/rehearse_n8.py [ranks=8] [steps=12]"""
import multiprocessing as mp
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def rank_process(rank, world, port, steps, q):
    try:
        import ctypes as C, hashlib
        import numpy as np
        import torch, torch.distributed as dist
        import ps_amd
        from ps_amd import native as N
        from ps_amd.sharded import NativeWorker
        from bench import C2, synth_batch
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        cfg = dict(C2)
        kv = ps_amd.KVStore(0, cfg["seed"])
        kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"], shard=rank, nshards=world)
        gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
        L = N.lib()
        f32 = np.float32

        class GlooOps:
            def __init__(self):
                self.ops = N.ps_comm_ops_t(); self.ops.ctx, self.ops.nranks, self.ops.rank = None, world, rank
                self._k = (N.ALL_GATHER_FN(self.ag), N.ALL_TO_ALL_V_FN(self.a2a), N.ALL_REDUCE_FN(self.ar))
                self.ops.all_gather, self.ops.all_to_all_v, self.ops.all_reduce_sum_f32 = self._k
                self.err = None
            def _down(self, p, n):
                a = np.empty(n, np.uint8)
                if n: N.check(L.ps_dev_download(kv.h, a.ctypes.data, p, n))
                return a
            def _up(self, p, a):
                if a.size: a = np.ascontiguousarray(a); N.check(L.ps_dev_upload(kv.h, p, a.ctypes.data, a.nbytes))
            def _g(self, fn, stream):
                try:
                    N.check(L.ps_stream_sync(kv.h, stream)); fn(); return 0
                except BaseException as e:      # noqa: BLE001
                    self.err = e; return 500
            def ag(self, ctx, send, recv, nb, stream):
                def f():
                    m = torch.from_numpy(self._down(send, nb)); parts = [torch.empty_like(m) for _ in range(world)]
                    dist.all_gather(parts, m); self._up(recv, torch.cat(parts).numpy())
                return self._g(f, stream)
            def a2a(self, ctx, send, sc, recv, rc, eb, stream):
                def f():
                    scl = [int(sc[i]) * eb for i in range(world)]; rcl = [int(rc[i]) * eb for i in range(world)]
                    out = torch.empty(sum(rcl), dtype=torch.uint8)
                    dist.all_to_all_single(out, torch.from_numpy(self._down(send, sum(scl))), output_split_sizes=rcl, input_split_sizes=scl)
                    self._up(recv, out.numpy())
                return self._g(f, stream)
            def ar(self, ctx, buf, n, stream):
                def f():
                    m = torch.from_numpy(self._down(buf, n * 4).view(f32)); parts = [torch.empty_like(m) for _ in range(world)]
                    dist.all_gather(parts, m)
                    tot = parts[0].numpy().copy()
                    for p in parts[1:]: tot = (tot + p.numpy()).astype(f32)
                    self._up(buf, tot)
                return self._g(f, stream)

        comm = GlooOps()
        wk = NativeWorker([gm], world, rank, ops=comm.ops)
        wk.selfcheck()
        rng = np.random.default_rng(cfg["seed"] + 1000 * rank)
        bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(4)]
        t0 = time.time()
        wk.run(bs, steps)
        kv.sync()
        if comm.err is not None: raise comm.err
        loss = wk.step(bs[0], want_loss=True)
        st = (C.c_int64 * 8)(); N.check(L.ps_shard_exchange_stats(gm.h, st, 8))
        why = C.create_string_buffer(256)
        mode = L.ps_store_join_mode(kv.h, why, 256)
        h = hashlib.sha256()
        for l in range(3): h.update(kv.get("fc%d.weights" % l).tobytes()); h.update(kv.get("fc%d.bias" % l).tobytes())
        h.update(kv.get_wide(np.arange(cfg["wide"])).tobytes())
        dist.barrier()
        q.put((rank, "ok", dict(loss=float(loss), digest=h.hexdigest(), stats=[int(x) for x in st], join_mode=mode, why=why.value.decode(),
                                timeouts=int(L.ps_store_wait_timeouts(kv.h)), seconds=time.time() - t0)))
        gm.close(); kv.close(); dist.destroy_process_group()
    except BaseException:       # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=rank_process, args=(r, world, port, steps, q), daemon=True) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(30)
        if p.is_alive(): p.kill()
    bad = [r for r in res if r[1] != "ok"]
    for r in bad: print("rank %d FAILED:\n%s" % (r[0], r[2]))
    if bad: sys.exit(1)
    res.sort()
    d0 = res[0][2]["digest"]
    same = all(r[2]["digest"] == d0 for r in res)
    for r in res:
        st = r[2]["stats"]; n = max(st[0], 1)
        print("rank %d: loss %.5f  joins %s%s  timeouts %d  per step: %d keys requested, %d served, id blocks %d B, rows %d B, gradients %d B, all-reduce %d B  (%.1f s)" % (
            r[0], r[2]["loss"], "flags" if r[2]["join_mode"] == 1 else "events", "" if r[2]["join_mode"] == 1 else " (" + r[2]["why"] + ")", r[2]["timeouts"],
            st[5] // n, st[6] // n, st[1] // n, st[2] // n, st[3] // n, st[4] // n, r[2]["seconds"]))
    print("%d ranks x %d steps at configs[2] size: replicated tensors %s across ranks" % (world, steps, "bit-identical" if same else "DIFFER"))
    sys.exit(0 if same else 1)
