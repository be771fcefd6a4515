"""Measurement only: what would two independent half-batch chains on one GPU buy?  N processes each train their own model at
batch 4096 / N on the same device at the same time (their kernel boundaries interleave); aggregate examples / s against one
process at batch 4096.     python tools/two_chains_probe.py [procs=2] [steps=3000]"""
import multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def work(rank, n, steps, bar, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import ps_amd
    from bench import C2, synth_batch
    cfg = dict(C2); cfg["B"] = 4096 // n
    rng = np.random.default_rng(cfg["seed"] + rank)
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(64)]
    for i in range(200): gm.train_async(bs[i % 64])
    gm.sync()
    bar.wait()
    t0 = time.perf_counter()
    for i in range(steps): gm.train_async(bs[i % 64])
    gm.sync()
    dt = time.perf_counter() - t0
    bar.wait()
    q.put((rank, dt / steps * 1e3))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    for procs in (1, n):
        ctx = mp.get_context("spawn")
        bar, q = ctx.Barrier(procs), ctx.Queue()
        ps = [ctx.Process(target=work, args=(r, procs, steps, bar, q)) for r in range(procs)]
        for p in ps: p.start()
        res = [q.get(timeout=300) for _ in ps]
        for p in ps: p.join()
        ms = max(r[1] for r in res)
        print("%d process(es) x batch %d: %.4f ms per step each -> %.2f M examples/s in all" % (procs, 4096 // procs, ms, 4096 / ms / 1e3))
