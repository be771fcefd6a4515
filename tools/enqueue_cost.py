"""Host-side enqueue cost of one fused training step: eager multi-stream vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2)
for graph in (0, 1):
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"], use_graph=graph)
    rng = np.random.default_rng(1)
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(4)]
    for i in range(20):
        gm.train_async(bs[i % 4])
    gm.sync()
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        gm.train_async(bs[i % 4])
    t1 = time.perf_counter()
    gm.sync()
    t2 = time.perf_counter()
    print("graph=%d: host enqueue %.1f us/step, total %.1f us/step" % (graph, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
    for b in bs: b.close()
    gm.close(); kv.close()
