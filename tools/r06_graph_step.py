"""The fused step replayed as a hipGraph (ps_model_config_t.use_graph: one graph per batch's pointers; joins by events, no device-side flags)
against the eager three-stream step, 32 resident batches, 2000 steps each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2)
for graph in (0, 1, 0, 1):
    rng = np.random.default_rng(cfg["seed"])
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"], use_graph=graph)
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(32)]
    for i in range(400): gm.train_async(bs[i % 32])
    gm.sync(); t0 = time.perf_counter()
    for i in range(2000): gm.train_async(bs[i % 32])
    th = time.perf_counter() - t0
    gm.sync(); dt = time.perf_counter() - t0
    print("use_graph %d: %.4f ms per step (host enqueue %.1f us per step)" % (graph, 1e3 * dt / 2000, 1e6 * th / 2000), flush=True)
    for b in bs: b.close()
    gm.close(); kv.close()
