"""configs[1] step when the boundary is handed HOST buffers (numpy, pageable): the PCIe-inclusive rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2); rng = np.random.default_rng(1)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
hb = [ps_amd.Batch(*synth_batch(cfg, rng)) for _ in range(8)]
db = [ps_amd.DeviceBatch(kv, b.E, b.X, b.Y, b.W) for b in hb]
for name, bs in (("resident", db), ("host (pageable numpy)", hb)):
    for i in range(30): gm.train_async(bs[i % 8])
    gm.sync(); n = 300; t0 = time.perf_counter()
    for i in range(n): gm.train_async(bs[i % 8])
    gm.sync(); dt = (time.perf_counter() - t0) / n
    print("%-24s %.4f ms/step  %.2f M examples/s" % (name, 1e3 * dt, cfg["B"] / dt / 1e6))
