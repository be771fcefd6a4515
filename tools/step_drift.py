"""Does the fused step drift over a long run?  ms/step per block of 500 steps over 12000 steps.  python tools/step_drift.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(64)]
K, blk = 12000, 500
out = []
t0 = time.perf_counter()
for i in range(K):
    gm.train_async(bs[i % 64])
    if (i + 1) % blk == 0:
        gm.sync(); t1 = time.perf_counter(); out.append(1e3 * (t1 - t0) / blk); t0 = t1
print("ms/step per block of %d: %s" % (blk, " ".join("%.4f" % x for x in out)))
