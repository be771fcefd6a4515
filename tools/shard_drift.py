"""Does the sharded step drift over a long run?  ms/step per block of 250 steps over 5000 steps (N = 1, device copies).  python tools/shard_drift.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd.sharded import NativeWorker
from bench import C2, synth_batch
cfg = dict(C2)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(32)]
wk = NativeWorker([gm], 1, 0)
K, blk = 5000, 250
wk.begin(0, bs[0], side=False)
out = []
t0 = time.perf_counter()
for i in range(K):
    wk.finish_begin(0, bs[(i + 1) % 32] if i + 1 < K else None, False)
    if (i + 1) % blk == 0:
        t1 = time.perf_counter(); out.append(1e3 * (t1 - t0) / blk); t0 = t1
kv.sync()
print("ms/step per block of %d: %s" % (blk, " ".join("%.4f" % x for x in out)))
