"""What a SHORT timed region costs (the driver's round-end line is `bench.py --steps 20 --warmup 5`): dt(K) for K = 10..320 steps of
configs[2]'s fused step, bracketed exactly like bench.py's timed region, fitted as dt = a + b K.  a = what the region pays once (first
launch from an idle queue, the closing wait), b = the step.      python tools/short_run_cost.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2)
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 7
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(64)]
for i in range(5): gm.train_async(bs[i])
gm.sync()
Ks = [10, 20, 40, 80, 160, 320]
res = {K: [] for K in Ks}
for r in range(rep):
    for K in Ks:
        time.sleep(0.002 * (r % 3))          # (idle gaps of different lengths in front of the region)
        t0 = time.perf_counter()
        for i in range(K): gm.train_async(bs[i % 64])
        gm.sync()
        res[K].append(time.perf_counter() - t0)
med = np.array([np.median(res[K]) for K in Ks]); mn = np.array([np.min(res[K]) for K in Ks])
b, a = np.polyfit(Ks, med, 1)
for K, m_, n_ in zip(Ks, med, mn):
    print("K = %4d: median %.1f us (%.4f ms/step), min %.1f us" % (K, 1e6 * m_, 1e3 * m_ / K, 1e6 * n_))
print("fit: %.1f us once + %.2f us per step" % (1e6 * a, 1e6 * b))
# the host alone: how long does it take to enqueue K steps (no wait)?
t0 = time.perf_counter()
for i in range(20): gm.train_async(bs[i % 64])
t1 = time.perf_counter(); gm.sync(); t2 = time.perf_counter()
print("20 steps: host enqueue %.1f us, then the closing wait %.1f us" % (1e6 * (t1 - t0), 1e6 * (t2 - t1)))
