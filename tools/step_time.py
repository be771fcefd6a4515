"""Fused-step time with NB rotating batches: python tools/step_time.py [NB ...]   (PS_AMD_LIB picks the build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
from ps_amd import native as N
if os.environ.get("SORT_ABLATE"): N.lib().ps_tune_set(b"sort_ablate", 1)
for kv_ in os.environ.get("PS_TUNE", "").split(","):
    if "=" in kv_: N.lib().ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
cfg = dict(C2)
for nb in [int(a) for a in sys.argv[1:]] or [1, 16]:
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    rng = np.random.default_rng(1)
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(nb)]
    for i in range(50): gm.train_async(bs[i % nb])
    gm.sync()
    best, host = 1e9, 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(500): gm.train_async(bs[i % nb])
        t1 = time.perf_counter()
        gm.sync()
        best = min(best, (time.perf_counter() - t0) / 500)
        host = min(host, (t1 - t0) / 500)
    print("%d rotating batches: %.4f ms/step (host enqueue alone %.4f)" % (nb, 1e3 * best, 1e3 * host))
    for b in bs: b.close()
    gm.close(); kv.close()
