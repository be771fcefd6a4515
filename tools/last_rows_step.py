"""Fused step time vs the split-K target of the dW GEMMs (the split is chosen when the model is built)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
for target in [int(a) for a in sys.argv[1:]] or (32, 16, 64):
    N.lib().ps_tune_set(b"last_rows", target)
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    rng = np.random.default_rng(1)
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(16)]
    for i in range(50): gm.train_async(bs[i % 16])
    gm.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(400): gm.train_async(bs[i % 16])
        gm.sync()
        best = min(best, (time.perf_counter() - t0) / 400)
    gm.set_profile(True)
    for i in range(40): gm.train_async(bs[i % 16])
    gm.sync(); rep = gm.profile_report(); gm.set_profile(False)
    print("last_rows %4d: %.4f ms/step, head_last_bwd %.1f us, dense_update %.1f us" % (target, 1e3 * best, 1e3 * rep["head_last_bwd"][1] / rep["head_last_bwd"][0], 1e3 * rep["dense_update"][1] / rep["dense_update"][0]))
    for b in bs: b.close()
    gm.close(); kv.close()
N.lib().ps_tune_set(b"last_rows", 0)
