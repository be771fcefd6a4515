"""What the sort chain on the side stream costs the step (ONE batch repeated; with sort_ablate the sorted arrays of
the first steps are reused, which is valid for a repeated batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
for abl in (0, 1, 0, 1):
    N.lib().ps_tune_set(b"sort_ablate", abl)
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    rng = np.random.default_rng(1)
    b = ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng))
    for i in range(50): gm.train_async(b)
    gm.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(500): gm.train_async(b)
        gm.sync()
        best = min(best, (time.perf_counter() - t0) / 500)
    print("sort_ablate %d: %.4f ms/step" % (abl, 1e3 * best))
    b.close(); gm.close(); kv.close()
N.lib().ps_tune_set(b"sort_ablate", 0)
