"""Where the long-key fold of k_emb_reduce_update spends its time (measurement only: ablated runs compute garbage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
for abl in (0, 1, 2, 4, 6):
    N.lib().ps_tune_set(b"seq_ablate", abl)
    kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    rng = np.random.default_rng(1)
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
    for i in range(20): gm.train_async(bs[i % 8])
    gm.sync(); gm.set_profile(True)
    for i in range(40): gm.train_async(bs[i % 8])
    gm.sync(); rep = gm.profile_report(); gm.set_profile(False)
    c, ms = rep["emb_bwd_update"]
    print("seq_ablate=%d: emb_bwd_update %.1f us" % (abl, 1e3 * ms / c))
    for b in bs: b.close()
    gm.close(); kv.close()
N.lib().ps_tune_set(b"seq_ablate", 0)
