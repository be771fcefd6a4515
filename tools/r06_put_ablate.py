"""What the mapped-peer put launch spends its time on (one GPU, 1-rank table, mapped_peer = 2): the device time of the rows /
gradient exchanges (HIP events around the launch, ps_tune_set("comm_timing")) with parts of the kernel switched off
(ps_tune_set("mapped_ablate"): 1 no stores, 2 plain stores, 4 no flags / no wait, 8 no source loads; results are wrong)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from ps_amd.sharded import NativeWorker, collective_times
from bench import C2, synth_batch
L = N.lib()
cfg = dict(C2)
rng = np.random.default_rng(cfg["seed"])
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(32)]
L.ps_tune_set(b"mapped_peer", 2)
wk = NativeWorker([gm], 1, 0)
wk.run(bs, 300); kv.sync()
for ab in (0, 4, 2, 6, 1, 5, 9, 13, 0):
    L.ps_tune_set(b"mapped_ablate", ab)
    wk.run(bs, 100); kv.sync()
    t0 = time.perf_counter(); wk.run(bs, 500); kv.sync(); ms = 1e3 * (time.perf_counter() - t0) / 500
    ct = collective_times(wk, lambda n: wk.run(bs, n), kv, gm, steps=200)
    print("ablate %2d: step %.4f ms   rows %.2f us  gradients %.2f us" % (ab, ms, ct["rows"]["avg_us"], ct["gradients"]["avg_us"]), flush=True)
L.ps_tune_set(b"mapped_ablate", 0)
