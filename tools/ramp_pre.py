"""Does GPU work of ANOTHER kind right in front of a short timed region leave the clocks up?  For each PRE in (none, gather, steps):
[pre-phase] -> sync -> [5 steps, sync, 20 steps timed] x 5, printing ms/step of the 20.   python tools/ramp_pre.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
L = N.lib()
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(64)]
kv2 = ps_amd.KVStore(0, 7)
def gather(nl):
    ms, br, bw = C.c_double(), C.c_double(), C.c_double()
    N.check(L.ps_bench_gather(kv2.h, 64 * 1000 * 1000, 64, 1 << 22, 1, nl, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw)))
    return ms.value
gather(5)
def region():
    for i in range(5): gm.train_async(bs[i])
    gm.sync()
    t0 = time.perf_counter()
    for i in range(20): gm.train_async(bs[(5 + i) % 64])
    gm.sync()
    return 1e3 * (time.perf_counter() - t0) / 20
for pre in ("none", "gather", "steps", "none", "gather", "steps"):
    out = []
    for r in range(5):
        time.sleep(0.3)           # (idle: clocks down)
        t0 = time.perf_counter()
        if pre == "gather": gather(60)
        elif pre == "steps":
            for i in range(300): gm.train_async(bs[i % 64])
            gm.sync()
        tp = time.perf_counter() - t0
        out.append(region())
    print("%-7s (pre-phase %.1f ms): %s" % (pre, 1e3 * tp, " ".join("%.4f" % x for x in out)))
