"""Measurement only: is the sharded step's early plan (the next step's plan on side chain 0 while the step trains) taken for a
test-sized model?  In-kernel stamps of one pipelined step at N = 1.  python tools/early_check.py"""
import sys, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from ps_amd.sharded import NativeWorker
L = N.lib()
F, D, X, fc, V, B, WS = 3, 4, 2, [6, 4, 1], 23, 10, 11
rng = np.random.default_rng(1)
kv = ps_amd.KVStore(0, 1)
kv.create_embedding([V] * F, D)
gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
bs = []
for _ in range(4):
    E = rng.integers(0, V, (B, F)).astype(np.int64)
    bs.append(ps_amd.DeviceBatch(kv, E, rng.standard_normal((B, X)).astype(np.float32), (rng.random(B) < 0.3).astype(np.float32), E % WS))
wk = NativeWorker([gm], 1, 0)
wk.run(bs, 8)
kv.sync()
L.ps_tune_set(b"stamps", 1)
wk.run(bs, 6)
kv.sync()
fn = L.ps_dbg_stamps
fn.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
names = C.create_string_buffer(1 << 16); vals = (C.c_ulonglong * 4096)()
n = fn(names, len(names), vals, 2048)
nm = names.value.decode().split("\n")[:n]
v = np.array(vals[:2 * n], np.int64).reshape(n, 2) / 100.0
i0 = [i for i, x in enumerate(nm) if x == "emb_fwd"][2]
early = 0
for i in range(i0, min(i0 + 21, n)):
    print("%-16s %9.1f -> %9.1f" % (nm[i], v[i, 0] - v[i0, 0], v[i, 1] - v[i0, 0]))
ebw = [i for i in range(i0, n) if nm[i] == "emb_bwd_update"][0]
sk = [i for i in range(ebw, n) if nm[i] == "shard_keys"][0]
print("next plan's key kernel starts %.1f us %s this step's embedding backward ends -> early plan %s" % (
    abs(v[sk, 0] - v[ebw, 1]), "BEFORE" if v[sk, 0] < v[ebw, 1] else "after", "ON" if v[sk, 0] < v[ebw, 1] else "OFF"))
