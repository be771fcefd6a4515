"""How long until the step reaches its steady duration after an idle queue?  600 stamped steps behind a sync: mean span and mean duration
of the first forward GEMM per block of 20 steps.    python tools/ramp_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
L = N.lib()
fn = L.ps_dbg_stamps
fn.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(64)]
for i in range(int(os.environ.get("WARM", "5"))): gm.train_async(bs[i % 64])
gm.sync()
L.ps_tune_set(b"stamps", 1)
K = 600
for i in range(K): gm.train_async(bs[i % 64])
gm.sync()
cap = 8192
names = C.create_string_buffer(1 << 18)
vals = (C.c_ulonglong * (2 * cap))()
n = fn(names, len(names), vals, cap)
nm = names.value.decode().split("\n")[:n]
v = np.array(vals[:2 * n], np.int64).reshape(n, 2) / 100.0
st = [i for i, x in enumerate(nm) if x == "emb_fwd"]
spans = np.diff([v[i, 0] for i in st])
g0 = np.array([v[i + 1, 1] - v[i + 1, 0] for i in st[:-1]])      # the launch behind emb_fwd: fc_fwd0
print("%d steps stamped" % len(spans))
for b in range(0, len(spans) - 19, 20):
    print("steps %3d-%3d: span %.1f us, fc_fwd0 %.2f us (%s)" % (b, b + 19, spans[b:b + 20].mean(), g0[b:b + 20].mean(), nm[st[0] + 1]))
