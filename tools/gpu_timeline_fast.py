"""tools/gpu_timeline.py, but the mean timeline of the FASTEST quarter of the steps and of the slowest quarter, with the span's percentiles:
what differs between a good and a bad step of one run.     PS_TUNE=... python tools/gpu_timeline_fast.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
L = N.lib()
fn = L.ps_dbg_stamps
for kv_ in os.environ.get("PS_TUNE", "").split(","):
    if "=" in kv_: L.ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
fn.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
rng = np.random.default_rng(1)
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(16)]
for i in range(400): gm.train_async(bs[i % 16])
gm.sync()
L.ps_tune_set(b"stamps", 1)
for i in range(300): gm.train_async(bs[i % 16])
gm.sync()
cap = 8192
names = C.create_string_buffer(1 << 18)
vals = (C.c_ulonglong * (2 * cap))()
n = fn(names, len(names), vals, cap)
L.ps_tune_set(b"stamps", 0)
nm = names.value.decode().split("\n")[:n]
v = np.array(vals[:2 * n], np.int64).reshape(n, 2) / 100.0
starts = [i for i, x in enumerate(nm) if x == "emb_fwd"]
spans = np.diff([v[i, 0] for i in starts])
per = starts[1] - starts[0]
print("spans: p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f" % tuple(np.percentile(spans[5:], [5, 25, 50, 75, 95])))
ok = [k for k in range(5, len(starts) - 1) if starts[k + 1] - starts[k] == per]
order = sorted(ok, key=lambda k: spans[k])
for label, sel in (("fastest quarter", order[:len(order) // 4]), ("slowest quarter", order[-(len(order) // 4):])):
    T = np.mean([v[starts[k]:starts[k] + per + 1] - v[starts[k], 0] for k in sel], axis=0)
    print("%s (%d steps, mean span %.1f):" % (label, len(sel), np.mean([spans[k] for k in sel])))
    for i in np.argsort(T[:, 0], kind="stable"):
        print("%8.1f -> %8.1f (%5.1f)  %s" % (T[i, 0], T[i, 1], T[i, 1] - T[i, 0], nm[starts[sel[0]] + i] if i < per else "emb_fwd (next step)"))
