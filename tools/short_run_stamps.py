"""The first steps after an idle queue, from in-kernel stamps: per-step span (emb_fwd start to emb_fwd start) and the duration of every
main-chain kernel in steps 0..39 of a region that starts right behind a sync (bench.py's K = 20 case).   python tools/short_run_stamps.py"""
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
warm = int(os.environ.get("WARM", "25"))
for i in range(warm): gm.train_async(bs[i % 64])
gm.sync()
if os.environ.get("IDLE_MS"): time.sleep(1e-3 * float(os.environ["IDLE_MS"]))
L.ps_tune_set(b"stamps", 1)
K = 40
t0 = time.perf_counter()
for i in range(K): gm.train_async(bs[i % 64])
t1 = time.perf_counter()
gm.sync()
t2 = time.perf_counter()
cap = 8192
names = C.create_string_buffer(1 << 18)
vals = (C.c_ulonglong * (2 * cap))()
n = fn(names, len(names), vals, cap)
L.ps_tune_set(b"stamps", 0)
nm = names.value.decode().split("\n")[:n]
v = np.array(vals[:2 * n], np.int64).reshape(n, 2) / 100.0
st = [i for i, x in enumerate(nm) if x == "emb_fwd"]
print("host: enqueue %.0f us, closing wait %.0f us; GPU first start -> last end %.0f us" % (1e6 * (t1 - t0), 1e6 * (t2 - t1), v[:, 1].max() - v[st[0], 0]))
spans = np.diff([v[i, 0] for i in st])
print("step spans (us):", " ".join("%.0f" % x for x in spans))
per = st[1] - st[0]
for k in (0, 1, 2, 5, 10, 20, 38):
    seg = range(st[k], st[k] + per)
    print("step %2d:" % k, "  ".join("%s %.1f" % (nm[i][:10], v[i, 1] - v[i, 0]) for i in seg if nm[i] in ("emb_fwd", "gemm_nt", "head_last_bwd", "emb_bwd_update", "gemm_tn", "dense_update", "field_sort")))
