"""One fused step as the GPU ran it, from in-kernel time stamps (ps_tune_set("stamps", 1): first workgroup start and
last workgroup end of every stamped launch, device wall clock).  The host pays nothing for them, so -- unlike under a
profiler's trace -- the streams stay ahead of the GPU exactly as in a normal run.
    python tools/gpu_timeline.py [rotating batches]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = N.lib()
fn = L.ps_dbg_stamps
for kv_ in os.environ.get("PS_TUNE", "").split(","):          # knobs read at model creation included
    if "=" in kv_: L.ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
fn.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
rng = np.random.default_rng(1)
if os.environ.get("MULTI_HOT"):           # configs[4]'s shape: Poisson(30) ids per (sample, field), FTRL rows
    kv.set_updater("emF", ps_amd.FtrlUpdater())
    B, F, V = cfg["B"], cfg["F"], cfg["V"]
    bs, nnz_max = [], 0
    for _ in range(nb):
        lens = np.clip(rng.poisson(30, size=B * F), 1, 100)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nnz_max = max(nnz_max, int(offsets[-1]))
        ids = __import__('ps_amd.synth', fromlist=['x']).draw_ids(rng, 1.05, V, int(offsets[-1]))
        bs.append(ps_amd.DeviceBatch(kv, ids, rng.standard_normal((B, cfg["X"])).astype(np.float32), (rng.random(B) < 0.25).astype(np.float32),
                                     rng.integers(0, cfg["wide"], size=(B, F)).astype(np.int64), offsets))
    gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=B, max_nnz=nnz_max)
else:
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(nb)]
for i in range(100): gm.train_async(bs[i % nb])
gm.sync()
L.ps_tune_set(b"stamps", 1)
for i in range(300): gm.train_async(bs[i % nb])
gm.sync()
cap = 8192
names = C.create_string_buffer(1 << 18)
vals = (C.c_ulonglong * (2 * cap))()
n = fn(names, len(names), vals, cap)
L.ps_tune_set(b"stamps", 0)
nm = names.value.decode().split("\n")[:n]
v = np.array(vals[:2 * n], np.int64).reshape(n, 2) / 100.0      # us
starts = [i for i, x in enumerate(nm) if x == "emb_fwd"]
main = {"emb_fwd", "gemm_nt", "fc_fwd_pair", "fwd_panel", "fwd_panel_head", "head_last_bwd", "emb_bwd_update"}
spans = np.diff([v[i, 0] for i in starts])
print("%d stamped launches, %d steps; step span median %.1f us (min %.1f, max %.1f)" % (n, len(starts) - 1, np.median(spans), spans.min(), spans.max()))
# average timeline over the steps whose span is within 2%% of the median
per = starts[1] - starts[0]
sel = [k for k in range(5, len(starts) - 1) if starts[k + 1] - starts[k] == per and abs(spans[k] - np.median(spans)) < 0.02 * np.median(spans)]
T = np.mean([v[starts[k]:starts[k] + per + 1] - v[starts[k], 0] for k in sel], axis=0)
print("mean over %d steps (time 0 = first workgroup of emb_fwd):" % len(sel))
prev_end = None
for i in range(per + 1):
    name = nm[starts[sel[0]] + i]
    on_main = name in main
    gap = ""
    if on_main and prev_end is not None:
        gap = "   gap %.1f" % (T[i, 0] - prev_end)
    print("%8.1f -> %8.1f (%5.1f)  %s%s%s" % (T[i, 0], T[i, 1], T[i, 1] - T[i, 0], "" if on_main else "          ", name if i < per else "emb_fwd (next step)", gap))
    if on_main:
        prev_end = T[i, 1]
