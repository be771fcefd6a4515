"""Every workgroup's start / end of the last k_emb_reduce_update launch (library built with -DPS_EMB_TIMING: tools/emb_timing.sh).
MULTI_HOT=1: configs[4]'s shape (chunked order); otherwise configs[1] (sequential order)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
L = N.lib()
for kv_ in os.environ.get("PS_TUNE", "").split(","):
    if "=" in kv_: L.ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
fn = L.ps_dbg_emb_timing
fn.argtypes = [C.POINTER(C.c_ulonglong)]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
rng = np.random.default_rng(1)
nb = 8
if os.environ.get("MULTI_HOT"):
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
for rep in range(3):
    for i in range(100 + rep): gm.train_async(bs[i % nb])
    gm.sync()
    buf = (C.c_ulonglong * (8192 * 2))()
    assert fn(buf) == 0
    t = np.array(buf[:], np.int64).reshape(8192, 2)
    n = int((t[:, 0] > 0).sum())
    t = t[:n] / 100.0
    t0 = t[:, 0].min()
    end = t[:, 1] - t0
    print("launch of %d workgroups: starts spread %.1f us, last end %.1f us; ends: median %.1f, 90%% %.1f, 99%% %.1f" %
          (n, t[:, 0].max() - t0, end.max(), np.median(end), np.percentile(end, 90), np.percentile(end, 99)))
    print("   last end by XCD (workgroup %% 8):", " ".join("%.1f" % end[x::8].max() for x in range(8)))
    print("   median end by XCD:             ", " ".join("%.1f" % np.median(end[x::8]) for x in range(8)))
    dur = t[:, 1] - t[:, 0]
    worst = np.argsort(-end)[:6]
    print("   latest workgroups:", ", ".join("%d (XCD %d): %.1f -> %.1f" % (w, w % 8, t[w, 0] - t0, end[w]) for w in worst))
    print("   busy fraction of the launch (sum of workgroup durations / (grid x span)): %.2f" % (dur.sum() / (n * end.max())))
