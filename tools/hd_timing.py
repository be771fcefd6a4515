"""Where a workgroup of k_gemm_nt_head (head_in_delta = 1) spends its time (library built with -DPS_HD_TIMING, see tools/hd_timing.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
L = N.lib()
L.ps_tune_set(b"head_in_delta", 1)
fn = L.ps_dbg_hd_timing
fn.argtypes = [C.POINTER(C.c_ulonglong)]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
for i in range(30): gm.train_async(bs[i % 8])
gm.sync()
buf = (C.c_ulonglong * (512 * 8))()
assert fn(buf) == 0
t = np.array(buf[:], np.int64).reshape(512, 8) / 100.0
t0 = t[:, 0].min()
print("starts spread %.1f us; kernel span %.1f us" % (t[:, 0].max() - t0, t[:, 7].max() - t0))
own = t[:, 5] > t[:, 3] - 1e-9
own &= t[:, 4] >= t[:, 3]
names = ["w_last + head sweep 0", "head sweep 1", "barrier", "slab body (owners)", "drain + barrier (owners)", "ticket (owners)", "GEMM"]
for k in range(7):
    if k in (3, 4):
        d = (t[:, k + 1] - t[:, k])[own]
    elif k == 5:
        d = (t[:, 6] - np.where(own, t[:, 5], t[:, 3]))
    else:
        d = t[:, k + 1] - t[:, k]
    print("%-28s mean %6.1f  p50 %6.1f  max %6.1f us" % (names[k], d.mean(), np.median(d), d.max()))
print("prologue end (GEMM start) after kernel start: mean %.1f, max %.1f; owners %d" % ((t[:, 6] - t0).mean(), (t[:, 6] - t0).max(), own.sum()))
