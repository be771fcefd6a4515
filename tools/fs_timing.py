"""Where k_field_sort_segments spends its time (library built with -DPS_FS_TIMING, see tools/fs_timing.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
fn = N.lib().ps_dbg_fs_timing
fn.argtypes = [C.POINTER(C.c_ulonglong)]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
for i in range(30): gm.train_async(bs[i % 8])
gm.sync()
buf = (C.c_ulonglong * (64 * 8))()
assert fn(buf) == 0
t = np.array(buf[:], np.int64).reshape(64, 8)[:cfg["F"], :6]
t0 = t[:, 0].min()
print("entry spread %.1f us; span first entry -> last exit %.1f us" % ((t[:, 0].max() - t0) / 100.0, (t[:, 5].max() - t0) / 100.0))
d = np.diff(t, axis=1) / 100.0
for name, col in zip(["key loads", "bitonic network", "heads + scans", "long runs + publish", "look-back wait", "output stores"], range(5)):
    print("%-22s mean %.1f  max %.1f us" % (name, d[:, col].mean(), d[:, col].max()))
