"""Where k_fwd_panel<HEAD> spends its time (library built with -DPS_PANEL_TIMING, see tools/panel_timing.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
fn = N.lib().ps_dbg_panel_timing
fn.argtypes = [C.POINTER(C.c_ulonglong)]
N.lib().ps_tune_set(b"fwd_panel", 2)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
for i in range(300): gm.train_async(bs[i % 8])
gm.sync()
buf = (C.c_ulonglong * (256 * 16))()
assert fn(buf) == 0
t = np.array(buf[:], np.int64).reshape(256, 16)
t0 = t[:, 0].min()
names = ["x staged", "layer 0 MFMAs", "epilogue 0 + barrier", "layer 1 MFMAs", "epilogue 1 + barrier", "head", "barrier", "last layer backward"]
print("workgroups: start spread %.1f us; kernel span first entry -> last exit %.1f us" % ((t[:, 0].max() - t0) / 100.0, (t[:, 8].max() - t0) / 100.0))
d = np.diff(t[:, :9], axis=1) / 100.0
for i, n in enumerate(names): print("  %-24s mean %5.2f  max %5.2f us" % (n, d[:, i].mean(), d[:, i].max()))
print("  head: entry -> dot done %.2f, wide sum %.2f, sigmoid %.2f, loss terms + stores %.2f us" % tuple(np.mean(x) / 100.0 for x in (t[:, 12] - t[:, 5], t[:, 13] - t[:, 12], t[:, 14] - t[:, 13], t[:, 6] - t[:, 14])))
