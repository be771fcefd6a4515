"""Where k_last_bwd<true> spends its time (library built with -DPS_HEAD_TIMING, see tools/head_timing.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
from bench import C2, synth_batch
cfg = dict(C2)
fn = N.lib().ps_dbg_head_timing
fn.argtypes = [C.POINTER(C.c_ulonglong)]
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
for i in range(30): gm.train_async(bs[i % 8])
gm.sync()
buf = (C.c_ulonglong * (256 * 8))()
assert fn(buf) == 0
t = np.array(buf[:], np.int64).reshape(256, 8)[:128, :4]
t0 = t[:, 0].min()
d = np.diff(t, axis=1) / 100.0
print("workgroups: start spread %.1f us (first to last entry)" % ((t[:, 0].max() - t0) / 100.0))
print("per workgroup mean (max): head %.1f (%.1f), barrier %.1f (%.1f), backward %.1f (%.1f) us" % (d[:, 0].mean(), d[:, 0].max(), d[:, 1].mean(), d[:, 1].max(), d[:, 2].mean(), d[:, 2].max()))
t = np.array(buf[:], np.int64).reshape(256, 8)[:128]
print("head phase mean: entry->dot done %.1f, wide sum %.1f, sigmoid %.1f, loss terms + stores %.1f us" % tuple(np.mean(x) / 100.0 for x in (t[:, 4] - t[:, 0], t[:, 5] - t[:, 4], t[:, 6] - t[:, 5], t[:, 1] - t[:, 6])))
t = t[:, :4]
print("kernel span first entry -> last exit: %.1f us" % ((t[:, 3].max() - t0) / 100.0))
