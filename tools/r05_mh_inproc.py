"""Round 5 diagnosis: the multi-hot leg (bench.multi_hot_step) behind other work of the same process -- which of bench.py's
earlier legs makes it slow (0.59 ms in the bench line against 0.377 in a fresh process)?   python tools/r05_mh_inproc.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, ps_amd
from ps_amd import native as N
for kv_ in os.environ.get("PS_TUNE", "").split(","):
    if "=" in kv_:
        N.lib().ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
cfg = dict(bench.C2)
def mh(tag):
    r = bench.multi_hot_step(dict(cfg), 60)
    print("%-46s multi-hot %.4f ms/step (%s)" % (tag, r["ms_per_step"], r.get("stream_joins")), flush=True)
mh("fresh process")
mh("again")
# a fused model trained and closed (what the headline does)
rng = np.random.default_rng(1)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
bs = [ps_amd.DeviceBatch(kv, *bench.synth_batch(cfg, rng)) for _ in range(16)]
for i in range(300): gm.train_async(bs[i % 16])
gm.sync()
mh("beside a LIVE fused model (events expected)")
gm.set_profile(True)
for i in range(20): gm.train_async(bs[i % 16])
gm.sync(); gm.profile_report(); gm.set_profile(False)
for i in range(50): gm.train_async(bs[i % 16])
gm.sync()
for b in bs: b.close()
gm.close()
mh("fused model closed, its store still open")
ms, br, bw = C.c_double(), C.c_double(), C.c_double()
N.check(N.lib().ps_bench_gather(kv.h, 64 * 1000 * 1000, 64, 1 << 22, 1, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw)))
mh("after the gather benchmark on that store")
kv.close()
mh("store closed")
mh("again")
