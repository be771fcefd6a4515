import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ps_amd
from ps_amd import native as N
from ps_amd.sharded import NativeWorker
from test_gpu_schedule import batches, SEED
L = N.lib()
def run(label, knobs={}, leak=False):
    for k, v in knobs.items(): L.ps_tune_set(k.encode(), v)
    F, D, X, fc, V, B, WS = 5, 16, 3, [32, 16, 1], 500, 512, 61
    rng = np.random.default_rng(21)
    data = batches(rng, 9, B, F, X, V, WS)
    res = []
    extra = None
    if leak:
        kvx = ps_amd.KVStore(0, SEED); kvx.create_embedding([V] * F, D)
        extra = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kvx, max_batch=B)
    for native in (False, True):
        kv = ps_amd.KVStore(0, SEED); kv.create_embedding([V] * F, D)
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
        bs = [ps_amd.DeviceBatch(kv, E, Xd, Y, W) for E, Xd, Y, W in data]
        if native:
            wk = NativeWorker([gm], 1, 0); wk.run(bs, 120); wk.close()
        else:
            for i in range(120): gm.train_async(bs[i % len(bs)])
        kv.sync()
        res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)]))
        for b in bs: b.close()
        gm.close(); kv.close()
    if extra: extra.close(); kvx.close()
    for k in knobs: L.ps_tune_set(k.encode(), 1)
    ok = all(np.array_equal(x, y) for x, y in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]))
    print(label, "OK" if ok else "MISMATCH", flush=True)
run("default")
run("dev_wait=0", {"dev_wait": 0})
run("leaked model (events)", leak=True)
run("shard_overlap=0", {"shard_overlap": 0})
run("dev_wait=0 again", {"dev_wait": 0})
run("default again")
if len(sys.argv) > 1:
    import pytest
    pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_parity.py")])
    run("after parity: default")
    run("after parity: tail_defer=0", {"tail_defer": 0})
    run("after parity: dev_wait=0", {"dev_wait": 0})
    run("after parity: shard_overlap=0", {"shard_overlap": 0})
    run("after parity: plan_early=0", {"plan_early": 0})
    run("after parity: default again")
