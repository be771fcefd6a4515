"""In-workgroup K-split GEMM variants (gemm_nt_cfg 20-22, gemm_tn_cfg 6-7) against the default tiles: the same three
training steps, losses and tables agree to float32 roundoff (another summation order, not another result); then the sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
L = N.lib()
def run(knobs, shape):
    F, D, X, fc, V, B, WS = shape
    for k, v in knobs.items(): L.ps_tune_set(k.encode(), v)
    rng = np.random.default_rng(5)
    kv = ps_amd.KVStore(0, 0x5EED); kv.create_embedding([V] * F, D)
    gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
    losses = []
    for _ in range(3):
        E = rng.integers(0, V, (B, F)).astype(np.int64)
        losses.append(gm.train({"E": E, "X": rng.standard_normal((B, X)).astype(np.float32), "Y": (rng.random(B) < 0.3).astype(np.float32), "W": E % WS}))
    out = (losses, [kv.get("fc%d.weights" % i) for i in range(len(fc))], kv.get_rows(0, np.arange(V)))
    gm.close(); kv.close()
    for k in knobs: L.ps_tune_set(k.encode(), 0)
    return out
for shape in [(26, 16, 13, [512, 256, 1], 1000, 4096, 97), (5, 8, 3, [40, 24, 1], 50, 333, 11), (3, 4, 2, [5, 3, 1], 7, 6, 5)]:
    ref = run({}, shape)
    for knobs in ({"gemm_nt_cfg": 20}, {"gemm_nt_cfg": 21}, {"gemm_nt_cfg": 30}, {"gemm_tn_cfg": 2}, {"gemm_tn_cfg": 7}, {"gemm_nt_cfg": 30, "gemm_tn_cfg": 6}):
        got = run(knobs, shape)
        dl = max(abs(a - b) / abs(b) for a, b in zip(got[0], ref[0]))
        dw = max(np.abs(a - b).max() for a, b in zip(got[1], ref[1]))
        dr = np.abs(got[2] - ref[2]).max()
        print(shape[:5], knobs, "loss rel %.2e  fc max %.2e  rows max %.2e" % (dl, dw, dr), "OK" if dl < 2e-5 and dw < 2e-4 and dr < 2e-4 else "MISMATCH")
