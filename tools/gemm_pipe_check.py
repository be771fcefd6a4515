"""The software-pipelined slab loop of k_gemm_nt (gemm_nt_cfg 45-60) against the same tiles without it (5, 6, 7, 8, 13, 20):
three training steps each on several shapes, losses and tables must agree BIT FOR BIT (same products, same order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from ps_amd import native as N
L = N.lib()
def run(cfg, shape, tn=0):
    F, D, X, fc, V, B, WS = shape
    L.ps_tune_set(b"gemm_nt_cfg", cfg)
    L.ps_tune_set(b"gemm_tn_cfg", tn)
    rng = np.random.default_rng(5)
    kv = ps_amd.KVStore(0, 0x5EED); kv.create_embedding([V] * F, D)
    gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
    losses = []
    for _ in range(3):
        E = rng.integers(0, V, (B, F)).astype(np.int64)
        losses.append(gm.train({"E": E, "X": rng.standard_normal((B, X)).astype(np.float32), "Y": (rng.random(B) < 0.3).astype(np.float32), "W": E % WS}))
    out = (losses, [kv.get("fc%d.weights" % i) for i in range(len(fc))], kv.get_rows(0, np.arange(V)))
    gm.close(); kv.close()
    L.ps_tune_set(b"gemm_nt_cfg", 0)
    L.ps_tune_set(b"gemm_tn_cfg", 0)
    return out
bad = 0
for shape in [(26, 16, 13, [512, 256, 1], 1000, 4096, 97), (5, 8, 3, [40, 24, 1], 50, 333, 11), (3, 4, 2, [5, 3, 1], 7, 6, 5), (9, 8, 1, [130, 70, 1], 40, 1000, 13), (2, 4, 0, [8, 1], 9, 70, 3)]:
    for base, pipe in ((5, 45), (6, 46), (7, 47), (8, 48), (13, 53), (20, 60), (5, 85), (6, 86), (13, 93), (20, 90), (5, 105), (13, 113), (20, 120), (5, 125), (13, 133)):
        ref, got = run(base, shape), run(pipe, shape)
        same = ref[0] == got[0] and all(np.array_equal(a, b) for a, b in zip(ref[1], got[1])) and np.array_equal(ref[2], got[2])
        bad += not same
        print(shape[:5], "cfg %d vs %d:" % (base, pipe), "bit-identical" if same else "MISMATCH  losses %r vs %r" % (ref[0], got[0]))
    for base, pipe in ((2, 12), (6, 16), (7, 17)):          # k_gemm_tn's pipelined loop
        ref, got = run(0, shape, base), run(0, shape, pipe)
        same = ref[0] == got[0] and all(np.array_equal(a, b) for a, b in zip(ref[1], got[1])) and np.array_equal(ref[2], got[2])
        bad += not same
        print(shape[:5], "tn cfg %d vs %d:" % (base, pipe), "bit-identical" if same else "MISMATCH  losses %r vs %r" % (ref[0], got[0]))
# k_gemm_nt16 (16x16x4 MFMAs): another summation grouping inside the instruction, so float32 roundoff apart, not bit-equal
for shape in [(26, 16, 13, [512, 256, 1], 1000, 4096, 97), (5, 8, 3, [40, 24, 1], 50, 333, 11), (3, 4, 2, [5, 3, 1], 7, 6, 5), (9, 8, 1, [130, 70, 1], 40, 1000, 13), (2, 4, 0, [8, 1], 9, 70, 3)]:
    ref = run(5, shape)
    for cfg in (65, 66, 67, 73):
        got = run(cfg, shape)
        dl = max(abs(a - b) / abs(b) for a, b in zip(got[0], ref[0]))
        dw = max(np.abs(a - b).max() for a, b in zip(got[1], ref[1]))
        dr = np.abs(got[2] - ref[2]).max()
        ok = dl < 2e-5 and dw < 2e-4 and dr < 2e-4
        bad += not ok
        print(shape[:5], "cfg 5 vs %d: loss rel %.2e  fc max %.2e  rows max %.2e" % (cfg, dl, dw, dr), "OK" if ok else "MISMATCH")
print("FAILED" if bad else "ALL OK")
