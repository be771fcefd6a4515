"""Parser / ingest throughput: CTR-shaped libsvm (label + 26 idx:1 + 13 idx:val per line) from memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd

rng = np.random.default_rng(1)
F, X, B, n = 26, 13, 4096, 4096 * 32
E = rng.integers(0, 100000, size=(n, F))
Xd = rng.standard_normal((n, X))
Y = (rng.random(n) < 0.25).astype(int)
lines = []
for i in range(n):
    lines.append(str(Y[i]) + " " + " ".join("%d:1" % v for v in E[i]) + " " + " ".join("%d:%.6f" % (F + 1 + j, Xd[i, j]) for j in range(X)))
text = ("\n".join(lines) + "\n").encode()
print("text: %d lines, %.1f MB" % (n, len(text) / 1e6))
for th in (1, 4, 16, 64):
    p = ps_amd.LibsvmParser(F, X, 100000, threads=th)
    t0 = time.perf_counter(); out = p.parse(text); dt = time.perf_counter() - t0
    print("host parse, %2d threads: %.1f ms  %.2f M lines/s  %.0f MB/s" % (th, 1e3 * dt, n / dt / 1e6, len(text) / dt / 1e6))
if len(sys.argv) > 1 and sys.argv[1] == "gpu":
    kv = ps_amd.KVStore(0, 1); kv.create_embedding([100000] * F, 16)
    gm = ps_amd.WideDeepNN.buildModel(F, 16, X, [512, 256, 1], 100000, store=kv, max_batch=B)
    for th in (4, 16, 32):
        ds = ps_amd.DataSet(kv, text, F, X, B, wide_size=100000, threads=th)
        for ep in range(3):
            t0 = time.perf_counter(); k = 0
            for b in ds:
                gm.train_async(b); k += 1
            gm.sync(); dt = time.perf_counter() - t0
            ds.reset()
        print("train from the pipeline, %2d parser threads: %.3f ms/step, %.2f M examples/s (%d steps)" % (th, 1e3 * dt / k, n / dt / 1e6, k))
        ds.close()
