"""The fused step beside an ingest pipeline that somebody ELSE drains (a second host thread takes its batches and drops them), and on
resident batches made from the arrays that pipeline parses -- separates "the pipeline's activity" from "its data"."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2); F, X, B, V = cfg["F"], cfg["X"], cfg["B"], cfg["V"]
rng = np.random.default_rng(cfg["seed"] + 77)
nlines, nbatch = 4 * B, 128
E, Xd, Y, _ = synth_batch(cfg, rng, B=nlines)
lines = np.array([(str(int(Y[i])) + " " + " ".join("%d:1" % v for v in E[i]) + " " + " ".join("%d:%.6f" % (F + 1 + j, Xd[i, j]) for j in range(X))).encode() for i in range(nlines)], dtype=object)
text = b"\n".join(lines[rng.integers(0, nlines, size=nbatch * B)]) + b"\n"
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([V] * F, cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], X, cfg["fc"], cfg["wide"], store=kv, max_batch=B)
res = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(64)]
def loop_ms(bs, n=1500):
    for i in range(200): gm.train_async(bs[i % len(bs)])
    gm.sync(); t0 = time.perf_counter()
    for i in range(n): gm.train_async(bs[i % len(bs)])
    gm.sync(); return 1e3 * (time.perf_counter() - t0) / n
print("resident batches (fresh synthetic), alone:                     %.4f ms/step" % loop_ms(res))
p = ps_amd.LibsvmParser(F, X, cfg["wide"], threads=16).parse(text)
res2 = [ps_amd.DeviceBatch(kv, p["E"][k * B:(k + 1) * B], p["X"][k * B:(k + 1) * B], p["Y"][k * B:(k + 1) * B], p["W"][k * B:(k + 1) * B]) for k in range(64)]
print("resident batches made from the leg's parsed text, alone:       %.4f ms/step" % loop_ms(res2))
for threads in (8, 32, 96):
    ds = ps_amd.DataSet(kv, text, F, X, B, wide_size=cfg["wide"], threads=threads)
    stop = [False]; cnt = [0]
    def drain():
        while not stop[0]:
            for _ in ds:
                cnt[0] += 1
                if stop[0]: break
            ds.reset()
    th = threading.Thread(target=drain); th.start(); time.sleep(0.3)
    c0, t0 = cnt[0], time.perf_counter()
    ms = loop_ms(res)
    rate = (cnt[0] - c0) / (time.perf_counter() - t0)
    print("resident batches beside a %2d-thread pipeline drained elsewhere: %.4f ms/step (the pipeline delivered %.0f batches/s)" % (threads, ms, rate))
    stop[0] = True; th.join(); ds.close()
