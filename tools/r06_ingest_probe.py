"""Why does the fused step run slower on batches that come out of the ingest ring?  ps_model_time_steps (a C loop of ps_model_train on ONE
batch, events around it) on (a) a DeviceBatch made from arrays, (b) batches handed out by ps_ingest_next, (c) a DeviceBatch made from the
arrays the ingest parsed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2); F, X, B, V = cfg["F"], cfg["X"], cfg["B"], cfg["V"]
rng = np.random.default_rng(5)
E, Xd, Y, W = synth_batch(cfg, rng, B=2 * B)
lines = [(str(int(Y[i])) + " " + " ".join("%d:1" % v for v in E[i]) + " " + " ".join("%d:%.6f" % (F + 1 + j, Xd[i, j]) for j in range(X))).encode() for i in range(2 * B)]
text = b"\n".join(lines) + b"\n"
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([V] * F, cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], X, cfg["fc"], cfg["wide"], store=kv, max_batch=B)
a = ps_amd.DeviceBatch(kv, E[:B], Xd[:B], Y[:B], W[:B])
for _ in range(300): gm.train_async(a)
gm.sync()
print("(a) DeviceBatch from arrays:           %.4f ms/step" % (gm.time_steps(a, 500) / 500))
ds = ps_amd.DataSet(kv, text, F, X, B, wide_size=cfg["wide"], threads=4)
b0 = ds.next(); b1 = ds.next()
print("(b) batch 0 out of the ingest ring:    %.4f ms/step" % (gm.time_steps(b0, 500) / 500))
print("(b) batch 1 out of the ingest ring:    %.4f ms/step" % (gm.time_steps(b1, 500) / 500))
p = ps_amd.LibsvmParser(F, X, cfg["wide"]).parse(text)
c = ps_amd.DeviceBatch(kv, p["E"][:B], p["X"][:B], p["Y"][:B], p["W"][:B])
print("(c) DeviceBatch from the parsed arrays: %.4f ms/step" % (gm.time_steps(c, 500) / 500))
print("(a) again:                              %.4f ms/step" % (gm.time_steps(a, 500) / 500))
print("max |X parsed - X| = %g, ids equal %s, labels equal %s" % (np.abs(p["X"][:B] - Xd[:B]).max(), np.array_equal(p["E"][:B], E[:B]), np.array_equal(p["Y"][:B], Y[:B])))
