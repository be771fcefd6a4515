"""Does host-side load alone slow the fused step on the GPU?  The resident step (C loop, events: ps_model_time_steps) alone, beside N host
threads parsing libsvm text (no HIP call, ps_libsvm_parse releases the GIL), and beside N threads that only spin."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2, synth_batch
cfg = dict(C2); F, X, B, V = cfg["F"], cfg["X"], cfg["B"], cfg["V"]
rng = np.random.default_rng(5)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([V] * F, cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], X, cfg["fc"], cfg["wide"], store=kv, max_batch=B)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
for i in range(400): gm.train_async(bs[i % 8])
gm.sync()
def step_ms(n=1500):
    return gm.time_steps(bs[0], n) / n
E, Xd, Y, W = synth_batch(cfg, rng, B=4 * B)
text = b"\n".join((str(int(Y[i])) + " " + " ".join("%d:1" % v for v in E[i]) + " " + " ".join("%d:%.6f" % (F + 1 + j, Xd[i, j]) for j in range(X))).encode() for i in range(4 * B)) + b"\n"
print("alone:                                   %.4f ms/step" % step_ms())
for nthr in (8, 32, 96):
    stop = False
    def work():
        p = ps_amd.LibsvmParser(F, X, cfg["wide"], threads=1)
        while not stop:
            p.parse(text)
    th = [threading.Thread(target=work) for _ in range(nthr)]
    for t in th: t.start()
    time.sleep(0.3)
    print("beside %2d threads parsing text:          %.4f ms/step" % (nthr, step_ms()))
    stop = True
    for t in th: t.join()
print("alone again:                             %.4f ms/step" % step_ms())
