"""What about 96 busy host threads slows the fused step (tools/r06_host_load_probe.py: 0.137 -> 0.203 ms)?  The same measurement with the
threads (a) only spinning (no memory traffic), (b) parsing at nice 19, (c) parsing pinned away from the launching thread's CPUs."""
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
step_ms = lambda n=1500: gm.time_steps(bs[0], n) / n
E, Xd, Y, W = synth_batch(cfg, rng, B=4 * B)
text = b"\n".join((str(int(Y[i])) + " " + " ".join("%d:1" % v for v in E[i]) + " " + " ".join("%d:%.6f" % (F + 1 + j, Xd[i, j]) for j in range(X))).encode() for i in range(4 * B)) + b"\n"
ncpu = os.cpu_count()
print("cpus %d, this thread may run on %d of them" % (ncpu, len(os.sched_getaffinity(0))))
print("alone: %.4f ms/step" % step_ms())
big = np.ones(1 << 22, np.float32)
def run(name, nthr, body, setup=None):
    stop = [False]
    def work():
        if setup: setup()
        while not stop[0]: body()
    th = [threading.Thread(target=work) for _ in range(nthr)]
    for t in th: t.start()
    time.sleep(0.4)
    print("%-58s %.4f ms/step" % (name, step_ms()))
    stop[0] = True
    for t in th: t.join()
p = ps_amd.LibsvmParser(F, X, cfg["wide"], threads=1)
run("beside 96 threads parsing", 96, lambda: p.parse(text))
run("beside 96 threads summing a 16 MB array (numpy, GIL released)", 96, lambda: big.sum())
run("beside 96 threads parsing at nice 19", 96, lambda: p.parse(text), lambda: os.setpriority(os.PRIO_PROCESS, threading.get_native_id(), 19))
allc = sorted(os.sched_getaffinity(0))
far = set(allc[len(allc) // 2:])
run("beside 96 threads parsing on the upper half of the CPUs", 96, lambda: p.parse(text), lambda: os.sched_setaffinity(threading.get_native_id(), far))
run("beside 48 threads parsing on the upper half of the CPUs", 48, lambda: p.parse(text), lambda: os.sched_setaffinity(threading.get_native_id(), far))
print("alone again: %.4f ms/step" % step_ms())
