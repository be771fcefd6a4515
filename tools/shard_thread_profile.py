"""Where the host time of the threaded sharded step goes (per call, both threads), N=1 over RCCL."""
import os, sys, time, threading, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
import ps_amd
from ps_amd import native as N
from ps_amd import sharded
from ps_amd.sharded import HipBackend, ShardedWorker, TorchComm
from bench import C2, synth_batch

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = dict(C2)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"], shard=0, nshards=1)
gms = [ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"]) for _ in range(4)]
torch.cuda.set_stream(torch.cuda.Stream(dev))
N.check(N.lib().ps_store_set_stream(kv.h, torch.cuda.current_stream().cuda_stream))
comm = TorchComm(dist, torch, dev, overlap=True)
be = HipBackend(gms, torch, dev)
wk = ShardedWorker(be, comm)
acc = collections.defaultdict(lambda: [0, 0.0])

def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        e = acc[(threading.current_thread().name, label or name)]
        e[0] += 1; e[1] += time.perf_counter() - t0
        return r
    setattr(obj, name, g)

for n in ("plan_launch", "plan_finish", "serve_pull", "forward_backward", "grads", "apply_push", "flat_grad", "apply_flat"):
    wrap(be, n)
for n in ("exchange_counts", "all_to_all_v", "all_reduce_sum_async", "join_side", "record_side", "record_done", "side_wait_for", "exchange_counts_launch", "exchange_counts_complete"):
    wrap(comm, n)
wrap(wk, "finish"); wrap(wk, "prepare"); wrap(wk, "prepare_launch"); wrap(wk, "prepare_counts"); wrap(wk, "prepare_complete")
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
for threaded in (False,):
    wk.run(bs, 300, threaded=threaded); torch.cuda.synchronize()
    acc.clear()
    n = 300
    t0 = time.perf_counter(); wk.run(bs, n, threaded=threaded); torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print("threaded=%s: %.1f us/step" % (threaded, 1e6 * tot / n))
    for k in sorted(acc, key=lambda k: (k[0], -acc[k][1])):
        print("   %-12s %-22s %4d calls  %7.1f us/step" % (k[0], k[1], acc[k][0], 1e6 * acc[k][1] / n))
dist.destroy_process_group()
