"""Host-side time of the two halves of a sharded step (enqueue cost vs blocking), N=1 over RCCL."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
import ps_amd
from ps_amd import native as N
from ps_amd.sharded import HipBackend, ShardedWorker, TorchComm
from bench import C2, synth_batch

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = dict(C2)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"], shard=0, nshards=1)
gms = [ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"]) for _ in range(2)]
N.check(N.lib().ps_store_set_stream(kv.h, torch.cuda.current_stream().cuda_stream))
comm = TorchComm(dist, torch, dev, overlap=True)
wk = ShardedWorker(HipBackend(gms, torch, dev), comm)
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(8)]
wk.run(bs, 30); torch.cuda.synchronize()
tl = tf = tc = 0.0
n = 300
t00 = time.perf_counter()
p = wk.prepare(bs[0])
for i in range(n):
    t0 = time.perf_counter(); nxt = wk.prepare_launch(bs[(i + 1) % 8]); t1 = time.perf_counter()
    wk.finish(p, False); t2 = time.perf_counter()
    p = wk.prepare_complete(nxt); t3 = time.perf_counter()
    tl += t1 - t0; tf += t2 - t1; tc += t3 - t2
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print("per step: total %.1f us; host: plan launch %.1f us, finish(enqueue) %.1f us, prepare complete %.1f us" % (1e6 * tot / n, 1e6 * tl / n, 1e6 * tf / n, 1e6 * tc / n))
# enqueue-only cost of finish: sync before each so the GPU is idle and nothing blocks
tf2 = 0.0
for i in range(100):
    torch.cuda.synchronize(); t0 = time.perf_counter(); wk.finish(p, False); tf2 += time.perf_counter() - t0
    p = wk.prepare(bs[i % 8])
print("finish enqueue on an idle GPU: %.1f us" % (1e6 * tf2 / 100))
if not os.environ.get('PROFILE'):
    dist.destroy_process_group(); sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    nxt = wk.prepare_launch(bs[i % 8]); wk.finish(p, False); p = wk.prepare_complete(nxt)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
dist.destroy_process_group()
