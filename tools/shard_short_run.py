"""The sharded step's first steps behind a sync (the driver's multi-GPU lines are --steps 20): host time of every ps_shard_step_finish_begin
call of a 40-step region after 300 priming steps, N = 1 (device copies).      python tools/shard_short_run.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = os.environ.get("PRE", "none")          # what runs in front of the region: none | nccl (torch.distributed barrier + all-reduce, as bench.py --sharded) | idle5 (5 ms of idle GPU)
if mode == "nccl":                            # (torch's HIP runtime first, as in bench.py --sharded)
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
import ps_amd
from ps_amd.sharded import NativeWorker
from bench import C2, synth_batch
cfg = dict(C2)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
rng = np.random.default_rng(1)
bs = [ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)) for _ in range(32)]
wk = NativeWorker([gm], 1, 0)
wk.run(bs, 300); kv.sync()
for rep in range(6):
    K = int(os.environ.get("K", "20"))
    if mode == "nccl":
        sub = os.environ.get("SUB", "full")
        kv.sync(); torch.cuda.synchronize(); dist.barrier()
        if sub == "full":
            t = torch.tensor([1.0], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); float(t.item())
        elif sub == "barrier_sync":
            torch.cuda.synchronize()
    elif mode == "idle5":
        time.sleep(0.005)
    ts = [time.perf_counter()]
    wk.begin(0, bs[0], side=False); ts.append(time.perf_counter())
    for i in range(K):
        wk.finish_begin(0, bs[(i + 1) % 32] if i + 1 < K else None, False); ts.append(time.perf_counter())
    kv.sync(); ts.append(time.perf_counter())
    d = np.diff(ts) * 1e6
    print("begin %.0f us | finish_begin: %s | closing sync %.0f us | total %.0f us = %.1f us/step" % (d[0], " ".join("%.0f" % x for x in d[1:-1]), d[-1], 1e6 * (ts[-1] - ts[0]), 1e6 * (ts[-1] - ts[0]) / K))
