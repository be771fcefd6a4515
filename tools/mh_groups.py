"""Measurement only: the kernel groups of the multi-hot step (configs[4]'s shape on one GPU), each alone, event-bracketed
(ps_model profile mode: one stream, no overlap).  python tools/mh_groups.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ps_amd

cfg = dict(bench.C2)
rng = np.random.default_rng(cfg["seed"] + 5)
B, F, V = cfg["B"], cfg["F"], cfg["V"]
kv = ps_amd.KVStore(0, cfg["seed"])
kv.create_embedding([V] * F, cfg["D"])
kv.set_updater("emF", ps_amd.FtrlUpdater())
bs, nnz_max = [], 0
for _ in range(4):
    lens = np.clip(rng.poisson(30, size=B * F), 1, 100)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(offsets[-1]); nnz_max = max(nnz_max, nnz)
    ids = __import__('ps_amd.synth', fromlist=['x']).draw_ids(rng, 1.05, V, nnz)
    W = rng.integers(0, cfg["wide"], size=(B, F)).astype(np.int64)
    bs.append(ps_amd.DeviceBatch(kv, ids, rng.standard_normal((B, cfg["X"])).astype(np.float32),
                                 (rng.random(B) < 0.25).astype(np.float32), W, offsets))
gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=B, max_nnz=nnz_max)
for i in range(8):
    gm.train_async(bs[i % 4])
gm.sync()
gm.set_profile(True)
for i in range(40):
    gm.train_async(bs[i % 4])
gm.sync()
rep = gm.profile_report()
gm.set_profile(False)
tot = 0.0
for k, v in rep.items():
    us = 1e3 * v[1] / max(v[0], 1)
    tot += us
    print("%-18s %8.1f us  (%d launches of the group)" % (k, us, v[0]))
print("sum %.1f us; nnz %d" % (tot, nnz_max))
