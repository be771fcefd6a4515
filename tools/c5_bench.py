"""BASELINE configs[4] shape on ONE GPU: C2's model with multi-hot bags (Poisson(30) ids per (sample, field),
sum pooling) and FTRL on every embedding row.  Reports the step time, the kernel groups and the gather's bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd
from bench import C2

from ps_amd import native as N_
for kv_ in os.environ.get("PS_TUNE", "").split(","):
    if "=" in kv_: N_.lib().ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
cfg = dict(C2); B, F, V = cfg["B"], cfg["F"], cfg["V"]
rng = np.random.default_rng(5)
kv = ps_amd.KVStore(0, cfg["seed"]); kv.create_embedding([V] * F, cfg["D"])
if len(sys.argv) > 1 and sys.argv[1] == "ftrl":
    kv.set_updater("emF", ps_amd.FtrlUpdater())
batches = []; nnz_tot = 0
for _ in range(4):
    lens = np.clip(rng.poisson(30, size=B * F), 1, 100)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(offsets[-1]); nnz_tot += nnz
    ids = (rng.integers(0, V, size=nnz) if 'uniform' in sys.argv else __import__('ps_amd.synth', fromlist=['x']).draw_ids(rng, 1.05, V, nnz, 'zipf_clamped' if 'clamped' in sys.argv else 'zipf_truncated')).astype(np.int64)
    uniq = len(np.unique(ids + V * (np.repeat(np.arange(B * F), lens) % F)))
    X = rng.standard_normal((B, cfg["X"])).astype(np.float32); Y = (rng.random(B) < 0.25).astype(np.float32)
    W = rng.integers(0, cfg["wide"], size=(B, F)).astype(np.int64)
    batches.append(ps_amd.DeviceBatch(kv, ids, X, Y, W, offsets))
nnz = nnz_tot // 4
gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=B, max_nnz=int(nnz * 1.1))
if "sharded" in sys.argv:          # the same batches through ps_shard_step with a 1-rank communicator
    from ps_amd.sharded import NativeWorker
    wk = NativeWorker(gm, 1, 0, is_async="async" in sys.argv)
    wk.run(batches, 20); kv.sync()
    t0 = time.perf_counter(); wk.run(batches, 100); kv.sync(); dt = (time.perf_counter() - t0) / 100
    print("sharded (N=1, ps_shard_step): nnz/step %d: %.3f ms/step, %.1f M ids/s" % (nnz, 1e3 * dt, nnz / dt / 1e6))
    gm.set_profile(True); wk.run(batches, 8); kv.sync(); prof = gm.profile_report(); gm.set_profile(False)
    for k, v in sorted(prof.items(), key=lambda kv_: -kv_[1][1])[:6]:
        print("  %-16s %8.1f us" % (k, 1e3 * v[1] / max(v[0], 1)))
    sys.exit(0)
for i in range(10): gm.train_async(batches[i % 4])
gm.sync()
gm.set_profile(True)
for i in range(12): gm.train_async(batches[i % 4])
gm.sync(); prof = gm.profile_report(); gm.set_profile(False)
t0 = time.perf_counter(); n = 100
for i in range(n): gm.train_async(batches[i % 4])
gm.sync(); dt = (time.perf_counter() - t0) / n
print("unique keys in the last batch: %d" % uniq)
print("nnz/step %d: %.3f ms/step, %.2f M examples/s, %.1f M ids/s" % (nnz, 1e3 * dt, B / dt / 1e6, nnz / dt / 1e6))
for k, v in sorted(prof.items(), key=lambda kv_: -kv_[1][1]):
    print("  %-16s %8.1f us" % (k, 1e3 * v[1] / max(v[0], 1)))
rd = nnz * (4 * cfg["D"] + 8) + 8 * (B * F + 1)
print("gather algorithmic read %.1f MB -> %.0f GB/s in emb_fwd" % (rd / 1e6, rd / (1e3 * prof["emb_fwd"][1] / prof["emb_fwd"][0]) / 1e3))
