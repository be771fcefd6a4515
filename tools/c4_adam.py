"""BASELINE configs[3], fused-Adam variant: DNN over ONE table of R rows x 64 (R = 320 M: W + Adam
M,V = 246 GB), uniformly random ids.  B samples x 1 field x bags of `bag` ids -> nnz lookups per step.
Reports the step and the algorithmic HBM rates of the gather and of the fused backward + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ps_amd

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 320_000_000
bag = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B, D, X = (1 << 22) // bag, 64, 13
rng = np.random.default_rng(7)
kv = ps_amd.KVStore(0, 0x5EED)
t0 = time.perf_counter(); kv.create_embedding([R], D); kv.sync()
print("table: %d rows x %d, W + Adam state = %.1f GB, created in %.2f s" % (R, D, R * D * 4 * 3 / 1e9, time.perf_counter() - t0))
nnz = B * bag
gm = ps_amd.DNN.buildModel(1, D, X, [256, 64, 1], store=kv, max_batch=B, max_nnz=nnz)
batches = []
for _ in range(3):
    ids = rng.integers(0, R, size=nnz).astype(np.int64)
    offsets = (np.arange(B + 1) * bag).astype(np.int64)
    batches.append(ps_amd.DeviceBatch(kv, ids, rng.standard_normal((B, X)).astype(np.float32), (rng.random(B) < 0.25).astype(np.float32), None, offsets))
for i in range(4): gm.train_async(batches[i % 3])
gm.sync()
gm.set_profile(True)
for i in range(6): gm.train_async(batches[i % 3])
gm.sync(); prof = gm.profile_report(); gm.set_profile(False)
n = 20; t0 = time.perf_counter()
for i in range(n): gm.train_async(batches[i % 3])
gm.sync(); dt = (time.perf_counter() - t0) / n
us = {k: 1e3 * v[1] / max(v[0], 1) for k, v in prof.items()}
print("nnz/step %d (unique ~%d): %.3f ms/step = %.1f M lookups/s" % (nnz, len(np.unique(ids)), 1e3 * dt, nnz / dt / 1e6))
for k, v in sorted(us.items(), key=lambda kv_: -kv_[1])[:6]:
    print("  %-16s %9.1f us" % (k, v))
U = len(np.unique(ids))
rd = nnz * (4 * D + 8) + 8 * (B + 1)
bw = nnz * 4 * D + U * 6 * 4 * D + nnz * 8
print("gather: %.2f GB read  -> %.0f GB/s;   backward + Adam: %.2f GB (delta rows + 3 read + 3 written per key + ids) -> %.0f GB/s"
      % (rd / 1e9, rd / us["emb_fwd"] / 1e3, bw / 1e9, bw / us["emb_bwd_update"] / 1e3))
