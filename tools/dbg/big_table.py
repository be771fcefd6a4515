import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ps_amd
from oracle import oracle as orc
f32 = np.float32
SEED = 0x5EED
for R in (1000 * 1000, 40 * 1000 * 1000, 320 * 1000 * 1000):
    D, X, bag = 64, 13, 32
    B = 1024
    nnz = B * bag
    rng = np.random.default_rng(7)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([R], D)
    gm = ps_amd.DNN.buildModel(1, D, X, [256, 64, 1], store=kv, max_batch=B, max_nnz=nnz)
    ids = rng.integers(0, R, size=nnz).astype(np.int64)
    offsets = (np.arange(B + 1) * bag).astype(np.int64)
    Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.25).astype(f32)
    xav = orc.xavier_scale(1, D)
    some = np.sort(ids[:6])
    print(R, "rows equal:", np.array_equal(kv.get_rows(0, some), orc.init_rows(SEED, 0, some, D, xav)), some)
    loss = gm.forward({"E": ids, "X": Xd, "Y": Y, "offsets": offsets})
    A0 = gm.act(0)
    print("A0 shape", A0.shape, "loss", loss)
    for b in (0, 5):
        acc = None
        for i in ids[b * bag:(b + 1) * bag]:
            r = kv.get_rows(0, [int(i)])[0]
            acc = r if acc is None else (r + acc).astype(f32)
        print("  bag", b, "== pooled kv rows:", np.array_equal(A0[b], np.maximum(acc, 0)), np.abs(A0[b] - np.maximum(acc, 0)).max())
    gm.close(); kv.close()
