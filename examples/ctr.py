#!/usr/bin/env python
"""CTR.java (BASELINE configs[0]) on the MI355X path: libsvm train/test files -> DataSet (host parser threads,
pinned double-buffered H2D) -> DNN.buildModel(23, 10, 45, {150, 10, 1}) or WideDeepNN -> epochs of
train + AUC on the test set, as CTR.main does (CTR.java:70-115).

    python examples/ctr.py --train train.txt --test test.txt [--wide 100000] [--epochs 3]
    python examples/ctr.py --synthetic 20000          # writes a CTR-format file first (the bundled train.txt is not in the repo)

Faithful to the reference's arithmetic, including what limits it: with its init scale 4*sqrt(6)/sqrt(in+out)
(layer/FcLayer.java:39-47, four times Xavier) and Adam at 0.005 the 10-unit ReLU layer of the {150, 10, 1}
tower is prone to dying within the first epoch (observed here: every unit <= 0 after ~15 steps, predictions
constant, test AUC stuck near 0.5), and the wide part gives every key the same batch-mean gradient (App. A.10).
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd  # noqa: E402


def synth_ctr_file(path, n, rng, F=23, X=45, V=1000):
    """label + F `idx:1` + X `idx:val` per line (CTR.java:55-61); the label depends on the features so AUC can rise."""
    w_id = rng.standard_normal((F, V)) * 0.6
    w_x = rng.standard_normal(X) * 0.3
    with open(path, "w") as f:
        for _ in range(n):
            ids = np.minimum(rng.zipf(1.3, F) - 1, V - 1)
            x = rng.standard_normal(X)
            z = w_id[np.arange(F), ids].sum() * 0.5 + x @ w_x - 1.0
            y = int(rng.random() < 1 / (1 + np.exp(-z)))
            f.write(str(y) + " " + " ".join("%d:1" % i for i in ids) + " " + " ".join("%d:%.6f" % (F + 1 + j, x[j]) for j in range(X)) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train"); ap.add_argument("--test")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--wide", type=int, default=0, help="wideSize: > 0 trains WideDeepNN (CTR.java:35)")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--vocab", type=int, default=1000, help="rows per embedding table (ids must be below it)")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    F, D, X, fc = 23, 10, 45, [150, 10, 1]
    if a.synthetic:
        d = tempfile.mkdtemp()
        a.train, a.test = os.path.join(d, "train.txt"), os.path.join(d, "test.txt")
        rng = np.random.default_rng(0)
        synth_ctr_file(a.train, a.synthetic, rng, F, X, a.vocab)
        synth_ctr_file(a.test, max(a.synthetic // 10, 100), rng, F, X, a.vocab)
    kv = ps_amd.KVStore.ins(0, 0x5EED)
    kv.create_embedding([a.vocab] * F, D)
    if a.wide:
        model = ps_amd.WideDeepNN.buildModel(F, D, X, fc, a.wide, store=kv, max_batch=1000)
    else:
        model = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=1000)
    train = ps_amd.DataSet(kv, a.train, F, X, 1000, wide_size=a.wide, threads=a.threads)      # batch 1000 (CTR.java:84)
    test = ps_amd.DataSet(kv, a.test, F, X, 100, wide_size=a.wide, threads=a.threads)         # batch 100  (CTR.java:87)
    for epoch in range(a.epochs):
        t0 = time.perf_counter(); n = 0; loss = 0.0
        for b in train:
            loss = model.train(b); n += b.B
        dt = time.perf_counter() - t0
        train.reset()
        ps, ys = [], []
        for b in test:                                   # CTR.auc: predict every test batch, AUC over all of them
            ps.append(model.predict(b)); ys.append(model.labels(b))
        test.reset()
        auc = ps_amd.AUC(np.concatenate(ps), np.concatenate(ys), store=kv).calculate()
        print("epoch %d  %d examples in %.3f s (%.0f ex/s)  last loss %.4f  test AUC %.4f" % (epoch, n, dt, n / dt, loss, auc))
    train.close(); test.close(); model.close(); kv.close()


if __name__ == "__main__":
    main()
