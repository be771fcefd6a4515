#!/usr/bin/env python
"""Generate the committed golden vectors from the oracle (oracle/ps_oracle.c).

    python tests/golden/make_golden.py

The reference (Java + jblas) cannot run in this environment, so these are
outputs of the RESTATEMENT, not of the reference: they pin the oracle against
regressions and let the GPU tests compare against fixed numbers without
executing the oracle.  Two tiny models, two steps each, duplicates in every
field so the double-backward factor (App. A.6) is exercised:
  dnn.npz       DNN  F=3 D=4 X=2 FC[5,3,1] B=6
  widedeep.npz  W&D  same shape, wideSize 7, Adam + Ftrl
Stored: inputs per step, the initial tables/tensors, and per step: embedding /
concat / fc activations, P, loss, deltas, per-key gradients, updated weights and
optimizer state for every touched key.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402

f32 = np.float32
SEED = 0x5EED
F, D, X, FC, B, V, WS = 3, 4, 2, [5, 3, 1], 6, 5, 7


def make(wide):
    rng = np.random.default_rng(1234 + wide)
    st = orc.Store(SEED)
    om = orc.Model(st, orc.WIDEDEEP if wide else orc.DNN, F, D, X, FC, wide_size=WS)
    out = {"meta": np.array([F, D, X, B, V, WS, int(wide)] + FC, np.int64), "seed": np.array([SEED], np.uint64)}
    dims = [F * D + X] + FC
    for l in range(len(FC)):
        out["init_fc%d_w" % l] = orc.init_dense(SEED, orc.TABLE_FC(l), dims[l] * dims[l + 1], orc.xavier_scale(dims[l], dims[l + 1]))
        out["init_fc%d_b" % l] = orc.init_dense(SEED, orc.TABLE_FC(l) + 1, dims[l + 1], orc.xavier_scale(dims[l], 1))
    out["init_emb"] = np.stack([orc.init_rows(SEED, f, range(V), D, orc.xavier_scale(1, D)) for f in range(F)])
    for step in range(2):
        E = rng.integers(0, V, size=(B, F)).astype(np.int64)
        E[1] = E[0]; E[3, 0] = E[0, 0]
        Xd = rng.standard_normal((B, X)).astype(f32)
        Y = (rng.random(B) < 0.4).astype(f32)
        Wd = E % WS
        loss = om.train(E.astype(f32), Xd, Y, Wd.astype(f32) if wide else None, do_update=False)
        p = "s%d_" % step
        out[p + "E"], out[p + "X"], out[p + "Y"], out[p + "loss"] = E, Xd, Y, np.array([loss], f32)
        out[p + "embA"], out[p + "concatA"], out[p + "P"] = om.act(0), om.act(1), om.p()
        for l in range(len(FC)):
            out[p + "fc%d_A" % l] = om.act(2 + l)
            out[p + "fc%d_delta" % l] = om.delta(2 + l)
            out[p + "fc%d_dW" % l] = om.grad("fc%d.weights" % l)
            out[p + "fc%d_db" % l] = om.grad("fc%d.bias" % l)
        gk = np.zeros((F, V, D), f32); gm = np.zeros((F, V), np.int8)
        for f in range(F):
            for i in np.unique(E[:, f]):
                gk[f, i] = om.grad(orc.emb_key(f, float(i))); gm[f, i] = 1
        out[p + "emb_grad"], out[p + "emb_touched"] = gk, gm
        if wide:
            out[p + "wide_gbar"] = om.grad("wide.bias")
        om.apply_update()
        W = np.zeros((F, V, D), f32); M = np.zeros((F, V, D), f32); Vv = np.zeros((F, V, D), f32); have = np.zeros((F, V), np.int8)
        for f in range(F):
            for i in range(V):
                k = orc.emb_key(f, float(i))
                if st.get(k) is not None:
                    W[f, i] = st.get(k); have[f, i] = 1
                    if st.state(k, 0) is not None:
                        M[f, i], Vv[f, i] = st.state(k, 0), st.state(k, 1)
        out[p + "emb_W"], out[p + "emb_M"], out[p + "emb_V"], out[p + "emb_have"] = W, M, Vv, have
        for l in range(len(FC)):
            out[p + "fc%d_w" % l], out[p + "fc%d_b" % l] = st.get("fc%d.weights" % l), st.get("fc%d.bias" % l)
        if wide:
            ww = np.zeros(WS, f32); wz = np.zeros(WS, f32); wn = np.zeros(WS, f32)
            for i in range(WS):
                k = orc.wide_key(float(i))
                if st.get(k) is not None:
                    ww[i] = st.get(k)[0]
                    if st.state(k, 2) is not None:
                        wz[i], wn[i] = st.state(k, 2)[0], st.state(k, 3)[0]
            out[p + "wide_w"], out[p + "wide_z"], out[p + "wide_n"], out[p + "wide_bias"] = ww, wz, wn, st.get("wide.bias")
    return out


if __name__ == "__main__":
    for wide, name in ((0, "dnn.npz"), (1, "widedeep.npz")):
        d = make(wide)
        np.savez_compressed(os.path.join(HERE, name), **d)
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes,", len(d), "arrays")
