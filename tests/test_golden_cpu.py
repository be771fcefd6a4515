"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the oracle).  Here: the C oracle still reproduces them bit for bit (guards
oracle regressions; the independent numpy restatement is held bit-exact to the
oracle by test_oracle_twin.py, so it reproduces them as well)."""
import os

import numpy as np
import pytest

f32 = np.float32
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    m = [int(x) for x in z["meta"]]
    return z, dict(F=m[0], D=m[1], X=m[2], B=m[3], V=m[4], WS=m[5], wide=bool(m[6]), fc=m[7:], seed=int(z["seed"][0]))


@pytest.mark.parametrize("name", ["dnn.npz", "widedeep.npz"])
def test_golden_files_are_complete(name):
    z, c = load(name)
    for s in range(2):
        for k in ("E", "X", "Y", "loss", "embA", "concatA", "P", "emb_grad", "emb_W", "emb_M", "emb_V"):
            assert "s%d_%s" % (s, k) in z.files
        assert z["s%d_E" % s].shape == (c["B"], c["F"])
        assert np.isfinite(z["s%d_loss" % s]).all()
    # duplicates present (double-backward factor exercised)
    assert (z["s0_E"][1] == z["s0_E"][0]).all()


def _check_run(z, c, model, store, get_state):
    F, D, V = c["F"], c["D"], c["V"]
    nfc = len(c["fc"])
    for s in range(2):
        p = "s%d_" % s
        E, Xd, Y = z[p + "E"], z[p + "X"], z[p + "Y"]
        Wd = (E % c["WS"]).astype(f32) if c["wide"] else None
        loss = model.train(E.astype(f32), Xd, Y, Wd, do_update=False)
        assert f32(loss) == z[p + "loss"][0]
        np.testing.assert_array_equal(model.act(0), z[p + "embA"])
        np.testing.assert_array_equal(model.act(1), z[p + "concatA"])
        np.testing.assert_array_equal(model.p(), z[p + "P"])
        for l in range(nfc):
            np.testing.assert_array_equal(model.act(2 + l), z[p + "fc%d_A" % l])
            np.testing.assert_array_equal(model.delta(2 + l), z[p + "fc%d_delta" % l])
            np.testing.assert_array_equal(model.grad("fc%d.weights" % l), z[p + "fc%d_dW" % l])
            np.testing.assert_array_equal(model.grad("fc%d.bias" % l), z[p + "fc%d_db" % l])
        for f in range(F):
            for i in range(V):
                if z[p + "emb_touched"][f, i]:
                    np.testing.assert_array_equal(model.grad("emF%d.%d.0" % (f, i)), z[p + "emb_grad"][f, i])
        model.apply_update()
        for f in range(F):
            for i in range(V):
                if z[p + "emb_have"][f, i]:
                    w, m, v = get_state("emF%d.%d.0" % (f, i))
                    np.testing.assert_array_equal(w, z[p + "emb_W"][f, i])
                    if m is not None:
                        np.testing.assert_array_equal(m, z[p + "emb_M"][f, i])
                        np.testing.assert_array_equal(v, z[p + "emb_V"][f, i])
        for l in range(nfc):
            np.testing.assert_array_equal(get_state("fc%d.weights" % l)[0], z[p + "fc%d_w" % l])
            np.testing.assert_array_equal(get_state("fc%d.bias" % l)[0], z[p + "fc%d_b" % l])


@pytest.mark.parametrize("name", ["dnn.npz", "widedeep.npz"])
def test_oracle_reproduces_golden(orc, name):
    z, c = load(name)
    st = orc.Store(c["seed"])
    om = orc.Model(st, orc.WIDEDEEP if c["wide"] else orc.DNN, c["F"], c["D"], c["X"], c["fc"], wide_size=c["WS"])
    _check_run(z, c, om, st, lambda k: (st.get(k), st.state(k, 0), st.state(k, 1)))
    if c["wide"]:
        for i in range(c["WS"]):
            v = st.get(orc.wide_key(float(i)))
            if v is not None:
                assert v[0] == z["s1_wide_w"][i]
        assert st.get("wide.bias")[0] == z["s1_wide_bias"][0]
