"""oracle/ps_oracle.c against the independent numpy restatement
(tests/np_restatement.py): two separately written restatements of the same
Java must agree BIT-FOR-BIT on every intermediate, gradient and updated
weight (string keys, by-reference aliasing, double backward, Adam, FTRL)."""
import numpy as np
import pytest

import np_restatement as npr

f32 = np.float32
SEED = 0x5EED


def _data(rng, B, F, X, V, dup=True):
    E = rng.integers(0, V, size=(B, F)).astype(f32)
    if dup and B >= 4:
        E[1] = E[0]            # whole-row duplicates -> n >= 2 for every field
        E[3, 0] = E[0, 0]      # n = 3 in field 0
    Xd = rng.standard_normal((B, X)).astype(f32)
    Y = (rng.random(B) < 0.3).astype(f32)
    return E, Xd, Y


def _make_pair(orc, wide, F, D, X, fc):
    st = orc.Store(SEED)
    om = orc.Model(st, orc.WIDEDEEP if wide else orc.DNN, F, D, X, fc)
    xav = orc.xavier_scale(1, D)
    kv = npr.KVStore()
    nm = npr.Model(
        kv, wide, F, D, X, fc,
        init_emb=lambda f, i: orc.init_rows(SEED, f, [i], D, xav).reshape(D, 1),
        init_fc_w=lambda i, out, inn: orc.init_dense(SEED, orc.TABLE_FC(i), out * inn, orc.xavier_scale(inn, out)).reshape(out, inn, order="F"),
        init_fc_b=lambda i, out: orc.init_dense(SEED, orc.TABLE_FC(i) + 1, out, orc.xavier_scale(
            (F * D + X) if i == 0 else fc[i - 1], 1)).reshape(out, 1),
    )
    return st, om, kv, nm


@pytest.mark.parametrize("wide", [False, True])
def test_twin_bit_exact(orc, wide):
    F, D, X, fc, B, V = 3, 4, 2, [5, 3, 1], 6, 5
    rng = np.random.default_rng(11 + wide)
    st, om, kv, nm = _make_pair(orc, wide, F, D, X, fc)
    for step in range(3):
        E, Xd, Y = _data(rng, B, F, X, V)
        Wd = np.array([[orc.matrixutil_hash(v, 4) for v in row] for row in E], f32) if wide else None
        loss_o = om.train(E, Xd, Y, Wd, do_update=False)
        grads_n = {}
        loss_n = nm.train(E.T.copy(), Xd.T.copy(), None if Wd is None else Wd.T.copy(), Y.reshape(1, B), grads_n)
        assert f32(loss_o) == f32(loss_n), step
        # forward intermediates
        np.testing.assert_array_equal(om.act(0), nm.embA.T)
        np.testing.assert_array_equal(om.act(1), nm.concatA.T)
        for li in range(len(fc)):
            np.testing.assert_array_equal(om.act(2 + li), nm.fc[li]["A"].T)
            np.testing.assert_array_equal(om.delta(2 + li), nm.fc[li]["delta"].T)
        np.testing.assert_array_equal(om.p(), nm.P.reshape(-1))
        # gradients per key (after /cnt), then weights and optimizer state
        keys = set(om.grad_keys())
        assert keys == set(grads_n.keys())
        for k in keys:
            go, gn = om.grad(k), grads_n[k].ravel(order="F")
            np.testing.assert_array_equal(go, gn, err_msg=k)
        om.apply_update()
        assert st.size() == len(kv.store)
        for k, w in kv.store.items():
            np.testing.assert_array_equal(st.get(k), w.ravel(order="F"), err_msg=k)
        adam = nm.updaters["default"]
        for k in adam.M:
            np.testing.assert_array_equal(st.state(k, 0), adam.M[k].ravel(order="F"))
            np.testing.assert_array_equal(st.state(k, 1), adam.V[k].ravel(order="F"))
        if wide:
            ft = nm.updaters["wide.weights"]
            for k in ft.Z:
                np.testing.assert_array_equal(st.state(k, 2), ft.Z[k])
                np.testing.assert_array_equal(st.state(k, 3), ft.N[k])


def test_double_backward_factor_in_model(orc):
    """App. A.6 inside the full model: the gradient handed to Adam for a key
    seen n times equals orc_emb_geff of its n masked per-sample gradients."""
    F, D, X, fc, B, V = 2, 4, 1, [4, 1], 8, 3
    rng = np.random.default_rng(5)
    st, om, kv, nm = _make_pair(orc, False, F, D, X, fc)
    E, Xd, Y = _data(rng, B, F, X, V)
    om.train(E, Xd, Y, None, do_update=False)
    delta = om.delta(2)            # [B][F*D+X]
    embA = om.act(0)
    for f in range(F):
        for idv in np.unique(E[:, f]):
            ks = np.nonzero(E[:, f] == idv)[0]
            gk = np.stack([delta[k, f * D:(f + 1) * D] * (embA[k, f * D:(f + 1) * D] > 0) for k in ks]).astype(f32)
            np.testing.assert_array_equal(om.grad(orc.emb_key(f, idv)), orc.emb_geff(gk, orc.GRAD_COMPAT))


def test_wide_gradient_touches_every_key_ever_seen(orc):
    """App. A.10: LRLayer.weights is never cleared, so keys seen in step 1 but
    not in step 2 still receive step 2's mean delta."""
    F, D, X, fc, B = 2, 2, 1, [3, 1], 4
    st = orc.Store(1)
    om = orc.Model(st, orc.WIDEDEEP, F, D, X, fc)
    rng = np.random.default_rng(2)
    E1 = np.array([[0, 1], [2, 3], [0, 1], [2, 3]], f32)
    E2 = np.array([[4, 5], [4, 5], [4, 5], [4, 5]], f32)
    Xd = rng.standard_normal((B, X)).astype(f32)
    Y = np.array([1, 0, 1, 0], f32)
    om.train(E1, Xd, Y, E1, do_update=True)
    om.train(E2, Xd, Y, E2, do_update=False)
    keys = set(k for k in om.grad_keys() if k.startswith("wide.weights."))
    assert keys == {orc.wide_key(v) for v in range(6)}
    g = {k: om.grad(k)[0] for k in keys}
    assert len(set(g.values())) == 1 and g[orc.wide_key(0.0)] == om.grad("wide.bias")[0]


def test_ps_bsp_mean_over_pushing_workers(orc):
    """net/PServer.java:164-214: BSP = mean over the workers that pushed the
    key, one updater step per key per global step; routing = Mod (floorMod)."""
    ps = orc.PS(4, seed=3)
    keys = [orc.emb_key(f, i) for f in range(3) for i in range(10)]
    w0 = {}
    rng = np.random.default_rng(0)
    for k in keys:
        w0[k] = rng.standard_normal(4).astype(f32)
        ps.shard(ps.route(k)).put(k, w0[k])
        assert ps.route(k) == orc.java_hashcode(k) % 4
    g1 = {k: rng.standard_normal(4).astype(f32) for k in keys}
    g2 = {k: rng.standard_normal(4).astype(f32) for k in keys[:10]}
    for k in keys:
        assert ps.push(k, g1[k]) == 200
    for k in g2:
        assert ps.push(k, g2[k]) == 200
    assert ps.push("emF0.0.0", g1[keys[0]], "simple@eta:0.1@") == 500     # no such updater
    assert ps.push("emF9.9.0", g1[keys[0]]) == 204                         # unknown key
    ps.barrier_update()
    assert ps.global_step() == 1
    for k in keys:
        g = ((g1[k] + g2[k]) / f32(2)) if k in g2 else g1[k] / f32(1)
        w, M, V = orc.adam_update(w0[k], g, np.zeros(4), np.zeros(4))
        np.testing.assert_array_equal(ps.shard(ps.route(k)).get(k), w)
