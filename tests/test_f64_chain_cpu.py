"""The float64 chain (tests/f64_chain.py) has the reference's semantics: against the C oracle (float32, the reference
restated) over three steps every quantity agrees to float32 roundoff -- DNN and Wide&Deep, duplicates, Adam and Ftrl.
(CPU only: this pins the yardstick the GPU tests' end-to-end tolerances are stated against.)"""
import numpy as np
import pytest

from f64_chain import Chain

f32 = np.float32
SEED = 0x5EED


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("wide", [False, True])
def test_chain_matches_oracle_semantics(orc, wide):
    F, D, X, fc, V, B, WS = 4, 8, 3, [12, 6, 1], 9, 40, 7
    rng = np.random.default_rng(3)
    st = orc.Store(SEED)
    om = orc.Model(st, orc.WIDEDEEP if wide else orc.DNN, F, D, X, fc, wide_size=WS)
    om.set_grad_mode(orc.GRAD_COMPAT, orc.GRAD_COMPAT, 0)
    ch = Chain(wide, F, D, X, fc, WS)
    dims = [F * D + X] + fc
    ch.load_fc([orc.init_dense(SEED, orc.TABLE_FC(l), dims[l] * dims[l + 1], orc.xavier_scale(dims[l], dims[l + 1])) for l in range(3)],
               [orc.init_dense(SEED, orc.TABLE_FC(l) + 1, dims[l + 1], orc.xavier_scale(dims[l], 1)) for l in range(3)])
    xav = orc.xavier_scale(1, D)
    for step in range(3):
        E = rng.integers(0, V, size=(B, F)).astype(np.int64)
        E[1] = E[0]
        Xd = rng.standard_normal((B, X)).astype(f32)
        Y = (rng.random(B) < 0.3).astype(f32)
        Wd = E % WS
        loss_o = om.train(E.astype(f32), Xd, Y, Wd.astype(f32) if wide else None, do_update=False)
        o = ch.step(E, Xd, Y, Wd if wide else None, lambda f, ids: orc.init_rows(SEED, f, ids, D, xav))
        assert rel(om.act(1), o["A"][0]) < 1e-6
        for l in range(3):
            assert rel(om.act(2 + l), o["A"][l + 1]) < 2e-5, (step, l)
        assert rel(om.p(), o["P"]) < 2e-5 and abs(loss_o - o["loss"]) < 2e-5 * o["loss"]
        for l in range(3):
            d_o = om.delta(2 + l)
            if l == 0:
                d_o = d_o[:, :F * D] * (om.act(0) > 0)
            assert rel(d_o, o["delta"][l]) < 1e-4, (step, l)
            assert rel(om.grad("fc%d.weights" % l), o["dW"][l].reshape(-1)) < 1e-4
            assert rel(om.grad("fc%d.bias" % l), o["db"][l]) < 1e-4
        for f in range(F):
            ids, g = o["geff"][f]
            for k, i in enumerate(ids):
                assert rel(om.grad(orc.emb_key(f, float(i))), g[k]) < 1e-4, (step, f, i)
        om.apply_update()
        # after the update: one Adam step is ill-conditioned where |g| ~ eps, so only a coarse agreement is asserted here
        for f in range(F):
            for i, r in ch.rows[f].items():
                assert np.abs(st.get(orc.emb_key(f, float(i))) - r[0]).max() < 5e-4
        for l in range(3):
            assert np.abs(st.get("fc%d.weights" % l) - ch.W[l].reshape(-1)).max() < 5e-4
        if wide:
            for k in np.nonzero(ch.seen)[0]:
                assert abs(st.get(orc.wide_key(float(k)))[0] - ch.ww[k]) < 5e-4
            assert abs(st.get("wide.bias")[0] - ch.wb[0]) < 5e-4
