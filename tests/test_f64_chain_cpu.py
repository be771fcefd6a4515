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


@pytest.mark.parametrize("wide,emb_ftrl", [(False, False), (True, False)])
def test_the_chains_float32_floors_hold_for_the_oracle(orc, wide, emb_ftrl):
    """The floors Chain.step propagates (how far ANY correctly rounded float32 evaluation may lie from the exact chain) must hold for
    the C oracle -- a float32 evaluation in the reference's own order -- over several steps, with NO other allowance than
    north_star's 1e-5 relative: forward, loss, deltas, dW / db, per-key gradients, and every parameter after its update.
    (They are what bound() adds for the HIP path in the GPU tests: tests/f64_chain.py.)"""
    F, D, X, fc, V, B, WS = 4, 8, 3, [12, 6, 1], 9, 40, 7
    rng = np.random.default_rng(11)
    st = orc.Store(SEED)
    om = orc.Model(st, orc.WIDEDEEP if wide else orc.DNN, F, D, X, fc, wide_size=WS)
    om.set_grad_mode(orc.GRAD_COMPAT, orc.GRAD_COMPAT, 0)
    ch = Chain(wide, F, D, X, fc, WS, emb_updater="ftrl" if emb_ftrl else "adam")
    dims = [F * D + X] + fc
    ch.load_fc([orc.init_dense(SEED, orc.TABLE_FC(l), dims[l] * dims[l + 1], orc.xavier_scale(dims[l], dims[l + 1])) for l in range(3)],
               [orc.init_dense(SEED, orc.TABLE_FC(l) + 1, dims[l + 1], orc.xavier_scale(dims[l], 1)) for l in range(3)])
    xav = orc.xavier_scale(1, D)

    def within(x_orc, x64, floor, what):
        x_orc, x64 = np.asarray(x_orc, np.float64), np.asarray(x64, np.float64)
        ex = np.abs(x_orc - x64) - 1e-5 * np.abs(x64) - np.asarray(floor, np.float64)
        assert ex.max() <= 0, "%s: the oracle is %.3e outside the chain's float32 floor (max floor %.3e, max err %.3e)" % (
            what, ex.max(), float(np.max(floor)), np.abs(x_orc - x64).max())
        return float(np.max(np.abs(x_orc - x64) / (1e-5 * np.abs(x64) + np.asarray(floor, np.float64) + 1e-300)))

    worst = 0.0
    for step in range(4):
        E = rng.integers(0, V, size=(B, F)).astype(np.int64)
        E[1] = E[0]
        Xd = rng.standard_normal((B, X)).astype(f32)
        Y = (rng.random(B) < 0.3).astype(f32)
        Wd = E % WS
        loss_o = om.train(E.astype(f32), Xd, Y, Wd.astype(f32) if wide else None, do_update=False)
        o = ch.step(E, Xd, Y, Wd if wide else None, lambda f, ids: orc.init_rows(SEED, f, ids, D, xav))
        tag = "step %d: " % step
        for l in range(3):
            worst = max(worst, within(om.act(2 + l), o["A"][l + 1], o["e_A"][l + 1], tag + "fc%d A" % l))
        worst = max(worst, within(om.p(), o["P"], o["e_P"], tag + "P"))
        worst = max(worst, within(loss_o, o["loss"], o["e_loss"], tag + "loss"))
        for l in range(3):
            d_o = om.delta(2 + l)
            if l == 0:
                d_o = d_o[:, :F * D] * (om.act(0) > 0)
            worst = max(worst, within(d_o, o["delta"][l], o["e_delta"][l], tag + "delta %d" % l))
            worst = max(worst, within(om.grad("fc%d.weights" % l), o["dW"][l].reshape(-1), o["e_dW"][l].reshape(-1), tag + "dW%d" % l))
            worst = max(worst, within(om.grad("fc%d.bias" % l), o["db"][l], o["e_db"][l], tag + "db%d" % l))
        for f in range(F):
            ids, g = o["geff"][f]
            go = np.stack([om.grad(orc.emb_key(f, float(i))) for i in ids])
            worst = max(worst, within(go, g, o["e_geff"][f], tag + "g_eff field %d" % f))
        om.apply_update()
        for f in range(F):
            ids = np.array(sorted(ch.rows[f]))
            wo = np.stack([st.get(orc.emb_key(f, float(i))) for i in ids])
            worst = max(worst, within(wo, np.stack([ch.rows[f][int(i)][0] for i in ids]), ch.floor_rows(f, ids), tag + "rows of field %d" % f))
        for l in range(3):
            worst = max(worst, within(st.get("fc%d.weights" % l), ch.W[l].reshape(-1), ch.floor_W(l).reshape(-1), tag + "fc%d.weights" % l))
            worst = max(worst, within(st.get("fc%d.bias" % l), ch.b[l], ch.floor_b(l), tag + "fc%d.bias" % l))
        if wide:
            k = np.nonzero(ch.seen)[0]
            wo = np.array([st.get(orc.wide_key(float(i)))[0] for i in k])
            worst = max(worst, within(wo, ch.ww[k], ch.floor_wide(k), tag + "wide weights"))
            worst = max(worst, within(st.get("wide.bias")[0], ch.wb[0], ch.floor_wide_bias()[0], tag + "wide.bias"))
    assert worst <= 1.0


@pytest.mark.parametrize("kind", ["adam", "ftrl"])
def test_updater_floors_hold_for_the_oracles_updaters(orc, kind):
    """adam_floor / ftrl_floor: the oracle's float32 updater, fed inputs that lie within (e_w, e_g, e_s1, e_s2) of the float64
    chain's, ends within the floor of the float64 result -- three chained updates, gradients from 1e-9 to 1 (where g ~ eps Adam's
    quotient is ill-conditioned: the floor must carry that), exact zeros (Ftrl's skip), rows of 8."""
    from f64_chain import adam, ftrl, adam_floor, ftrl_floor
    rng = np.random.default_rng(5)
    R, D = 400, 8
    w64 = rng.standard_normal((R, D)).astype(f32).astype(np.float64); s1 = np.zeros((R, D)); s2 = np.zeros((R, D))
    ew = np.zeros((R, D)); e1 = np.zeros((R, D)); e2 = np.zeros((R, D))
    w32, a32, b32 = w64.astype(f32), s1.astype(f32), s2.astype(f32)
    for step in range(3):
        g64 = (rng.standard_normal((R, D)) * 10.0 ** rng.uniform(-9, 0, size=(R, 1))).astype(f32).astype(np.float64)
        g64[::7] = 0.0                                            # whole rows of exact zeros: Ftrl skips them on both sides
        eg = np.abs(g64) * 3e-6 * rng.random((R, D))             # the float32 side's gradient: within eg of the chain's
        g32 = (g64 + eg * rng.choice([-1.0, 1.0], size=(R, D))).astype(f32)
        eg = np.abs(g32.astype(np.float64) - g64)
        if kind == "adam":
            fw, f1, f2 = adam_floor(w64, g64, s1, s2, ew, eg, e1, e2)
            w64, s1, s2 = adam(w64, g64, s1, s2)
            out = [orc.adam_update(w32[r], g32[r], a32[r], b32[r]) for r in range(R)]
        else:
            fw, f1, f2 = ftrl_floor(w64, g64, s1, s2, ew, eg, e1, e2)
            w64, s1, s2 = ftrl(w64, g64, s1, s2)
            out = [orc.ftrl_update(w32[r], g32[r], a32[r], b32[r])[:3] for r in range(R)]
        w32 = np.stack([o[0] for o in out]); a32 = np.stack([o[1] for o in out]); b32 = np.stack([o[2] for o in out])
        for name, x32, x64, fl in (("w", w32, w64, fw), ("s1", a32, s1, f1), ("s2", b32, s2, f2)):
            ex = np.abs(x32.astype(np.float64) - x64) - 1e-5 * np.abs(x64) - fl
            assert ex.max() <= 0, "%s step %d: %s is %.3e outside its floor" % (kind, step, name, ex.max())
        ew, e1, e2 = fw, f1, f2
