"""Parity of the HIP path (through the C ABI) against the oracle, on the same
seeded inputs.  Bit-exact for everything that is integer/copy/elementwise
(gather, relu, per-key reduction incl. the double-backward factor, Adam,
Ftrl); FP32 GEMM-fed quantities within 1e-5 relative (the tolerance
BASELINE.json's north_star states; jblas' sgemm order is unknowable)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

f32 = np.float32
SEED = 0x5EED
RTOL = 1e-5      # BASELINE.json north_star: 1e-5 relative on FP32 forward/backward
EPS = 2.0 ** -24
# End-to-end quantities (after 2 * nfc chained float32 GEMMs, after updates, after several steps) are bounded against the
# float64 chain of the same step (tests/f64_chain.py): |gpu - f64| <= RTOL (|f64| + max|f64|) + C max|oracle - f64| + the roundoff
# floor of the quantity's own last operation -- the HIP path may be off by north_star's 1e-5, plus a small multiple of
# what the reference-order float32 chain itself is off by.  No hand-picked absolute floors.
from f64_chain import Chain, bound  # noqa: E402


def close64(x, x64, mag, what):
    """|x - x64| <= 1e-5*|x64| + 8*eps*mag : mag = sum of |terms| of the contraction (f32 roundoff floor;
    exact-f32 MFMA measures ~2.5 eps * sum|a*b|, MI355X guide)."""
    err = np.abs(np.asarray(x, np.float64) - x64) - RTOL * np.abs(x64) - 8 * EPS * mag
    assert err.max() <= 0, "%s: excess %.3e" % (what, err.max())


def layerwise_f64(gm, wb, E, Y, F, D, X, fc, wide):
    """Each FcLayer forward / backward contraction checked in float64 on the GPU's own inputs."""
    nfc = len(fc)
    dims = [F * D + X] + list(fc)
    B = E.shape[0]
    A = [gm.act(1).astype(np.float64)] + [gm.act(2 + l).astype(np.float64) for l in range(nfc)]
    W = [wb["fc%d.weights" % l].astype(np.float64).reshape(dims[l], dims[l + 1]) for l in range(nfc)]
    b = [wb["fc%d.bias" % l].astype(np.float64) for l in range(nfc)]
    for l in range(nfc):
        z = A[l] @ W[l] + b[l]
        mag = np.abs(A[l]) @ np.abs(W[l]) + np.abs(b[l])
        if l < nfc - 1:
            close64(A[l + 1], np.maximum(z, 0), mag, "fc%d forward" % l)
        elif not wide:
            close64(A[l + 1], 0.001 + 0.998 / (1 + np.exp(-z)), mag, "fc%d forward (sigmoid)" % l)
        else:
            close64(A[l + 1], z, mag, "fc%d forward (logit)" % l)
    p = gm.p(B)
    d = ((p - Y) / (p * (1 - p)) * (p * (1 - p))).astype(f32).astype(np.float64).reshape(B, 1)   # head: CE' * sigmoid'
    for l in range(nfc - 1, -1, -1):
        close64(gm.fc_grad(l).reshape(dims[l], dims[l + 1]), A[l].T @ d / B, np.abs(A[l]).T @ np.abs(d) / B, "dW%d" % l)
        close64(gm.fc_grad(l, True), d.mean(0), np.abs(d).mean(0), "db%d" % l)
        dn = d @ W[l].T
        mag = np.abs(d) @ np.abs(W[l]).T
        if l == 0:
            dn, mag = dn[:, :F * D] * (A[0][:, :F * D] > 0), mag[:, :F * D]
        else:
            dn = dn * (A[l] > 0)
        got = gm.delta(2 + l).astype(np.float64)
        close64(got, dn, mag, "delta into fc%d" % l)
        d = got


def close(a, b, floor, rtol=RTOL, what=""):
    """|a - b| <= rtol |b| + floor, elementwise: north_star's 1e-5 relative plus `floor`, the float32 roundoff the two
    evaluations of b's chain may differ by -- propagated sums of |terms| (forward_floors below), never a fraction of the
    tensor's largest element (the floor used to be 1e-5 max|b|)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b) - rtol * np.abs(b) - floor
    assert err.max() <= 0, "%s: max excess %.3e (max|b| %.3e, max floor %.3e)" % (what, err.max(), np.abs(b).max(), np.max(floor))


def forward_floors(A, W, b, dA0=0.0, dW=None, db=None):
    """How far two float32 evaluations of the FC chain may lie apart, elementwise, layer by layer.
    A[l]: this side's input of layer l ([B][in], A[0] = the concat layer's output), W[l]: [in][out], b[l]: [out];
    dA0: |difference of the two sides' A[0]| (0 when both gathered the same rows), dW / db: |difference of their parameters|
    (0 at a first step).  With e_l the bound on |A_l - A_l'|:

        e_0 = dA0,   e_{l+1} = e_l |W_l| + |A_l| dW_l + db_l + 16 eps (|A_l| |W_l| + |b_l|)

    -- the inputs' difference carried through |W|, the parameters' difference, and BOTH sides' roundoff of the layer's
    own contraction (8 eps sum|terms| each: exact-f32 MFMA measures ~2.5 eps, a sequential sgemm loop more).  relu is
    1-Lipschitz, so the bound passes through it unchanged.  Returns [e_1 .. e_nfc] for the PRE-activation of the last layer."""
    e = np.zeros_like(np.asarray(A[0], np.float64)) + dA0
    out = []
    for l in range(len(W)):
        Al, Wl = np.abs(np.asarray(A[l], np.float64)), np.abs(np.asarray(W[l], np.float64))
        e = e @ Wl + 16 * EPS * (Al @ Wl + np.abs(np.asarray(b[l], np.float64)))
        if dW is not None:
            e = e + Al @ np.abs(np.asarray(dW[l], np.float64)) + np.abs(np.asarray(db[l], np.float64))
        out.append(e)
    return out


def head_floors(e_z, P, Y, e_wide=0.0):
    """... and through the head: P = clipped sigmoid(z [+ wide logit]) is 0.998 / 4-Lipschitz in its argument (+ 4 eps for its own
    evaluation); a loss term moves by at most |dP| / min(P, 1 - P) (+ the mean's own roundoff, 8 eps mean|terms|)."""
    P = np.asarray(P, np.float64).reshape(-1); Y = np.asarray(Y, np.float64).reshape(-1)
    e_P = 0.25 * (np.asarray(e_z, np.float64).reshape(-1) + e_wide) + 4 * EPS
    terms = -Y * np.log(P) - (1 - Y) * np.log(1 - P)
    e_loss = float(np.mean(e_P / np.minimum(P, 1 - P)) + 8 * EPS * np.mean(np.abs(terms)))
    return e_P, e_loss


def check_forward(gm, ref_act, ref_P, ref_loss, loss_g, Y, nfc, dims, wide, Wg, bg, Wr=None, br=None, e_wide=0.0, tag=""):
    """Forward activations, P and loss of the HIP path against a reference evaluation (the oracle, a golden file):
    1e-5 relative + forward_floors / head_floors.  Wg, bg: the HIP side's FC parameters; Wr, br: the reference's (None: the same)."""
    B = len(Y)
    A = [gm.act(1)] + [gm.act(2 + l) for l in range(nfc)]
    Wg = [np.asarray(w, np.float64).reshape(dims[l], dims[l + 1]) for l, w in enumerate(Wg)]
    dW = db = None
    if Wr is not None:
        dW = [np.abs(Wg[l] - np.asarray(Wr[l], np.float64).reshape(dims[l], dims[l + 1])) for l in range(nfc)]
        db = [np.abs(np.asarray(bg[l], np.float64) - np.asarray(br[l], np.float64)) for l in range(nfc)]
    fl = forward_floors(A[:nfc], Wg, bg, dA0=np.abs(A[0].astype(np.float64) - np.asarray(ref_act(1), np.float64)), dW=dW, db=db)
    for l in range(nfc):
        last = l == nfc - 1
        # (the DNN's last FcLayer applies the clipped sigmoid itself: 0.25-Lipschitz)
        f = (0.25 * fl[l] + 4 * EPS) if (last and not wide) else fl[l]
        close(A[l + 1], ref_act(2 + l), f, what="%sfc%d A" % (tag, l))
    e_z = fl[nfc - 1]
    p = gm.p(B)
    e_P, e_loss = head_floors(e_z, p, Y, e_wide)
    close(p, ref_P, e_P, what=tag + "P")
    close(loss_g, ref_loss, e_loss, what=tag + "loss")


def make_pair(orc, wide, F, D, X, fc, V, B, seed=SEED, **kw):
    import ps_amd
    st = orc.Store(seed)
    om = orc.Model(st, orc.WIDEDEEP if wide else orc.DNN, F, D, X, fc, wide_size=kw.get("wide_size", 1000))
    kv = ps_amd.KVStore(0, seed)
    kv.create_embedding([V] * F, D)
    if wide:
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, kw.get("wide_size", 1000), store=kv, max_batch=B)
    else:
        gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
    return st, om, kv, gm


def data(rng, B, F, X, V, zipf=False):
    if zipf:
        E = np.minimum(rng.zipf(1.3, size=(B, F)) - 1, V - 1).astype(np.int64)
    else:
        E = rng.integers(0, V, size=(B, F)).astype(np.int64)
    if B >= 4:
        E[1] = E[0]
        E[3, 0] = E[0, 0]
    Xd = rng.standard_normal((B, X)).astype(f32)
    Y = (rng.random(B) < 0.3).astype(f32)
    return E, Xd, Y


def test_init_matches_oracle(orc):
    """Counter-based init is the same pure function on both sides (bit-exact)."""
    import ps_amd
    F, D, V = 3, 8, 50
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    kv.create_fc(0, F * D + 2, 5)
    xav = orc.xavier_scale(1, D)
    for f in range(F):
        ids = np.arange(0, V, 7)
        np.testing.assert_array_equal(kv.get_rows(f, ids), orc.init_rows(SEED, f, ids, D, xav))
    w = kv.get("fc0.weights")
    np.testing.assert_array_equal(w, orc.init_dense(SEED, orc.TABLE_FC(0), 5 * (F * D + 2), orc.xavier_scale(F * D + 2, 5)))
    b = kv.get("fc0.bias")
    np.testing.assert_array_equal(b, orc.init_dense(SEED, orc.TABLE_FC(0) + 1, 5, orc.xavier_scale(F * D + 2, 1)))
    # string keys: "emF1.14.0"
    np.testing.assert_array_equal(kv.get(orc.emb_key(1, 14.0)), orc.init_rows(SEED, 1, [14], D, xav)[0])
    assert kv.get("emF1.%d.0" % (V + 3)) is None                  # absent key -> null
    kv.put("emF2.3.0", np.arange(D, dtype=f32))
    np.testing.assert_array_equal(kv.get_rows(2, [3])[0], np.arange(D, dtype=f32))
    kv.close()


CASES = [
    # wide, F, D, X, fc, V, B, zipf
    (False, 3, 4, 2, [5, 3, 1], 5, 6, False),          # tiny, heavy duplicates
    (True, 3, 4, 2, [5, 3, 1], 5, 6, False),
    (False, 23, 10, 45, [150, 10, 1], 40, 100, False), # CTR.java shape (C1): D=10 -> scalar lanes
    (True, 26, 16, 13, [64, 32, 1], 300, 256, True),   # Criteo-like, zipf duplicates, runs > 32 (long-key waves)
    (False, 2, 64, 3, [32, 1], 50, 96, True),          # configs[3]'s row width: 16 lanes per row
    (True, 3, 32, 0, [16, 8, 1], 20, 40, False),       # no dense features at all (X = 0), 8 lanes per row
]


@pytest.mark.parametrize("wide,F,D,X,fc,V,B,zipf", CASES)
def test_step_parity(orc, wide, F, D, X, fc, V, B, zipf):
    rng = np.random.default_rng(7)
    st, om, kv, gm = make_pair(orc, wide, F, D, X, fc, V, B, wide_size=97)
    om.set_grad_mode(orc.GRAD_COMPAT, orc.GRAD_COMPAT, 0)       # single-hot: the reference's sequential order on both sides
    nfc = len(fc)
    ch = Chain(wide, F, D, X, fc, 97)                           # the float64 chain, from the same initial parameters
    ch.load_fc([kv.get("fc%d.weights" % i) for i in range(nfc)], [kv.get("fc%d.bias" % i) for i in range(nfc)])
    worst = {}
    for step in range(3):
        E, Xd, Y = data(rng, B, F, X, V, zipf)
        Wd = (E % 97) if wide else None
        # state before the step (for the bit-exact updater check)
        uniq = [np.unique(E[:, f]) for f in range(F)]
        w0 = [kv.get_rows(f, uniq[f]) for f in range(F)]
        m0 = [kv.get_rows(f, uniq[f], 1) for f in range(F)]
        v0 = [kv.get_rows(f, uniq[f], 2) for f in range(F)]
        kv_before = {"fc%d.%s" % (i, k): kv.get("fc%d.%s" % (i, k)) for i in range(nfc) for k in ("weights", "bias")}
        st_before = {"fc%d.%s" % (i, k): np.array(st.get("fc%d.%s" % (i, k)), f32) for i in range(nfc) for k in ("weights", "bias")} if step else kv_before
        e_wide = 0.0
        if wide and step:      # how far the two sides' wide logits can lie apart: their wide weights' differences over a sample's ids
            wo = np.array([0.0 if st.get(orc.wide_key(float(k))) is None else st.get(orc.wide_key(float(k)))[0] for k in range(97)])
            bo = st.get("wide.bias")
            e_wide = np.abs(kv.get_wide(np.arange(97)).astype(np.float64) - wo)[Wd].sum(axis=1) + abs(float(kv.get("wide.bias")[0]) - (0.0 if bo is None else float(bo[0])))
        loss_o = om.train(E.astype(f32), Xd, Y, None if Wd is None else Wd.astype(f32), do_update=False)
        loss_g = gm.forward({"E": E, "X": Xd, "Y": Y, "W": Wd})
        # (ids the chain has not seen were never trained on the GPU either: their rows are still the initial ones)
        if step:        # the parameters the GPU holds at the start of this step: their distance to the chain's is the floors' input (f64_chain.Chain)
            ch.anchor([kv_before["fc%d.weights" % i] for i in range(nfc)], [kv_before["fc%d.bias" % i] for i in range(nfc)],
                      lambda f, ids: kv.get_rows(f, ids), kv.get_wide(np.arange(97)) if wide else None, kv.get("wide.bias") if wide else None)
        c64 = ch.step(E, Xd, Y, Wd, lambda f, ids: kv.get_rows(f, ids))
        # ---- forward: gather + relu + concat are copies -> bit-exact (when weights are)
        if step == 0:
            np.testing.assert_array_equal(gm.act(0), om.act(0))
            np.testing.assert_array_equal(gm.act(1), om.act(1))
        # (later steps: the rows were trained on both sides -- their distance is bounded against the float64 chain below, and
        #  enters the forward's bound as the inputs' difference)
        dims = [F * D + X] + list(fc)
        check_forward(gm, om.act, om.p(), loss_o, loss_g, Y, nfc, dims, wide,
                      [kv_before["fc%d.weights" % i] for i in range(nfc)], [kv_before["fc%d.bias" % i] for i in range(nfc)],
                      [st_before["fc%d.weights" % i] for i in range(nfc)] if step else None, [st_before["fc%d.bias" % i] for i in range(nfc)] if step else None,
                      e_wide=e_wide, tag="step %d: " % step)
        worst["P"] = bound(gm.p(B), om.p(), c64["P"], "P (step %d)" % step, floor=c64["e_P"])
        worst["loss"] = bound(loss_g, loss_o, c64["loss"], "loss (step %d)" % step, floor=c64["e_loss"])
        gm.backward()
        # ---- every FC contraction against float64 on ITS OWN inputs: 1e-5 relative + f32 roundoff floor
        layerwise_f64(gm, kv_before, E, Y, F, D, X, fc, wide)
        # ---- end to end: the same quantities after 2*nfc chained f32 GEMMs, against the float64 chain, with the oracle's
        # own distance to it as the yardstick (the two float32 chains sum in different orders)
        for li in range(nfc):
            d_o = om.delta(2 + li)
            if li == 0:
                d_o = d_o[:, :F * D] * (om.act(0) > 0)       # our dx is already relu'-masked, embedding columns only
            worst["delta"] = bound(gm.delta(2 + li), d_o, c64["delta"][li], "delta into fc%d (step %d)" % (li, step), floor=c64["e_delta"][li])
        for li in range(nfc):
            worst["dW"] = bound(gm.fc_grad(li), om.grad("fc%d.weights" % li), c64["dW"][li].reshape(-1), "dW%d (step %d)" % (li, step),
                                floor=c64["e_dW"][li].reshape(-1))
            worst["db"] = bound(gm.fc_grad(li, True), om.grad("fc%d.bias" % li), c64["db"][li], "db%d (step %d)" % (li, step), floor=c64["e_db"][li])
        # ---- per-key embedding gradient: BIT-EXACT against the oracle's reduction of OUR delta
        dx = gm.delta(2)
        g_gpu = []
        for f in range(F):
            ids, g = gm.emb_grads(f)
            np.testing.assert_array_equal(ids, uniq[f])
            for i, idv in enumerate(ids):
                ks = np.nonzero(E[:, f] == idv)[0]
                gk = dx[ks, f * D:(f + 1) * D]
                np.testing.assert_array_equal(g[i], orc.emb_geff(gk, orc.GRAD_COMPAT, 0), err_msg="emF%d.%d" % (f, idv))
            ids64, g64 = c64["geff"][f]
            np.testing.assert_array_equal(ids, ids64)
            worst["g_eff"] = bound(g, np.stack([om.grad(orc.emb_key(f, float(idv))) for idv in ids]), g64, "per-key gradients of field %d (step %d)" % (f, step), floor=c64["e_geff"][f])
            g_gpu.append(g)
        gm.update()
        om.apply_update()
        # ---- Adam on rows: BIT-EXACT given our gradient
        for f in range(F):
            w1 = kv.get_rows(f, uniq[f]); m1 = kv.get_rows(f, uniq[f], 1); v1 = kv.get_rows(f, uniq[f], 2)
            for i in range(len(uniq[f])):
                we, me, ve = orc.adam_update(w0[f][i], g_gpu[f][i], m0[f][i], v0[f][i])
                np.testing.assert_array_equal(w1[i], we); np.testing.assert_array_equal(m1[i], me); np.testing.assert_array_equal(v1[i], ve)
        # ---- parameters after the step: against the float64 chain's, the oracle's own drift as the yardstick.  (One Adam
        # step moves a weight by ~alfa*g/(|g|+eps): where |g| ~ eps the quotient is ill-conditioned for BOTH float32
        # chains -- which is exactly what max|oracle - f64| measures.)
        for f in range(F):
            w1 = kv.get_rows(f, uniq[f])
            wo = np.stack([st.get(orc.emb_key(f, float(i))) for i in uniq[f]])
            w64 = np.stack([ch.rows[f][int(i)][0] for i in uniq[f]])
            worst["rows"] = bound(w1, wo, w64, "embedding rows of field %d after step %d" % (f, step), floor=ch.floor_rows(f, uniq[f]))
        for li in range(nfc):
            worst["W"] = bound(kv.get("fc%d.weights" % li), st.get("fc%d.weights" % li), ch.W[li].reshape(-1), "fc%d.weights after step %d" % (li, step),
                               floor=ch.floor_W(li).reshape(-1))
            worst["b"] = bound(kv.get("fc%d.bias" % li), st.get("fc%d.bias" % li), ch.b[li], "fc%d.bias after step %d" % (li, step), floor=ch.floor_b(li))
        if wide:
            touched = np.unique(E % 97)
            wo = np.array([st.get(orc.wide_key(float(i)))[0] for i in touched], f32)
            worst["wide"] = bound(kv.get_wide(touched), wo, ch.ww[touched], "wide weights after step %d" % step, floor=ch.floor_wide(touched))
            worst["wide.bias"] = bound(kv.get("wide.bias"), st.get("wide.bias"), ch.wb, "wide.bias after step %d" % step, floor=ch.floor_wide_bias())
    # (max |gpu - f64|, max |oracle - f64|) of the last step, per kind of quantity: on the record
    import json, os
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/e2e_f64_errors.jsonl", "a") as fjs:
            fjs.write(json.dumps({"case": [int(wide), F, D, X, fc, V, B, int(zipf)], "gpu_vs_f64__oracle_vs_f64": worst}) + "\n")
    except OSError:
        pass
    gm.close(); kv.close()


@pytest.mark.parametrize("F,D,X,fc,V,B", [(4, 8, 3, [16, 8, 1], 30, 64), (23, 10, 45, [150, 10, 1], 50, 100),
                                          (26, 16, 13, [512, 256, 1], 2000, 1024)])
def test_fused_train_equals_split_form(orc, F, D, X, fc, V, B):
    """ps_model_train (fused updaters, three streams) == forward/backward/update (split form), bit for bit.
    Also the regression test of a stream-ordering bug: the fused dense update (side stream) once could overwrite
    W while the main chain's last delta GEMM was still reading it -- shape/timing dependent, hence several shapes."""
    import ps_amd
    rng = np.random.default_rng(3)
    res = []
    for fused in (True, False):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, 50, store=kv, max_batch=B)
        r2 = np.random.default_rng(3)
        for _ in range(6):
            E, Xd, Y = data(r2, B, F, X, V, True)
            d = {"E": E, "X": Xd, "Y": Y, "W": E % 50}
            if fused:
                gm.train(d)
            else:
                gm.forward(d); gm.backward(); gm.update()
        res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)],
                    kv.get_wide(np.arange(50)), kv.get("wide.bias"), kv.global_step()))
        gm.close(); kv.close()
    a, b = res
    for x, y in zip(a[0], b[0]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(a[1], b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2]); np.testing.assert_array_equal(a[3], b[3])
    assert a[4] == b[4] == 6


@pytest.mark.gpu
def test_fused_step_keeps_its_gradients_only_when_asked():
    """ps_model_set_keep_grads (ps_native.h): the fused step consumes every key's gradient in registers (KVStore.sum's map does not
    outlive update either, store/KVStore.java:268-276); emb_grads() after train() then fails loudly.  A model asked to keep them
    hands out the same gradients the split form does, and the rows it trains are bit-identical either way."""
    import ps_amd
    F, D, X, fc, V, B = 4, 8, 3, [16, 1], 60, 96
    out = []
    for keep in (False, True):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, keep_grads=keep)
        r2 = np.random.default_rng(11)
        E, Xd, Y = data(r2, B, F, X, V, False)
        d = {"E": E, "X": Xd, "Y": Y}
        gm.forward(d); gm.backward()
        split = [gm.emb_grads(f) for f in range(F)]                  # the split form always has them
        gm.train(d)
        if keep:
            for f in range(F):
                ids, g = gm.emb_grads(f)
                np.testing.assert_array_equal(ids, split[f][0]); np.testing.assert_array_equal(g, split[f][1])
        else:
            with pytest.raises(ps_amd.native.PsError, match="keep_grads"):
                gm.emb_grads(0)
        gm.train(d)
        out.append([kv.get_rows(f, np.arange(V)) for f in range(F)])
        gm.close(); kv.close()
    for x, y in zip(*out):
        np.testing.assert_array_equal(x, y)


def test_ftrl_rows_bit_exact(orc):
    """Ftrl fused into the sparse scatter (config 5's updater) is bit-exact with the oracle,
    including the dw[0]==0 skip and w lagging z,n by one update."""
    import ps_amd
    F, D, X, fc, V, B = 2, 8, 1, [8, 1], 12, 32
    rng = np.random.default_rng(9)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    kv.set_updater("emF", ps_amd.FtrlUpdater(0.005, 1.0, 0.001, 0.001))
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, keep_grads=True)
    for step in range(4):
        E, Xd, Y = data(rng, B, F, X, V)
        uniq = [np.unique(E[:, f]) for f in range(F)]
        w0 = [kv.get_rows(f, uniq[f]) for f in range(F)]
        z0 = [kv.get_rows(f, uniq[f], 1) for f in range(F)]
        n0 = [kv.get_rows(f, uniq[f], 2) for f in range(F)]
        gm.train({"E": E, "X": Xd, "Y": Y})
        for f in range(F):
            ids, g = gm.emb_grads(f)
            w1 = kv.get_rows(f, ids); z1 = kv.get_rows(f, ids, 1); n1 = kv.get_rows(f, ids, 2)
            for i in range(len(ids)):
                we, ze, ne, _ = orc.ftrl_update(w0[f][i], g[i], z0[f][i], n0[f][i])
                np.testing.assert_array_equal(w1[i], we); np.testing.assert_array_equal(z1[i], ze); np.testing.assert_array_equal(n1[i], ne)
    gm.close(); kv.close()


def _expect_row(orc, kind, w, g, s1, s2):
    if kind == "ftrl":
        return orc.ftrl_update(w, g, s1, s2)[:3]
    if kind == "adam2":
        return orc.adam_update(w, g, s1, s2, alfa=0.02, beta1=0.8)
    if kind == "simple":
        return (np.asarray(g, f32) * f32(-0.05) + np.asarray(w, f32)).astype(f32), s1, s2
    return orc.adam_update(w, g, s1, s2)


@pytest.mark.parametrize("form", ["fused", "split", "keyed"])
def test_per_field_updaters_bit_exact(orc, form):
    """KVStore.update(Map) picks the updater per KEY: exact key, then a map key that is a prefix, then "default"
    (store/KVStore.java:240-252).  "emF1." -> Ftrl, "emF2" -> another Adam, "emF3." -> Simple, fields 0 and 4 fall through
    to "default": every field's rows move by ITS updater, bit for bit -- in the fused step (short keys and the long-key
    role: V is small, so the hottest key's run is far above 16), the split form (k_rows_apply) and the keyed push."""
    import ps_amd
    F, D, X, fc, V, B = 5, 8, 1, [8, 1], 7, 192
    kinds = ["adam", "ftrl", "adam2", "simple", "adam"]
    rng = np.random.default_rng(21)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    kv.set_updater("emF1.", ps_amd.FtrlUpdater(0.005, 1.0, 0.001, 0.001))
    kv.set_updater("emF2", ps_amd.AdamUpdater(0.02, 0.8))
    kv.set_updater("emF3.", ps_amd.SimpleUpdater(0.05))
    # ... and two updater keys that ARE a row's key: the exact match KVStore.update(Map) tries first (store/KVStore.java:242)
    kv.set_updater("emF0.2.0", ps_amd.FtrlUpdater(0.005, 1.0, 0.001, 0.001))      # row 2 of a "default" (Adam) field
    kv.set_updater("emF3.5.0", ps_amd.AdamUpdater(0.02, 0.8))                      # row 5 of the Simple field
    row_kind = {(0, 2): "ftrl", (3, 5): "adam2"}
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, keep_grads=True)
    for step in range(3):
        E, Xd, Y = data(rng, B, F, X, V)
        uniq = [np.unique(E[:, f]) for f in range(F)]
        before = [[kv.get_rows(f, uniq[f], k) for k in range(3)] for f in range(F)]
        if form == "fused":
            gm.train({"E": E, "X": Xd, "Y": Y})
            grads = [gm.emb_grads(f) for f in range(F)]
        elif form == "split":
            gm.forward({"E": E, "X": Xd, "Y": Y}); gm.backward()
            grads = [gm.emb_grads(f) for f in range(F)]
            gm.update()
        else:
            grads = [(uniq[f], rng.standard_normal((len(uniq[f]), D)).astype(f32)) for f in range(F)]
            kv.push_update([("emF%d.%d" % (f, int(i)), grads[f][1][k]) for f in range(F) for k, i in enumerate(grads[f][0])])
        for f in range(F):
            ids, g = grads[f]
            np.testing.assert_array_equal(ids, uniq[f])
            after = [kv.get_rows(f, ids, k) for k in range(3)]
            for i in range(len(ids)):
                exp = _expect_row(orc, row_kind.get((f, int(ids[i])), kinds[f]), before[f][0][i], g[i], before[f][1][i], before[f][2][i])
                for k in range(3):
                    np.testing.assert_array_equal(after[k][i], np.asarray(exp[k], f32).ravel(), err_msg="field %d id %d slot %d step %d" % (f, ids[i], k, step))
    gm.close(); kv.close()


def test_single_row_updater_keys_are_refused(orc):
    """an updater map key that ENDS INSIDE a row's id ("emF1.3": by String.startsWith a prefix of emF1.3.0, emF1.30.0, emF1.31.0 ...)
    is neither a field prefix nor a row's key: said, not ignored.  (A row's full key, "emF1.3.0", is honoured:
    test_per_field_updaters_bit_exact.)"""
    import ps_amd
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([6, 6], 4)
    kv.set_updater("emF1.3", ps_amd.FtrlUpdater())
    gm = ps_amd.DNN.buildModel(2, 4, 1, [4, 1], store=kv, max_batch=8)
    with pytest.raises(ps_amd.native.PsError) as ei:
        gm.train({"E": np.zeros((8, 2), np.int64), "X": np.zeros((8, 1), f32), "Y": np.ones(8, f32)})
    assert ei.value.code == ps_amd.native.PS_E_UNSUPPORTED
    gm.close(); kv.close()


def test_more_updater_groups_than_the_kernels_carry_are_refused(orc):
    """nine distinct updaters over the fields of one table group (the kernels carry eight, PS_EMB_UPD_GROUPS): PS_E_UNSUPPORTED,
    nothing is updated"""
    import ps_amd
    F = 10
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([6] * F, 4)
    for f, alfa in enumerate((0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07, 0.08)):
        kv.set_updater("emF%d." % f, ps_amd.AdamUpdater(alfa))          # + "default" for fields 8, 9 = the ninth
    gm = ps_amd.DNN.buildModel(F, 4, 1, [4, 1], store=kv, max_batch=8)
    before = kv.get_rows(0, np.arange(6)).copy()
    with pytest.raises(ps_amd.native.PsError) as ei:
        gm.train({"E": np.zeros((8, F), np.int64), "X": np.zeros((8, 1), f32), "Y": np.ones(8, f32)})
    assert ei.value.code == ps_amd.native.PS_E_UNSUPPORTED
    np.testing.assert_array_equal(kv.get_rows(0, np.arange(6)), before)
    gm.close(); kv.close()


def test_loss_slim_stops_backward(orc):
    """model/DNN.java:58-63: loss <= 0.01 (or NaN) returns before backward: nothing is updated."""
    import ps_amd
    F, D, X, fc, V, B = 2, 4, 1, [4, 1], 6, 8
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
    # force P -> 0.999 for label 1: huge positive bias on the last layer
    kv.put("fc1.bias", np.array([50.0], f32))
    E = np.zeros((B, F), np.int64); Xd = np.zeros((B, X), f32); Y = np.ones(B, f32)
    before = kv.get_rows(0, [0]).copy(); wb = kv.get("fc0.weights").copy()
    loss = gm.train({"E": E, "X": Xd, "Y": Y})
    assert loss <= 0.01
    np.testing.assert_array_equal(kv.get_rows(0, [0]), before)
    np.testing.assert_array_equal(kv.get("fc0.weights"), wb)
    gm.close(); kv.close()


def test_predict_matches_forward(orc):
    import ps_amd
    F, D, X, fc, V, B = 3, 4, 2, [6, 1], 9, 16
    rng = np.random.default_rng(1)
    st, om, kv, gm = make_pair(orc, False, F, D, X, fc, V, B)
    E, Xd, Y = data(rng, B, F, X, V)
    p_g = gm.predict({"E": E, "X": Xd})
    dims = [F * D + X] + list(fc)
    A = [gm.act(1)] + [gm.act(2 + l) for l in range(len(fc))]
    fl = forward_floors(A[:len(fc)], [kv.get("fc%d.weights" % l).reshape(dims[l], dims[l + 1]) for l in range(len(fc))], [kv.get("fc%d.bias" % l) for l in range(len(fc))])
    close(p_g, om.predict(E.astype(f32), Xd), 0.25 * fl[-1].reshape(-1) + 4 * EPS, what="predict")
    gm.close(); kv.close()


@pytest.mark.parametrize("per_field", [False, True])
def test_sharded_path_n1_equals_fused_step(orc, per_field):
    """The PS exchange with one shard (every collective the identity) is the fused step, bit for bit:
    pull -> train on the cached rows -> push -> owner mean over 1 worker -> updater; dense/wide g/1.
    Also with two plan contexts (two models on one store) and step t+1 prepared before step t finishes.
    per_field: "emF1." -> Ftrl, "emF3" -> another Adam (the owner-side push resolves the updater per field like the fused step)."""
    import ps_amd
    from ps_amd.sharded import HipBackend, LocalComm, ShardedWorker
    F, D, X, fc, V, B, WS = 5, 8, 3, [16, 8, 1], 40, 96, 31
    res = []
    for nctx in (0, 1, 2, 3):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        if per_field:
            kv.set_updater("emF1.", ps_amd.FtrlUpdater(0.005, 1.0, 0.001, 0.001))
            kv.set_updater("emF3", ps_amd.AdamUpdater(0.02, 0.8))
        gms = [ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B) for _ in range(max(nctx, 1))]
        worker = ShardedWorker(HipBackend(gms), LocalComm())
        rng = np.random.default_rng(4)
        bs = []
        for _ in range(5):
            E, Xd, Y = data(rng, B, F, X, V, True)
            bs.append(ps_amd.Batch(E, Xd, Y, E % WS))
        if nctx == 0:
            losses = [gms[0].train(b) for b in bs]
        elif nctx == 1:
            losses = [worker.step(b) for b in bs]
        elif nctx == 3:
            losses = res[0][0][:-1] + [worker.run(bs, len(bs), want_loss=True)]      # the 3-stage pipeline; last loss
        else:
            losses = []
            p = worker.prepare(bs[0])
            for i in range(len(bs)):
                nxt = worker.prepare(bs[i + 1]) if i + 1 < len(bs) else None     # planned BEFORE step i runs
                losses.append(worker.finish(p))
                p = nxt
        res.append((losses, [kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get_rows(f, np.arange(V), 1) for f in range(F)],
                    [kv.get("fc%d.weights" % i) for i in range(3)], [kv.get("fc%d.bias" % i) for i in range(3)],
                    kv.get_wide(np.arange(WS)), kv.get("wide.bias"), kv.global_step()))
        for g in gms:
            g.close()
        kv.close()
    a = res[0]
    for b in res[1:]:
        assert a[0] == b[0]
        for i in (1, 2, 3, 4):
            for x, y in zip(a[i], b[i]):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a[5], b[5]); np.testing.assert_array_equal(a[6], b[6])
        assert a[7] == b[7]


def test_multi_hot_bags(orc):
    """Variable-length bags (config 5's shape: CSR offsets over (sample, field), sum pooling), incl. empty
    bags and a key repeated inside one bag.  The reference is single-hot, so the oracle here is its arithmetic
    applied to the bag semantics: forward = relu(sequential f32 sum of the bag's rows); every id of a bag
    receives that bag's relu'-masked delta; per key the entries reduce in entry order with the reference's
    double-backward factor; Adam as written.  All bit-exact."""
    import ps_amd
    F, D, X, fc, V, B = 3, 8, 2, [8, 1], 25, 48
    rng = np.random.default_rng(21)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, max_nnz=B * F * 12)
    for step in range(3):
        lens = rng.integers(0, 9, size=B * F)
        lens[0] = 0; lens[5] = 40                         # an empty bag and one longer than the 32-entry chunk
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
        ids[offsets[5]:offsets[5] + 40] = ids[offsets[5]]   # one key 40 times in one bag -> run > 32
        Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.4).astype(f32)
        tabs = [kv.get_rows(f, np.arange(V)) for f in range(F)]
        m0 = [kv.get_rows(f, np.arange(V), 1) for f in range(F)]
        v0 = [kv.get_rows(f, np.arange(V), 2) for f in range(F)]
        gm.forward({"E": ids, "X": Xd, "Y": Y, "offsets": offsets})
        A = gm.act(0)
        for bag in range(B * F):
            b, f = divmod(bag, F)
            s = np.zeros(D, f32)
            for k, p in enumerate(range(offsets[bag], offsets[bag + 1])):
                s = tabs[f][ids[p]].copy() if k == 0 else (tabs[f][ids[p]] + s).astype(f32)
            np.testing.assert_array_equal(A[b, f * D:(f + 1) * D], np.maximum(s, 0), err_msg="bag %d" % bag)
        gm.backward()
        dx = gm.delta(2)
        field_of = np.repeat(np.arange(B * F) % F, lens)
        samp_of = np.repeat(np.arange(B * F) // F, lens)
        g_gpu = {}
        for f in range(F):
            gi, gg = gm.emb_grads(f)
            sel = np.nonzero(field_of == f)[0]
            np.testing.assert_array_equal(gi, np.unique(ids[sel]))
            for i, idv in enumerate(gi):
                ps = sel[ids[sel] == idv]                    # entries of this key, in entry (bag-major) order
                gk = np.stack([dx[samp_of[p], f * D:(f + 1) * D] for p in ps])
                np.testing.assert_array_equal(gg[i], orc.emb_geff(gk, orc.GRAD_COMPAT, 32), err_msg="emF%d.%d n=%d" % (f, idv, len(ps)))
                g_gpu[(f, idv)] = gg[i]
        gm.update()
        for (f, idv), g in g_gpu.items():
            we, me, ve = orc.adam_update(tabs[f][idv], g, m0[f][idv], v0[f][idv])
            np.testing.assert_array_equal(kv.get_rows(f, [idv])[0], we)
            np.testing.assert_array_equal(kv.get_rows(f, [idv], 1)[0], me)
    # a bag of exactly one id per (sample, field) is the single-hot path, bit for bit
    E = rng.integers(0, V, size=(B, F)).astype(np.int64)
    Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.4).astype(f32)
    l1 = gm.forward({"E": E, "X": Xd, "Y": Y}); a1 = gm.act(0).copy()
    l2 = gm.forward({"E": E.ravel(), "X": Xd, "Y": Y, "offsets": np.arange(B * F + 1, dtype=np.int64)})
    assert l1 == l2
    np.testing.assert_array_equal(a1, gm.act(0))
    gm.close(); kv.close()


def test_out_of_range_id_is_reported(orc):
    """An id outside its table is Resp 204 (PS_MISSING), reported after the step; nothing crashes."""
    import ps_amd
    F, D, X, fc, V, B = 2, 4, 1, [4, 1], 10, 8
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
    E = np.zeros((B, F), np.int64); E[3, 1] = V + 7
    with pytest.raises(ps_amd.native.PsError) as e:
        gm.train({"E": E, "X": np.zeros((B, X), f32), "Y": np.ones(B, f32)})
    assert e.value.code == ps_amd.native.PS_MISSING
    assert kv.get("emF0.%d.0" % (V + 1)) is None
    with pytest.raises(ps_amd.native.PsError):
        gm.train({"E": np.zeros((B + 1, F), np.int64), "X": np.zeros((B + 1, X), f32), "Y": np.ones(B + 1, f32)})   # B > max_batch
    gm.close(); kv.close()


@pytest.mark.parametrize("name", ["dnn.npz", "widedeep.npz"])
def test_against_committed_golden(name):
    """HIP path against tests/golden/*.npz (fixed numbers; the oracle is not executed here).
    Copies / init are bit-exact; GEMM-fed values within 1e-5 relative of the stored ones."""
    import os
    import ps_amd
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    m = [int(x) for x in z["meta"]]
    F, D, X, B, V, WS, wide, fc = m[0], m[1], m[2], m[3], m[4], m[5], bool(m[6]), m[7:]
    kv = ps_amd.KVStore(0, int(z["seed"][0]))
    kv.create_embedding([V] * F, D)
    gm = (ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B) if wide
          else ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B))
    for f in range(F):
        np.testing.assert_array_equal(kv.get_rows(f, np.arange(V)), z["init_emb"][f])
    for l in range(len(fc)):
        np.testing.assert_array_equal(kv.get("fc%d.weights" % l), z["init_fc%d_w" % l])
        np.testing.assert_array_equal(kv.get("fc%d.bias" % l), z["init_fc%d_b" % l])
    ch = Chain(wide, F, D, X, fc, WS)              # the float64 chain from the stored initial parameters (tests/f64_chain.py)
    ch.load_fc([z["init_fc%d_w" % l] for l in range(len(fc))], [z["init_fc%d_b" % l] for l in range(len(fc))])
    for s in range(2):
        p = "s%d_" % s
        E = z[p + "E"]
        loss = gm.forward({"E": E, "X": z[p + "X"], "Y": z[p + "Y"], "W": (E % WS) if wide else None})
        if s:
            ch.anchor([kv.get("fc%d.weights" % l) for l in range(len(fc))], [kv.get("fc%d.bias" % l) for l in range(len(fc))],
                      lambda f, ids: kv.get_rows(f, ids), kv.get_wide(np.arange(WS)) if wide else None, kv.get("wide.bias") if wide else None)
        c64 = ch.step(E, z[p + "X"], z[p + "Y"], (E % WS) if wide else None, lambda f, ids: z["init_emb"][f][ids])
        if s == 0:
            np.testing.assert_array_equal(gm.act(0), z[p + "embA"])
            np.testing.assert_array_equal(gm.act(1), z[p + "concatA"])
        nfc, dims = len(fc), [F * D + X] + list(fc)
        ref_act = lambda layer: z[p + "concatA"] if layer == 1 else z[p + "fc%d_A" % (layer - 2)]      # noqa: E731
        prev = "init_" if s == 0 else "s%d_" % (s - 1)
        e_wide = 0.0
        if wide and s:
            e_wide = (np.abs(kv.get_wide(np.arange(WS)).astype(np.float64) - z[prev + "wide_w"])[E % WS].sum(axis=1)
                      + abs(float(kv.get("wide.bias")[0]) - float(np.ravel(z[prev + "wide_bias"])[0])))
        check_forward(gm, ref_act, z[p + "P"], z[p + "loss"][0], loss, z[p + "Y"], nfc, dims, wide,
                      [kv.get("fc%d.weights" % l) for l in range(nfc)], [kv.get("fc%d.bias" % l) for l in range(nfc)],
                      [z[prev + "fc%d_w" % l] for l in range(nfc)] if s else None, [z[prev + "fc%d_b" % l] for l in range(nfc)] if s else None,
                      e_wide=e_wide, tag="golden step %d: " % s)
        gm.backward()
        # end to end: against the float64 chain, the stored (restatement) values' own distance to it as the yardstick
        for l in range(len(fc)):
            bound(gm.fc_grad(l), z[p + "fc%d_dW" % l], c64["dW"][l].reshape(-1), "dW%d" % l, floor=c64["e_dW"][l].reshape(-1))
            bound(gm.fc_grad(l, True), z[p + "fc%d_db" % l], c64["db"][l], "db%d" % l, floor=c64["e_db"][l])
        for f in range(F):
            ids, g = gm.emb_grads(f)
            np.testing.assert_array_equal(ids, np.nonzero(z[p + "emb_touched"][f])[0])
            np.testing.assert_array_equal(ids, c64["geff"][f][0])
            bound(g, z[p + "emb_grad"][f][ids], c64["geff"][f][1], "emb grad f%d" % f, floor=c64["e_geff"][f])
        gm.update()
        for f in range(F):
            have = np.nonzero(z[p + "emb_have"][f])[0]
            w64 = np.stack([ch.rows[f][int(i)][0] if int(i) in ch.rows[f] else z["init_emb"][f][i].astype(np.float64) for i in have])
            fl = np.stack([ch.floor_rows(f, [i])[0] if int(i) in ch.rows[f] else np.zeros(D) for i in have])      # (rows never trained: initial values, no floor)
            bound(kv.get_rows(f, have), z[p + "emb_W"][f][have], w64, "rows of field %d after step %d" % (f, s), floor=fl)
        for l in range(len(fc)):
            bound(kv.get("fc%d.weights" % l), z[p + "fc%d_w" % l], ch.W[l].reshape(-1), "fc%d.weights after step %d" % (l, s), floor=ch.floor_W(l).reshape(-1))
            bound(kv.get("fc%d.bias" % l), z[p + "fc%d_b" % l], ch.b[l], "fc%d.bias after step %d" % (l, s), floor=ch.floor_b(l))
        if wide:
            bound(kv.get_wide(np.arange(WS)), z[p + "wide_w"], ch.ww, "wide weights after step %d" % s, floor=ch.floor_wide())
            bound(kv.get("wide.bias"), z[p + "wide_bias"], ch.wb, "wide.bias after step %d" % s, floor=ch.floor_wide_bias())
    gm.close(); kv.close()


@pytest.mark.parametrize("is_async,per_field", [(0, False), (1, False), (0, True), (1, True)])
def test_owner_push_from_many_workers(orc, is_async, per_field):
    """PServer.push + psUpdate on one owner receiving the lists of 5 workers (net/PServer.java:164-214):
    the sort-free path (worker-grouped lists) == the stable-sort path == the oracle's arithmetic
    (BSP: mean over the pushing workers in worker order; async: one Adam step per push in worker order).
    per_field: the rows of field 1 are under "emF1." -> Ftrl, the others under "default" (store/KVStore.java:240-252)."""
    import ctypes as C
    import ps_amd
    from ps_amd import native as N
    F, D, V, NW = 3, 8, 60, 5
    rng = np.random.default_rng(17 + is_async)
    R = F * V
    lists = []
    for w in range(NW):
        k = int(rng.integers(0, 70))
        rows = np.sort(rng.choice(R, size=k, replace=False)).astype(np.uint32)     # unique inside one worker's list
        lists.append((rows, rng.standard_normal((k, D)).astype(f32)))
    lists[2] = (np.zeros(0, np.uint32), np.zeros((0, D), f32))                      # a worker that pushed nothing
    rows = np.concatenate([l[0] for l in lists]); grads = np.concatenate([l[1] for l in lists])
    counts = [len(l[0]) for l in lists]
    n = len(rows)
    out = []
    for grouped in (True, False):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        if per_field:
            kv.set_updater("emF1.", ps_amd.FtrlUpdater(0.005, 1.0, 0.001, 0.001))
        w0 = np.concatenate([kv.get_rows(f, np.arange(V)) for f in range(F)])
        L = N.lib()
        dr, dg = C.c_void_p(), C.c_void_p()
        N.check(L.ps_dev_alloc(kv.h, max(rows.nbytes, 4), C.byref(dr))); N.check(L.ps_dev_alloc(kv.h, max(grads.nbytes, 4), C.byref(dg)))
        N.check(L.ps_dev_upload(kv.h, dr, rows.ctypes.data, rows.nbytes)); N.check(L.ps_dev_upload(kv.h, dg, grads.ctypes.data, grads.nbytes))
        for rep in range(2):          # twice: the second pass sees the state the first left (mask cleared, Adam moments)
            pc = (C.c_int64 * NW)(*counts)
            N.check(L.ps_shard_apply_push(kv.h, dr, dg, n, pc if grouped else None, NW if grouped else 0, is_async))
        kv.sync()
        out.append([np.concatenate([kv.get_rows(f, np.arange(V), which) for f in range(F)]) for which in (0, 1, 2)])
        assert kv.global_step() == 2
        L.ps_dev_free(kv.h, dr); L.ps_dev_free(kv.h, dg)
        kv.close()
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)
    # the oracle's arithmetic
    W, M, Vv = w0.copy(), np.zeros_like(w0), np.zeros_like(w0)

    def upd(r, g):
        if per_field and V <= r < 2 * V:
            W[r], M[r], Vv[r] = orc.ftrl_update(W[r], g, M[r], Vv[r])[:3]
        else:
            W[r], M[r], Vv[r] = orc.adam_update(W[r], g, M[r], Vv[r])
    for rep in range(2):
        for r in np.unique(rows):
            gs = grads[rows == r]                        # worker order
            if is_async:
                for g in gs:
                    upd(r, g)
            else:
                S = gs[0].copy()
                for g in gs[1:]:
                    S = (g + S).astype(f32)
                upd(r, (S / f32(len(gs))).astype(f32))
    np.testing.assert_array_equal(out[0][0], W); np.testing.assert_array_equal(out[0][1], M); np.testing.assert_array_equal(out[0][2], Vv)


def test_very_long_runs_two_level_order(orc):
    """A hot key with a run above 128 chunks (here 4500 and 9000 entries, multi-hot) is folded in two levels
    (32-entry chunks, then 32 chunks at a time): bit-exact against the oracle's same order, including the
    double-backward pass; shorter runs in the same batch keep the one-level / sequential order."""
    import ps_amd
    F, D, X, fc, V, B = 2, 8, 1, [8, 1], 12, 64
    rng = np.random.default_rng(33)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, max_nnz=B * F * 200)
    for step in range(2):
        lens = rng.integers(1, 6, size=B * F)
        lens[0::2] += 70                                   # field 0: every bag carries 70 copies of the hot key
        lens[1::2] += 141 * (np.arange(B) < 64)            # field 1: 141 copies -> 9024 entries of one key
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
        for bag in range(B * F):
            hot = 70 if bag % 2 == 0 else 141
            ids[offsets[bag]:offsets[bag] + hot] = 3 if bag % 2 == 0 else 7
        Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.4).astype(f32)
        gm.forward({"E": ids, "X": Xd, "Y": Y, "offsets": offsets})
        gm.backward()
        dx = gm.delta(2)
        field_of = np.repeat(np.arange(B * F) % F, lens)
        samp_of = np.repeat(np.arange(B * F) // F, lens)
        longest = 0
        for f in range(F):
            gi, gg = gm.emb_grads(f)
            sel = np.nonzero(field_of == f)[0]
            np.testing.assert_array_equal(gi, np.unique(ids[sel]))
            for i, idv in enumerate(gi):
                ps = sel[ids[sel] == idv]
                gk = dx[samp_of[ps], f * D:(f + 1) * D]
                longest = max(longest, len(ps))
                np.testing.assert_array_equal(gg[i], orc.emb_geff(gk, orc.GRAD_COMPAT, 32), err_msg="emF%d.%d n=%d" % (f, idv, len(ps)))
        assert longest > 128 * 32
        gm.update()
    gm.close(); kv.close()


def test_library_driven_step_n1_equals_fused_step():
    """ps_shard_step with a 1-rank communicator (every collective a device copy) == the fused step, bit for bit."""
    import ps_amd
    from ps_amd.sharded import NativeWorker
    F, D, X, fc, V, B, WS = 5, 8, 3, [16, 8, 1], 40, 96, 31
    res = []
    for native in (0, 1, 2):                   # fused | one call per step | begin(t+1) before finish(t), two models
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gms = [ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B) for _ in range(max(native, 1))]
        gm = gms[0]
        wk = NativeWorker(gms, 1, 0) if native else None
        rng = np.random.default_rng(4)
        bs = []
        for _ in range(5):
            E, Xd, Y = data(rng, B, F, X, V, True)
            bs.append(ps_amd.Batch(E, Xd, Y, E % WS))
        if native == 2:
            losses = res[0][0][:-1] + [wk.run(bs, len(bs), want_loss=True)]
        else:
            losses = [wk.step(b) if native else gm.train(b) for b in bs]
        res.append((losses, [kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)],
                    kv.get_wide(np.arange(WS)), kv.get("wide.bias"), kv.global_step()))
        if wk:
            wk.close()
        for g in gms:
            g.close()
        kv.close()
    a = res[0]
    for b in res[1:]:
        assert a[0] == b[0] and a[5] == b[5]
        for i in (1, 2):
            for x, y in zip(a[i], b[i]):
                np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a[3], b[3]); np.testing.assert_array_equal(a[4], b[4])


def test_mapped_peer_one_rank_equals_fused_step():
    """ps_tune_set("mapped_peer", 2) on a 1-rank table: the rows / gradient exchanges of ps_shard_step run as the mapped-peer launch
    (this rank's own part stored through the same kernel into its own cache / receive region, own flag raised and awaited) -- the mode
    bench.py's sharded_n1 leg times as `mapped_peer`.  120 pipelined steps leave the fused step's tables, bit for bit."""
    import ctypes as C
    import ps_amd
    from ps_amd import native as N
    from ps_amd.sharded import NativeWorker
    F, D, X, fc, V, B, WS = 5, 8, 3, [16, 8, 1], 400, 256, 31
    rng = np.random.default_rng(14)
    data_ = []
    for _ in range(6):
        E, Xd, Y = data(rng, B, F, X, V, True)
        data_.append((E, Xd, Y, E % WS))
    res = []
    for mapped in (0, 2):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
        bs = [ps_amd.DeviceBatch(kv, *d) for d in data_]
        info = [0] * 5
        if mapped:
            N.lib().ps_tune_set(b"mapped_peer", mapped)
            try:
                wk = NativeWorker([gm], 1, 0)
                wk.run(bs, 120)
            finally:
                N.lib().ps_tune_set(b"mapped_peer", 0)
            kv.sync()
            mp5 = (C.c_int64 * 5)()
            N.check(N.lib().ps_shard_mapped_info(gm.h, mp5))
            info = [int(x) for x in mp5]
            wk.close()
        else:
            for i in range(120):
                gm.train_async(bs[i % len(bs)])
            kv.sync()
        res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)],
                    kv.get_wide(np.arange(WS)), kv.get("wide.bias"), kv.global_step(), info, int(N.lib().ps_store_wait_timeouts(kv.h))))
        for b in bs:
            b.close()
        gm.close(); kv.close()
    a, b = res
    assert b[5][0] == 1 and (b[5][1] & 1) and b[5][2] == 120 and b[5][3] == 120, b[5]
    assert a[6] == 0 and b[6] == 0 and a[4] == b[4] == 120
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2]); np.testing.assert_array_equal(a[3], b[3])


def test_simple_updater_rows_and_dense():
    """update/SimpleUpdater.java:20-22 (w += g * -eta) on embedding rows and FC tensors: bit-exact given the gradients."""
    import ps_amd
    F, D, X, fc, V, B = 3, 8, 2, [8, 1], 20, 32
    rng = np.random.default_rng(12)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    upd = ps_amd.SimpleUpdater(0.05)
    kv.set_updater("default", upd)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
    eta = f32(0.05)
    for _ in range(2):
        E, Xd, Y = data(rng, B, F, X, V)
        w0 = [kv.get_rows(f, np.arange(V)) for f in range(F)]
        fw0 = [kv.get("fc%d.weights" % l) for l in range(2)]; fb0 = [kv.get("fc%d.bias" % l) for l in range(2)]
        gm.forward({"E": E, "X": Xd, "Y": Y}); gm.backward()
        grads = [gm.emb_grads(f) for f in range(F)]
        fg = [(gm.fc_grad(l), gm.fc_grad(l, True)) for l in range(2)]
        gm.update()
        for f in range(F):
            ids, g = grads[f]
            want = w0[f].copy()
            want[ids] = ((g * -eta).astype(f32) + w0[f][ids]).astype(f32)
            np.testing.assert_array_equal(kv.get_rows(f, np.arange(V)), want)
        for l in range(2):
            np.testing.assert_array_equal(kv.get("fc%d.weights" % l), ((fg[l][0] * -eta).astype(f32) + fw0[l]).astype(f32))
            np.testing.assert_array_equal(kv.get("fc%d.bias" % l), ((fg[l][1] * -eta).astype(f32) + fb0[l]).astype(f32))
    gm.close(); kv.close()


def test_intended_embedding_gradient_mode(orc):
    """emb_grad_mode = PS_GRAD_INTENDED (the evident intent, the mean S/n instead of the double-
    backward factor (n+1)/(2n^2)): per-key gradients bit-exact against the oracle's intended mode, incl. runs > 32."""
    import ps_amd
    from ps_amd import native as N
    F, D, X, fc, V, B = 3, 8, 2, [8, 1], 6, 128
    rng = np.random.default_rng(31)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, emb_grad_mode=N.PS_GRAD_INTENDED)
    E, Xd, Y = data(rng, B, F, X, V, True)                       # V = 6: every key has a run of ~20, some above 32
    gm.forward({"E": E, "X": Xd, "Y": Y}); gm.backward()
    dx = gm.delta(2)
    longest = 0
    for f in range(F):
        ids, g = gm.emb_grads(f)
        for i, idv in enumerate(ids):
            ks = np.nonzero(E[:, f] == idv)[0]
            longest = max(longest, len(ks))
            np.testing.assert_array_equal(g[i], orc.emb_geff(dx[ks, f * D:(f + 1) * D], orc.GRAD_INTENDED, 0), err_msg="emF%d.%d n=%d" % (f, idv, len(ks)))
    assert longest > 32
    gm.close(); kv.close()


def test_intended_wide_gradient_mode_on_device(orc):
    """wide_grad_mode = intended (the oracle's switch, not a reference path): the gradient of a wide
    key is the sum of delta over its (sample, field) occurrences in THIS batch / B, Ftrl on those keys only -- bit-exact
    against orc.ftrl_update of that sum built from the GPU's own P; keys outside the batch keep their state."""
    import ps_amd
    from ps_amd import native as N
    F, D, X, fc, V, B, WS = 4, 8, 2, [16, 8, 1], 50, 64, 23
    rng = np.random.default_rng(12)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B, wide_grad_mode=N.PS_GRAD_INTENDED)
    allw = np.arange(WS)
    for step in range(3):
        E, Xd, Y = data(rng, B, F, X, V, True)
        Wd = (E * 7 + 3) % WS
        Wd[:, 0] = 5                                                   # one key in every sample
        w0, z0, n0 = kv.get_wide(allw), kv.get_wide(allw, 1), kv.get_wide(allw, 2)
        for fused in (step != 1,):
            if fused:
                gm.train({"E": E, "X": Xd, "Y": Y, "W": Wd})
            else:
                gm.forward({"E": E, "X": Xd, "Y": Y, "W": Wd}); gm.backward(); gm.update()
        p = gm.p(B)
        d = ((p - Y) / (p * (f32(1) - p))).astype(f32)
        d = (d * (p * (f32(1) - p))).astype(f32)                       # CrossEntropy' then Sigmoid' (f32, op by op)
        w1, z1, n1 = kv.get_wide(allw), kv.get_wide(allw, 1), kv.get_wide(allw, 2)
        for key in range(WS):
            occ = np.argwhere(Wd == key)                               # (sample, field) order
            if len(occ) == 0:
                assert w1[key] == w0[key] and z1[key] == z0[key] and n1[key] == n0[key]
                continue
            S = f32(0)
            for i, _ in occ:
                S = f32(d[i] + S)
            g = f32(S / f32(B))
            we, ze, ne, _ = orc.ftrl_update([w0[key]], [g], [z0[key]], [n0[key]])
            assert w1[key] == we[0] and z1[key] == ze[0] and n1[key] == ne[0], "wide.weights.%d" % key
    assert kv.get("wide.bias")[0] != 0          # "wide.bias" keeps its compat gradient rowMeans(delta) (Ftrl: w lags z by one step)
    gm.close(); kv.close()


def test_intended_wide_gradient_mode_in_the_sharded_step():
    """wide_grad_mode = intended through the key-sharded step (VERDICT r2 missing #5): the worker writes G[key] = its batch's
    sum / B and C[key] = 1 for the keys of its batch into the flat buffer, the owner side takes G / C (the mean over the workers
    that pushed the key) through Ftrl.  N = 1: equal to the fused intended step, bit for bit, for both drivers of the
    exchange, device and host batches; keys outside the batch keep their state."""
    import ps_amd
    from ps_amd import native as N
    from ps_amd.sharded import HipBackend, LocalComm, NativeWorker, ShardedWorker
    F, D, X, fc, V, B, WS = 5, 8, 3, [16, 8, 1], 40, 96, 211
    res = []
    for mode in ("fused", "python", "native", "native-dev"):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B, wide_grad_mode=N.PS_GRAD_INTENDED)
        rng = np.random.default_rng(4)
        raw = []
        for _ in range(5):
            E, Xd, Y = data(rng, B, F, X, V, True)
            Wd = (E * 7 + 3) % WS
            Wd[:, 0] = 5
            raw.append((E, Xd, Y, Wd))
        bs = [ps_amd.DeviceBatch(kv, *r) if mode == "native-dev" else ps_amd.Batch(*r) for r in raw]
        if mode == "fused":
            losses = [gm.train(b) for b in bs]
        elif mode == "python":
            wk = ShardedWorker(HipBackend([gm]), LocalComm())
            losses = [wk.step(b) for b in bs]
        else:
            wk = NativeWorker([gm], 1, 0)
            losses = [wk.step(b) for b in bs[:2]] + [None, None] + [wk.run(bs[2:], 3, want_loss=True)]
            wk.close()
        res.append((losses, kv.get_wide(np.arange(WS)), kv.get_wide(np.arange(WS), 1), kv.get_wide(np.arange(WS), 2), kv.get("wide.bias"),
                    [kv.get("fc%d.weights" % i) for i in range(3)], [kv.get_rows(f, np.arange(V)) for f in range(F)]))
        gm.close(); kv.close()
    a = res[0]
    used = np.unique(np.concatenate([r[3].ravel() for r in raw]))
    assert len(used) < WS and np.all(a[1][np.setdiff1d(np.arange(WS), used)] == 0)       # untouched keys: still zero
    assert np.abs(a[1][used]).max() > 0
    for b in res[1:]:
        assert [x for x in b[0] if x is not None][-1] == a[0][-1] and b[0][0] == a[0][0]
        for i in (1, 2, 3, 4):
            np.testing.assert_array_equal(a[i], b[i])
        for x, y in zip(a[5] + a[6], b[5] + b[6]):
            np.testing.assert_array_equal(x, y)
