"""BASELINE.json's configs at their FULL sizes, through the C ABI, against the oracle.

  configs[0]  CTR.java as shipped: 23 fields x D=10, 45 dense, FC[150,10,1], B=1000 (CTR.java:83-93),
              from libsvm text through the ingest path
  configs[1]  Wide&Deep 26 x 100k x 16, FC[512,256,1], B=4096, Zipf(1.05): one whole step vs the oracle
  configs[3]  one 1e9-row x 64 table (256 GB): sampled gather outputs vs the table's definition; and the
              fused-Adam variant on 320 M rows (W + M + V = 246 GB): sampled rows vs orc.adam_update
  configs[4]  multi-hot bags (~3.19 M ids), FTRL rows: determinism + sampled keys vs the oracle
(configs[2] needs 8 GPUs: covered by the N-ranks-on-one-GPU tests and the gloo tests at small sizes.)

Bit-exact wherever the arithmetic is a copy or a sequence of individually rounded f32 ops (gather, per-key
reduction, Adam, Ftrl); FP32 contractions within north_star's 1e-5 relative + the f32 roundoff floor."""
import ctypes as C

import numpy as np
import pytest

from f64_chain import Chain, bound
from test_gpu_parity import EPS, RTOL, check_forward, close, close64, layerwise_f64

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED
C2 = dict(F=26, V=100000, D=16, X=13, fc=[512, 256, 1], B=4096, wide=100000, zipf=1.05)


IDGENS = ["zipf_truncated", "zipf_clamped"]     # SURVEY 8d's law (inverse CDF over V) | rounds 1-2's clamped unbounded Zipf


def c2_batch(rng, cfg=C2, idgen="zipf_truncated"):
    from ps_amd import synth
    E = synth.draw_ids(rng, cfg["zipf"], cfg["V"], (cfg["B"], cfg["F"]), idgen)
    X = rng.standard_normal((cfg["B"], cfg["X"])).astype(f32)
    Y = (rng.random(cfg["B"]) < 0.25).astype(f32)
    return E, X, Y, E % cfg["wide"]


@pytest.mark.parametrize("idgen", IDGENS)
def test_config1_full_size_step_vs_oracle(orc, idgen):
    """configs[1] at B=4096 / V=100k / FC[512,256,1] / Zipf(1.05): forward, loss, every FC contraction, the per-key
    embedding gradient of ALL unique keys (~46k under the truncated law, hottest key ~480 entries; ~25k under the clamped
    generator, hot keys with n in the thousands) and the Adam / Ftrl updates."""
    import ps_amd
    cfg = C2
    F, D, X, fc, V, B, WS = cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["V"], cfg["B"], cfg["wide"]
    rng = np.random.default_rng(SEED)
    st = orc.Store(SEED)
    om = orc.Model(st, orc.WIDEDEEP, F, D, X, fc, wide_size=WS)
    om.set_grad_mode(orc.GRAD_COMPAT, orc.GRAD_COMPAT, 0)          # the reference's sequential order (single-hot)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
    E, Xd, Y, Wd = c2_batch(rng, idgen=idgen)
    uniq = [np.unique(E[:, f]) for f in range(F)]
    if idgen == "zipf_clamped":
        assert max(int((E[:, f] == V - 1).sum()) for f in range(F)) > 1000   # the unbounded tail piles up on one key per field
    else:
        hot = max(int((E[:, f] == 0).sum()) for f in range(F))               # rank 1: P = 1 / H(V, 1.05) = 0.107
        assert 350 < hot < 560 and sum(len(u) for u in uniq) > 40000
    w0 = [kv.get_rows(f, uniq[f]) for f in range(F)]
    m0 = [kv.get_rows(f, uniq[f], 1) for f in range(F)]
    v0 = [kv.get_rows(f, uniq[f], 2) for f in range(F)]
    kv_before = {"fc%d.%s" % (i, k): kv.get("fc%d.%s" % (i, k)) for i in range(3) for k in ("weights", "bias")}
    loss_o = om.train(E.astype(f32), Xd, Y, Wd.astype(f32), do_update=False)
    loss_g = gm.forward({"E": E, "X": Xd, "Y": Y, "W": Wd})
    # index gather (+ relu + concat): bit-exact
    np.testing.assert_array_equal(gm.act(0), om.act(0))
    np.testing.assert_array_equal(gm.act(1), om.act(1))
    # forward vs the oracle's own run (same parameters on both sides at this first step): 1e-5 relative + the propagated
    # roundoff floor of the chain's contractions (sum |terms|: test_gpu_parity.forward_floors), P and the loss through the head
    dims = [F * D + X] + list(fc)
    check_forward(gm, om.act, om.p(), loss_o, loss_g, Y, 3, dims, True,
                  [kv_before["fc%d.weights" % i] for i in range(3)], [kv_before["fc%d.bias" % i] for i in range(3)], tag="configs[1]: ")
    # the loss op itself: float64 on the GPU's own P.  Against the oracle's own chain only loosely: with random-init
    # weights of this width most logits sit on the sigmoid's clipped ends (p = 0.001 / 0.999), where
    # d(-ln p)/dp = 1/p = 1000 turns the f32 GEMM-order roundoff of P (<= 2e-5) into percent-level term differences
    pg = gm.p(B).astype(np.float64)
    loss64 = float(np.mean(-Y * np.log(pg) - (1 - Y) * np.log(1 - pg)))
    assert abs(loss_g - loss64) <= RTOL * loss64, (loss_g, loss64)
    # ... and against the float64 CHAIN of the same step from the same parameters (tests/f64_chain.py), the oracle's own
    # distance to it as the yardstick: |gpu - f64| <= 1e-5 |f64| + 4 |oracle - f64|
    ch = Chain(True, F, D, X, fc, WS)
    ch.load_fc([kv_before["fc%d.weights" % i] for i in range(3)], [kv_before["fc%d.bias" % i] for i in range(3)])
    c64 = ch.step(E, Xd, Y, Wd, lambda f, ids: w0[f][np.searchsorted(uniq[f], ids)], update=False)
    e_loss = bound(loss_g, loss_o, c64["loss"], "loss vs the float64 chain", floor=c64["e_loss"])
    e_p = bound(gm.p(B), om.p(), c64["P"], "P vs the float64 chain", floor=c64["e_P"])
    gm.backward()
    layerwise_f64(gm, kv_before, E, Y, F, D, X, fc, True)
    e_d = [bound(gm.delta(2 + li), om.delta(2 + li)[:, :F * D] * (om.act(0) > 0) if li == 0 else om.delta(2 + li), c64["delta"][li],
                 "delta into fc%d vs the float64 chain" % li, floor=c64["e_delta"][li]) for li in range(3)]
    e_w = [bound(gm.fc_grad(li), om.grad("fc%d.weights" % li), c64["dW"][li].reshape(-1), "dW%d vs the float64 chain" % li,
                 floor=c64["e_dW"][li].reshape(-1)) for li in range(3)]
    import json, os
    try:        # (max |gpu - f64|, max |oracle - f64|) on the record
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/e2e_f64_errors.jsonl", "a") as fjs:
            fjs.write(json.dumps({"case": "configs[1] full size, %s" % idgen, "loss": e_loss, "P": e_p, "delta": e_d, "dW": e_w}) + "\n")
    except OSError:
        pass
    # per-key gradient of every unique key: bit-exact against the oracle's reduction of the GPU's own delta
    dx = gm.delta(2)
    nkeys = 0
    g_gpu = []
    for f in range(F):
        ids, g = gm.emb_grads(f)
        np.testing.assert_array_equal(ids, uniq[f])
        order = np.argsort(E[:, f], kind="stable")
        bounds = np.searchsorted(E[order, f], ids)
        bounds = np.append(bounds, B)
        for i in range(len(ids)):
            ks = order[bounds[i]:bounds[i + 1]]                        # samples carrying the key, in batch order
            np.testing.assert_array_equal(g[i], orc.emb_geff(dx[ks, f * D:(f + 1) * D], orc.GRAD_COMPAT, 0),
                                          err_msg="emF%d.%d (n=%d)" % (f, ids[i], len(ks)))
        nkeys += len(ids)
        g_gpu.append(g)
    assert nkeys > 20000
    gm.update()
    om.apply_update()
    # Adam on every touched row: bit-exact given the GPU's gradient
    for f in range(F):
        w1 = kv.get_rows(f, uniq[f]); m1 = kv.get_rows(f, uniq[f], 1); v1 = kv.get_rows(f, uniq[f], 2)
        we, me, ve = orc.adam_update(w0[f].reshape(-1), g_gpu[f].reshape(-1), m0[f].reshape(-1), v0[f].reshape(-1))
        np.testing.assert_array_equal(w1.reshape(-1), we); np.testing.assert_array_equal(m1.reshape(-1), me)
        np.testing.assert_array_equal(v1.reshape(-1), ve)
    # rows no sample touched are untouched (lazy Adam)
    cold = np.setdiff1d(np.arange(0, V, 997), uniq[0])
    np.testing.assert_array_equal(kv.get_rows(0, cold), orc.init_rows(SEED, 0, cold, D, orc.xavier_scale(1, D)))
    # after the step: against the float64 chain's updated parameters, the oracle's own distance as the yardstick
    ch.step(E, Xd, Y, Wd, lambda f, ids: w0[f][np.searchsorted(uniq[f], ids)])       # (same step again, now with its updates)
    for f in range(0, F, 5):
        wo = np.stack([st.get(orc.emb_key(f, float(i))) for i in uniq[f][:400]])
        w64 = np.stack([ch.rows[f][int(i)][0] for i in uniq[f][:400]])
        bound(kv.get_rows(f, uniq[f][:400]), wo, w64, "rows of field %d after the step" % f, floor=ch.floor_rows(f, uniq[f][:400]))
    for li in range(3):
        bound(kv.get("fc%d.weights" % li), st.get("fc%d.weights" % li), ch.W[li].reshape(-1), "fc%d.weights after the step" % li, floor=ch.floor_W(li).reshape(-1))
    gm.close(); kv.close()


def ctr_text(rng, n, F=23, X=45, V=3000):
    lines = []
    for _ in range(n):
        ids = rng.integers(1, V, size=F)
        vals = np.round(rng.standard_normal(X), 4)
        lines.append("%d " % int(rng.random() < 0.3) + " ".join("%d:1" % i for i in ids) + " " +
                     " ".join("%d:%s" % (F + j + 1, repr(float(v))) for j, v in enumerate(vals)))
    return "\n".join(lines) + "\n"


def test_config0_ctr_shape_b1000_from_libsvm_text(orc, tmp_path):
    """configs[0]: CTR.java's DNN (23 x 10-dim, 45 dense, FC[150,10,1], Adam) at its own batch of 1000 -- two steps
    read from CTR-format libsvm text by the ingest pipeline, against the oracle's parser + model."""
    import ps_amd
    F, D, X, fc, V, B = 23, 10, 45, [150, 10, 1], 3000, 1000
    rng = np.random.default_rng(11)
    text = ctr_text(rng, 2 * B, F, X, V)
    path = tmp_path / "train.txt"
    path.write_text(text)
    st = orc.Store(SEED)
    om = orc.Model(st, orc.DNN, F, D, X, fc)
    om.set_grad_mode(orc.GRAD_COMPAT, orc.GRAD_COMPAT, 0)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
    ds = ps_amd.DataSet(kv, str(path), F, X, B, threads=2)
    Eo, Xo, Yo, _ = orc.parse_libsvm(text, F, X)
    ch = Chain(False, F, D, X, fc)                 # the float64 chain of the same two steps (tests/f64_chain.py)
    ch.load_fc([kv.get("fc%d.weights" % i) for i in range(3)], [kv.get("fc%d.bias" % i) for i in range(3)])
    for step in range(2):
        b = ds.next()
        assert b is not None
        sl = slice(step * B, (step + 1) * B)
        lo = om.train(Eo[sl].astype(f32), Xo[sl], Yo[sl], None, do_update=False)
        lg = gm.forward(b)
        if step:        # (the floors' input: how far the GPU's parameters are from the chain's at the start of this step)
            ch.anchor([kv.get("fc%d.weights" % i) for i in range(3)], [kv.get("fc%d.bias" % i) for i in range(3)], lambda f, ids: kv.get_rows(f, ids))
        c64 = ch.step(Eo[sl].astype(np.int64), Xo[sl], Yo[sl], None, lambda f, ids: kv.get_rows(f, ids))
        if step == 0:
            np.testing.assert_array_equal(gm.act(1), om.act(1))            # parser + gather + concat: bit-exact
        # end to end: |gpu - f64| <= 1e-5 |f64| + 4 |oracle - f64| (the oracle's own f32 chain as the yardstick)
        bound(lg, lo, c64["loss"], "loss step %d" % step, floor=c64["e_loss"])
        bound(gm.p(B), om.p(), c64["P"], "P step %d" % step, floor=c64["e_P"])
        kvb = {"fc%d.%s" % (i, k): kv.get("fc%d.%s" % (i, k)) for i in range(3) for k in ("weights", "bias")}
        gm.backward()
        layerwise_f64(gm, kvb, Eo[sl].astype(np.int64), Yo[sl], F, D, X, fc, False)   # every contraction: 1e-5 + f32 floor
        dx = gm.delta(2)
        E = Eo[sl].astype(np.int64)
        for f in (0, 11, 22):
            ids, g = gm.emb_grads(f)
            for i, idv in enumerate(ids):
                ks = np.nonzero(E[:, f] == idv)[0]
                np.testing.assert_array_equal(g[i], orc.emb_geff(dx[ks, f * D:(f + 1) * D], orc.GRAD_COMPAT, 0))
        gm.update(); om.apply_update()
        for li in range(3):
            bound(kv.get("fc%d.weights" % li), st.get("fc%d.weights" % li), ch.W[li].reshape(-1), "fc%d.weights after step %d" % (li, step), floor=ch.floor_W(li).reshape(-1))
    ds.close(); gm.close(); kv.close()


# ---- configs[3] -------------------------------------------------------------------------------------------------
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """ps_splitmix64 (ps_amd/csrc/ps_common.h) on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return x ^ (x >> np.uint64(31))


def table_rows(seed, rows_idx, D):
    """Rows of the synthetic gather table: float4 i = k_fill_table's hash of (seed, i)  (ps_ops.hip)."""
    q = (np.asarray(rows_idx, np.uint64)[:, None] * np.uint64(D // 4) + np.arange(D // 4, dtype=np.uint64)[None, :])
    h = splitmix64(np.uint64(seed) ^ q)
    parts = [((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.uint32).astype(f32) * f32(1.0 / 65536.0) - f32(0.5) for k in range(4)]
    return np.stack(parts, axis=-1).reshape(len(rows_idx), D).astype(f32)


@pytest.mark.parametrize("bag", [1, 32])
def test_config3_gather_outputs_on_the_256GB_table(bag):
    """configs[3]: one 1e9-row x 64-dim f32 table (256 GB of the 288 GB HBM), 2^22 random lookups in one launch:
    2048 output rows spread over the launch are bit-identical to relu(sum of the table rows, in bag order)."""
    import ps_amd
    from ps_amd import native as N
    rows, D, nlook, ns, seed = 1000 * 1000 * 1000, 64, 1 << 22, 2048, 0x5EED
    n = nlook // bag
    kv = ps_amd.KVStore(0, 1)
    bi = np.zeros(ns, np.int64); ids = np.zeros((ns, bag), np.int64); out = np.zeros((ns, D), f32)
    N.check(N.lib().ps_bench_gather_check(kv.h, rows, D, n, bag, seed, ns, bi.ctypes.data_as(C.POINTER(C.c_int64)),
                                          ids.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_float))))
    # the ids are the generator's: splitmix64((seed ^ 0xABCDEF) + j * golden) mod rows
    j = (bi[:, None] * bag + np.arange(bag)[None, :]).astype(np.uint64)
    with np.errstate(over="ignore"):
        want_ids = splitmix64((np.uint64(seed ^ 0xABCDEF) + j * np.uint64(0x9E3779B97F4A7C15)) & M64) % np.uint64(rows)
    np.testing.assert_array_equal(ids, want_ids.astype(np.int64))
    assert ids.max() > rows * 0.9 and len(np.unique(bi)) == ns           # the whole 256 GB is addressed
    acc = table_rows(seed, ids[:, 0], D)
    for k in range(1, bag):
        acc = (table_rows(seed, ids[:, k], D) + acc).astype(f32)        # sum pooling strictly in bag order
    np.testing.assert_array_equal(out, np.maximum(acc, f32(0)))
    kv.close()


def test_config3_fused_adam_on_320M_rows(orc):
    """configs[3], fused-Adam variant: DNN over ONE table of 320 M rows x 64 (W + Adam M,V = 246 GB), bags of 32
    uniform ids; after one step the touched rows equal orc.adam_update of (init row, GPU gradient, zero state) bit for
    bit, the gradient is the oracle's reduction of the GPU's delta, and untouched rows still hold their init values."""
    import ps_amd
    R, D, X, bag = 320 * 1000 * 1000, 64, 13, 32
    B = (1 << 19) // bag
    nnz = B * bag
    rng = np.random.default_rng(7)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([R], D)
    gm = ps_amd.DNN.buildModel(1, D, X, [256, 64, 1], store=kv, max_batch=B, max_nnz=nnz)
    ids = rng.integers(0, R, size=nnz).astype(np.int64)
    ids[5 * bag:5 * bag + 40] = ids[0]                                   # one key with 41 occurrences across two bags
    offsets = (np.arange(B + 1) * bag).astype(np.int64)
    Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.25).astype(f32)
    xav = orc.xavier_scale(1, D)
    loss = gm.forward({"E": ids, "X": Xd, "Y": Y, "offsets": offsets})
    assert np.isfinite(loss) and loss > 0.01
    # forward: pooled rows of sampled bags, bit-exact against the counter-based init
    A0 = gm.act(0)
    for b in (0, 5, 6, B // 2, B - 1):
        acc = None
        for i in ids[b * bag:(b + 1) * bag]:
            r = orc.init_rows(SEED, 0, [int(i)], D, xav)[0]
            acc = r if acc is None else (r + acc).astype(f32)
        np.testing.assert_array_equal(A0[b], np.maximum(acc, f32(0)))
    gm.backward()
    dx = gm.delta(2)
    uids, g = gm.emb_grads(0)
    np.testing.assert_array_equal(uids, np.unique(ids))
    bag_of = np.repeat(np.arange(B), bag)
    sample = np.unique(np.concatenate([[ids[0]], uids[:: max(1, len(uids) // 300)]]))
    pos = np.searchsorted(uids, sample)
    for k, idv in zip(pos, sample):
        ents = np.nonzero(ids == idv)[0]                                # entries in (bag, position) order
        np.testing.assert_array_equal(g[k], orc.emb_geff(dx[bag_of[ents]], orc.GRAD_COMPAT, 32), err_msg="id %d" % idv)
    gm.update()
    w1 = kv.get_rows(0, sample); m1 = kv.get_rows(0, sample, 1); v1 = kv.get_rows(0, sample, 2)
    w0 = orc.init_rows(SEED, 0, sample, D, xav)
    z = np.zeros(len(sample) * D, f32)
    we, me, ve = orc.adam_update(w0.reshape(-1), g[pos].reshape(-1), z, z)
    np.testing.assert_array_equal(w1.reshape(-1), we); np.testing.assert_array_equal(m1.reshape(-1), me)
    np.testing.assert_array_equal(v1.reshape(-1), ve)
    cold = np.setdiff1d(rng.integers(0, R, size=64), uids)
    np.testing.assert_array_equal(kv.get_rows(0, cold), orc.init_rows(SEED, 0, cold, D, xav))
    np.testing.assert_array_equal(kv.get_rows(0, cold, 1), np.zeros((len(cold), D), f32))
    gm.close(); kv.close()


# ---- configs[4] -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("idgen", IDGENS)
def test_config4_multi_hot_ftrl_full_size(orc, idgen):
    """configs[4]'s per-GPU shape: C2's model, bags of Poisson(30) ids per (sample, field) (~3.19 M ids per step, hot
    keys with tens of thousands of occurrences), sum pooling, FTRL on every embedding row.  Two runs are bit-identical
    (no float atomics anywhere), sampled pooled activations and per-key gradients match the oracle bit for bit, and
    the FTRL state of the sampled keys equals orc.ftrl_update of the GPU's gradient."""
    import ps_amd
    cfg = C2
    F, D, X, fc, V, B, WS = cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["V"], cfg["B"], cfg["wide"]
    rng = np.random.default_rng(5)
    lens = np.clip(rng.poisson(30, size=B * F), 1, 100)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(offsets[-1])
    assert nnz > 3_000_000
    from ps_amd import synth
    ids = synth.draw_ids(rng, 1.05, V, nnz, idgen)
    hot_id = V - 1 if idgen == "zipf_clamped" else 0
    Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.25).astype(f32)
    Wd = rng.integers(0, WS, size=(B, F)).astype(np.int64)
    bag_of = np.repeat(np.arange(B * F), lens)
    fld = bag_of % F
    xav = orc.xavier_scale(1, D)
    res = []
    for run in range(2):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        kv.set_updater("emF", ps_amd.FtrlUpdater())
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B, max_nnz=nnz)
        batch = {"E": ids, "X": Xd, "Y": Y, "W": Wd, "offsets": offsets}
        loss = gm.forward(batch)
        A0 = gm.act(0)
        gm.backward()
        dx = gm.delta(2)
        grads = [gm.emb_grads(f) for f in (0, 13, 25)]
        gm.update()
        if run == 0:
            # FTRL fused into the sparse scatter: (w, z, n) of sampled keys == orc.ftrl_update(init row, GPU gradient, 0, 0)
            uids, g = grads[0]
            pick = np.unique(np.concatenate([[0, len(uids) - 1], np.arange(0, len(uids), max(1, len(uids) // 200))]))
            w1 = kv.get_rows(0, uids[pick]); z1 = kv.get_rows(0, uids[pick], 1); n1 = kv.get_rows(0, uids[pick], 2)
            w0 = orc.init_rows(SEED, 0, uids[pick], D, xav)
            for i, k in enumerate(pick):
                we, ze, ne, _ = orc.ftrl_update(w0[i], g[k], np.zeros(D, f32), np.zeros(D, f32))
                np.testing.assert_array_equal(w1[i], we); np.testing.assert_array_equal(z1[i], ze); np.testing.assert_array_equal(n1[i], ne)
        loss2 = gm.train(batch)                                       # a second, fused step on the updated weights
        snap = [kv.get_rows(f, np.arange(0, V, 1)) for f in (0, 25)] + [kv.get_rows(0, np.arange(V), 1), kv.get_rows(0, np.arange(V), 2),
                                                                         kv.get("fc0.weights"), kv.get_wide(np.arange(0, WS, 7))]
        res.append((loss, loss2, A0, dx, grads, snap))
        if run == 0:
            # pooled activations of sampled bags: bit-exact against the init rows summed in bag order
            for bagi in (0, 1, F, B * F // 2 + 3, B * F - 1):
                acc = None
                for i in ids[offsets[bagi]:offsets[bagi + 1]]:
                    r = orc.init_rows(SEED, bagi % F, [int(i)], D, xav)[0]
                    acc = r if acc is None else (r + acc).astype(f32)
                b, f = divmod(bagi, F)
                np.testing.assert_array_equal(A0[b, f * D:(f + 1) * D], np.maximum(acc, f32(0)))
            # per-key gradients (incl. the hottest keys: id V-1 collects the Zipf tail) bit-exact vs the oracle
            for (uids, g), f in zip(grads, (0, 13, 25)):
                ent_f = np.nonzero(fld == f)[0]
                idf = ids[ent_f]
                cnt = np.bincount(idf, minlength=V)
                assert cnt[hot_id] > (30000 if idgen == "zipf_clamped" else 10000)
                np.testing.assert_array_equal(uids, np.nonzero(cnt)[0])
                sample = np.unique(np.concatenate([[hot_id, 0, 1], uids[:: max(1, len(uids) // 40)]]))
                for idv in sample:
                    ents = ent_f[idf == idv]                           # entries of the key in batch order
                    b_of = bag_of[ents] // F
                    k = int(np.searchsorted(uids, idv))
                    np.testing.assert_array_equal(g[k], orc.emb_geff(dx[b_of, f * D:(f + 1) * D], orc.GRAD_COMPAT, 32),
                                                  err_msg="emF%d.%d n=%d" % (f, idv, len(ents)))
        gm.close(); kv.close()
    a, b = res
    assert a[0] == b[0] and a[1] == b[1]
    np.testing.assert_array_equal(a[2], b[2]); np.testing.assert_array_equal(a[3], b[3])
    for (ia, ga), (ib, gb) in zip(a[4], b[4]):
        np.testing.assert_array_equal(ia, ib); np.testing.assert_array_equal(ga, gb)
    for x, y in zip(a[5], b[5]):
        np.testing.assert_array_equal(x, y)
