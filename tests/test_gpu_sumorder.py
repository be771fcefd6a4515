"""The order in which a key's per-sample gradients are added.

The REFERENCE adds strictly in sample order (layer/EmbeddingField.java:86-104, one addi per sample; then the second
pass of SURVEY App. A.6).  The HIP path has two orders (ps_model_config_t.emb_sum_order):
  * PS_SUM_SEQUENTIAL -- the reference's order for every key, whatever its count: bit-exact with
    orc.emb_geff(chunk = 0).  It is what PS_SUM_AUTO uses for single-hot batches, i.e. for every input the
    reference itself can express.
  * PS_SUM_CHUNKED -- runs above 32 entries as 32-entry chunks folded in two levels (bit-exact with
    orc.emb_geff(chunk = 32)); PS_SUM_AUTO uses it for multi-hot bags, which the reference does not have.
Keys with n = 33, 1 000, 4 096 and 63 000 occurrences are checked in both; for the chunked order the distance to the
reference order is measured (written to gpurun_out/sumorder_*.json, quoted in DESIGN.md) and bounded by the
reference order's OWN f32 roundoff against a float64 sum."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED
EPS = 2.0 ** -24
COUNTS = {0: 63000, 1: 4096, 2: 1000, 3: 33, 4: 32, 5: 31, 6: 65}


def multi_hot_case(rng, V, B, bag):
    nnz = B * bag
    ids = np.concatenate([np.full(c, k, np.int64) for k, c in COUNTS.items()])
    ids = np.concatenate([ids, 7 + rng.permutation(nnz - len(ids)) % (V - 7)]).astype(np.int64)
    ids = ids[rng.permutation(nnz)]
    return ids, (np.arange(B + 1) * bag).astype(np.int64)


@pytest.mark.parametrize("mode", ["compat", "intended"])
@pytest.mark.parametrize("order", ["sequential", "auto_chunked"])
def test_hot_keys_multi_hot(orc, mode, order):
    import ps_amd
    from ps_amd import native as N
    F, D, X, fc, V, B, bag = 1, 16, 3, [32, 1], 40000, 4096, 24
    rng = np.random.default_rng(21)
    ids, offsets = multi_hot_case(rng, V, B, bag)
    nnz = len(ids)
    Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.4).astype(f32)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V], D)
    gmode = N.PS_GRAD_COMPAT if mode == "compat" else N.PS_GRAD_INTENDED
    omode = orc.GRAD_COMPAT if mode == "compat" else orc.GRAD_INTENDED
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, max_nnz=nnz, emb_grad_mode=gmode,
                               emb_sum_order=N.PS_SUM_SEQUENTIAL if order == "sequential" else N.PS_SUM_AUTO)
    gm.forward({"E": ids, "X": Xd, "Y": Y, "offsets": offsets})
    gm.backward()
    dx = gm.delta(2)
    uids, g = gm.emb_grads(0)
    np.testing.assert_array_equal(uids, np.unique(ids))
    sample_of = np.repeat(np.arange(B), bag)
    report = {}
    for key, n in COUNTS.items():
        ents = np.nonzero(ids == key)[0]
        assert len(ents) == n
        gk = dx[sample_of[ents]]                                        # the key's per-sample gradients, in batch order
        got = g[int(np.searchsorted(uids, key))]
        seq = orc.emb_geff(gk, omode, 0)                                # the reference's order
        if order == "sequential" or n <= 32:
            np.testing.assert_array_equal(got, seq, err_msg="n=%d" % n)  # bit-exact with the reference order
            continue
        np.testing.assert_array_equal(got, orc.emb_geff(gk, omode, 32), err_msg="n=%d" % n)   # the chunked order, bit-exact
        scale = (n + 1) / (2.0 * n * n) if mode == "compat" else 1.0 / n
        ref64 = gk.astype(np.float64).sum(0) * scale
        mag = max(np.abs(ref64).max(), 1e-30)
        floor = 4 * EPS * (np.abs(gk).astype(np.float64).sum(0) * scale).max() / mag
        e_gpu = np.abs(got - ref64).max() / mag
        e_seq = np.abs(seq - ref64).max() / mag
        d = np.abs(got.astype(np.float64) - seq).max() / mag
        report[n] = {"gpu_vs_reference_order": float(d), "gpu_vs_float64": float(e_gpu), "reference_order_vs_float64": float(e_seq)}
        # the chunked order is at least as close to the exact sum as the reference's own order is, so its distance to
        # the reference result is bounded by (twice) the reference's own rounding error
        assert e_gpu <= e_seq + floor, "n=%d: chunked order is LESS accurate than the sequential one (%.3e vs %.3e)" % (n, e_gpu, e_seq)
        assert d <= 2 * e_seq + floor
    if report:
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        try:
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, "sumorder_%s.json" % mode), "w") as f:
                json.dump(report, f, indent=1, sort_keys=True)
        except OSError:
            pass
    gm.close(); kv.close()


def test_hot_keys_single_hot_use_the_reference_order(orc):
    """Single-hot batch of 4096 (what the reference feeds): keys with n = 4096, 2000, 1000, 33 and the tail, through the
    fused training step: gradients bit-exact with the sequential order, rows bit-exact with orc.adam_update of them."""
    import ps_amd
    F, D, X, fc, V, B = 2, 16, 2, [16, 1], 5000, 4096
    rng = np.random.default_rng(4)
    E = np.zeros((B, F), np.int64)
    E[:, 0] = 7                                                         # one key carried by every sample
    col = np.concatenate([np.full(2000, 1), np.full(1000, 2), np.full(33, 3), 10 + np.arange(B - 3033)])
    E[:, 1] = col[rng.permutation(B)]
    Xd = rng.standard_normal((B, X)).astype(f32); Y = (rng.random(B) < 0.4).astype(f32)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, keep_grads=True)
    uniq = [np.unique(E[:, f]) for f in range(F)]
    w0 = [kv.get_rows(f, uniq[f]) for f in range(F)]
    gm.train({"E": E, "X": Xd, "Y": Y})
    dx = gm.delta(2)
    for f in range(F):
        ids, g = gm.emb_grads(f)
        np.testing.assert_array_equal(ids, uniq[f])
        for i, idv in enumerate(ids):
            ks = np.nonzero(E[:, f] == idv)[0]
            np.testing.assert_array_equal(g[i], orc.emb_geff(dx[ks, f * D:(f + 1) * D], orc.GRAD_COMPAT, 0), err_msg="emF%d.%d n=%d" % (f, idv, len(ks)))
        z = np.zeros(g.size, f32)
        we, me, ve = orc.adam_update(w0[f].reshape(-1), g.reshape(-1), z, z)
        np.testing.assert_array_equal(kv.get_rows(f, ids).reshape(-1), we)
        np.testing.assert_array_equal(kv.get_rows(f, ids, 1).reshape(-1), me)
        np.testing.assert_array_equal(kv.get_rows(f, ids, 2).reshape(-1), ve)
    gm.close(); kv.close()
