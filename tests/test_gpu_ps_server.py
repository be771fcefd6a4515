"""`service PS` (ps.proto) served from a GPU shard: ps_amd/ps_server.py over real gRPC on the loopback interface.

Workers are PsClient objects (net/PSClient.java's calls).  What is checked is net/PServer.java's behaviour:
get / getList (null weights -> ec 204 / the key alone), upsert (no overwrite unless `replace`), push with an unknown
updater -> ec 500, BSP (push = sum in arrival order; the barrier of the last of worker_num workers runs psUpdate: sum /
count through the key's updater, globalStep + 1) and async (one updater step per push; barrier only bumps globalStep)
-- with the updated values compared bit for bit against the oracle's Adam / Ftrl on the same gradients."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED
ADAM = "adam@alfa:0.005@beta1:0.9@beta2:0.999@epsilon:1.0E-8"
FTRL = "ftrl@alfa:0.005@beta:1.0@l1:0.001@l2:0.001"


@pytest.fixture()
def shard():
    import ps_amd
    from ps_amd import ps_server as S
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([10, 6], 4)
    kv.create_wide(7)
    kv.create_fc(0, 3, 2)
    made = []

    def start(worker_num, is_async=False):
        server, port, sv = S.serve(kv, 0, worker_num, is_async)
        made.append(server)
        return S, "127.0.0.1:%d" % port, sv

    yield kv, start
    for s in made:
        s.stop(0)
    kv.close()


def names():
    import ps_amd
    return ps_amd.AdamUpdater().getName(), ps_amd.FtrlUpdater().getName()


def test_get_upsert_and_errors(shard):
    kv, start = shard
    S, target, _ = start(1)
    c = S.PsClient(target)
    row, ec = c.get("emF0.3.0")
    assert ec == 200
    np.testing.assert_array_equal(row, kv.get_rows(0, [3])[0])
    assert c.get("fc7.weights") == (None, 204)                       # store.get == null -> error(204, "null weights")
    got = c.getList(["emF1.5.0", "nosuch.key", "fc0.bias"])
    np.testing.assert_array_equal(got["emF1.5.0"], kv.get_rows(1, [5])[0])
    assert got["nosuch.key"] is None and len(got["fc0.bias"]) == 2   # a null matrix travels as the key alone
    w = kv.get("fc0.weights")
    new = np.arange(6, dtype=f32)
    res, ec = c.upsertList({"fc0.weights": new})                     # exists, no replace: the stored value comes back
    assert ec == 200 and res["fc0.weights"][1] is True
    np.testing.assert_array_equal(res["fc0.weights"][0], w)
    res, ec = c.upsertList({"fc0.weights": new}, replace=True)
    assert res["fc0.weights"][1] is False
    np.testing.assert_array_equal(kv.get("fc0.weights"), new)
    assert c.push("emF0.1.0", np.ones(4, f32), "no-such-updater@x:1") == 500     # updaterMap.get == null
    c.close()


def test_bsp_round_two_workers(shard, orc):
    kv, start = shard
    S, target, sv = start(2)
    adam, ftrl = names()
    a, b = S.PsClient(target, "w0"), S.PsClient(target, "w1")
    rng = np.random.default_rng(3)
    g = {k: rng.standard_normal(n).astype(f32) for k, n in
         [("a_e3", 4), ("b_e3", 4), ("b_e5", 4), ("a_fc", 6), ("b_fc", 6), ("a_fb", 2), ("a_w2", 1), ("b_w2", 1), ("b_wb", 1)]}
    before = {"e3": kv.get_rows(0, [3])[0], "e5": kv.get_rows(1, [5])[0], "fc": kv.get("fc0.weights"), "fb": kv.get("fc0.bias"),
              "w2": kv.get_wide([2]), "wb": kv.get("wide.bias"), "e4": kv.get_rows(0, [4])[0]}
    step0 = kv.global_step()
    # arrival order is the call order (one thread): a, b, b, a, ...
    assert a.push("emF0.3.0", g["a_e3"], adam) == 0
    assert b.push("emF0.3.0", g["b_e3"], adam) == 0
    assert b.push("emF1.5.0", g["b_e5"], adam) == 0
    assert a.push("fc0.weights", g["a_fc"], adam) == 0
    assert b.push("fc0.weights", g["b_fc"], adam) == 0
    assert a.push("fc0.bias", g["a_fb"], adam) == 0
    assert a.push("wide.weights.2.0", g["a_w2"], ftrl) == 0
    assert b.push("wide.weights.2.0", g["b_w2"], ftrl) == 0
    assert b.push("wide.bias", g["b_wb"], ftrl) == 0
    np.testing.assert_array_equal(kv.get_rows(0, [3])[0], before["e3"])          # nothing applied before the barrier
    done = []
    t = threading.Thread(target=lambda: done.append(a.barrier()), daemon=True)
    t.start()
    t.join(0.5)
    assert t.is_alive() and kv.global_step() == step0                             # the first worker waits for the second
    assert b.barrier() == 200
    t.join(20)
    assert done == [200] and kv.global_step() == step0 + 1

    def mean(*xs):
        s = xs[0].copy()
        for x in xs[1:]:
            s = (x + s).astype(f32)                                              # sum.addi(val), arrival order
        return (s / f32(len(xs))).astype(f32)                                    # divi(sumCnt)

    z = np.zeros
    np.testing.assert_array_equal(kv.get_rows(0, [3])[0], orc.adam_update(before["e3"], mean(g["a_e3"], g["b_e3"]), z(4, f32), z(4, f32))[0])
    np.testing.assert_array_equal(kv.get_rows(1, [5])[0], orc.adam_update(before["e5"], g["b_e5"], z(4, f32), z(4, f32))[0])
    np.testing.assert_array_equal(kv.get("fc0.weights"), orc.adam_update(before["fc"], mean(g["a_fc"], g["b_fc"]), z(6, f32), z(6, f32))[0])
    np.testing.assert_array_equal(kv.get("fc0.bias"), orc.adam_update(before["fb"], g["a_fb"], z(2, f32), z(2, f32))[0])
    np.testing.assert_array_equal(kv.get_wide([2]), orc.ftrl_update(before["w2"], mean(g["a_w2"], g["b_w2"]), z(1, f32), z(1, f32))[0])
    np.testing.assert_array_equal(kv.get("wide.bias"), orc.ftrl_update(before["wb"], g["b_wb"], z(1, f32), z(1, f32))[0])
    np.testing.assert_array_equal(kv.get_rows(0, [4])[0], before["e4"])          # an unpushed key is untouched
    # second round: the updater state carries over (Adam M, V of round 1)
    w1, M1, V1 = orc.adam_update(before["e3"], mean(g["a_e3"], g["b_e3"]), z(4, f32), z(4, f32))
    assert a.push("emF0.3.0", g["b_e5"], adam) == 0
    t = threading.Thread(target=lambda: done.append(a.barrier()), daemon=True)
    t.start()
    assert b.barrier() == 200
    t.join(20)
    np.testing.assert_array_equal(kv.get_rows(0, [3])[0], orc.adam_update(w1, g["b_e5"], M1, V1)[0])
    assert kv.global_step() == step0 + 2
    a.close(); b.close()


def test_async_pushes_apply_at_once(shard, orc):
    kv, start = shard
    S, target, _ = start(2, is_async=True)
    adam, _ = names()
    c = S.PsClient(target)
    rng = np.random.default_rng(4)
    g1, g2 = rng.standard_normal(4).astype(f32), rng.standard_normal(4).astype(f32)
    w0 = kv.get_rows(0, [7])[0]
    step0 = kv.global_step()
    assert c.push("emF0.7.0", g1, adam, is_async=True) == 0
    w1, M1, V1 = orc.adam_update(w0, g1, np.zeros(4, f32), np.zeros(4, f32))
    np.testing.assert_array_equal(kv.get_rows(0, [7])[0], w1)                     # no barrier needed
    assert c.push("emF0.7.0", g2, adam, is_async=True) == 0
    np.testing.assert_array_equal(kv.get_rows(0, [7])[0], orc.adam_update(w1, g2, M1, V1)[0])
    fc0 = kv.get("fc0.weights")
    gf = rng.standard_normal(6).astype(f32)
    assert c.push("fc0.weights", gf, adam, is_async=True) == 0
    np.testing.assert_array_equal(kv.get("fc0.weights"), orc.adam_update(fc0, gf, np.zeros(6, f32), np.zeros(6, f32))[0])
    assert c.barrier() == 200 and kv.global_step() == step0 + 1                   # async barrier: globalStep++ only
    c.close()


def test_push_update_sums_in_arrival_order(shard, orc):
    """ps_store_push_update (the C ABI under psUpdate), three pushes per key whose f32 sum depends on the order:
    (a + b) + c in ARRIVAL order, / 3, one updater step -- embedding row, dense tensor and wide key alike."""
    kv, _ = shard
    import ps_amd
    kv.set_updater("emF", ps_amd.AdamUpdater()); kv.set_updater("fc0.weights", ps_amd.AdamUpdater())
    kv.set_updater("wide", ps_amd.FtrlUpdater())
    big, small = f32(1.0e8), f32(3.0)
    trip = [big, small, -big]                                   # (1e8 + 3) - 1e8 = 0 in f32; 1e8 - 1e8 + 3 = 3: order matters
    e = [np.full(4, v, f32) for v in trip]
    d = [np.full(6, v, f32) for v in trip]
    w = [np.full(1, v, f32) for v in trip]
    e0, d0, w0 = kv.get_rows(0, [2])[0], kv.get("fc0.weights"), kv.get_wide([4])
    kv.push_update([("emF0.2.0", e[0]), ("fc0.weights", d[0]), ("wide.weights.4.0", w[0]),
                    ("emF0.2.0", e[1]), ("fc0.weights", d[1]), ("wide.weights.4.0", w[1]),
                    ("emF0.2.0", e[2]), ("fc0.weights", d[2]), ("wide.weights.4.0", w[2])])

    def mean(xs):
        s = xs[0].copy()
        for x in xs[1:]:
            s = (x + s).astype(f32)
        return (s / f32(len(xs))).astype(f32)

    assert mean(e)[0] == 0.0 and mean([e[0], e[2], e[1]])[0] == 1.0   # the order is visible in the result
    z = np.zeros
    np.testing.assert_array_equal(kv.get_rows(0, [2])[0], orc.adam_update(e0, mean(e), z(4, f32), z(4, f32))[0])
    np.testing.assert_array_equal(kv.get("fc0.weights"), orc.adam_update(d0, mean(d), z(6, f32), z(6, f32))[0])
    np.testing.assert_array_equal(kv.get_wide([4]), orc.ftrl_update(w0, mean(w), z(1, f32), z(1, f32))[0])
    # async: three updater steps, in order
    e1 = kv.get_rows(0, [3])[0]
    kv.push_update([("emF0.3.0", e[1]), ("emF0.3.0", e[0]), ("emF0.3.0", e[1])], is_async=True)
    wv, M, V = e1, z(4, f32), z(4, f32)
    for g in (e[1], e[0], e[1]):
        wv, M, V = orc.adam_update(wv, g, M, V)
    np.testing.assert_array_equal(kv.get_rows(0, [3])[0], wv)


@pytest.mark.parametrize("m", [33, 40, 257])
def test_push_update_more_than_32_workers(shard, orc, m):
    """with 32 < m <= 256 pushes of one dense tensor in a BSP round the slab pre-fold ignored the row window
    of "fc<i>.weights" / "fc<i>.bias" (wrong sums, writes outside the slabs, butterfly instead of arrival order).
    m pushes of both tensors, order-sensitive values: (..(g0 + g1) + ..) / m in arrival order, one Adam step, and the
    neighbouring tensor is untouched."""
    kv, _ = shard
    import ps_amd
    kv.set_updater("fc0.weights", ps_amd.AdamUpdater()); kv.set_updater("fc0.bias", ps_amd.AdamUpdater())
    rng = np.random.default_rng(m)
    gw = [(rng.standard_normal(6) * 10.0 ** rng.integers(-3, 6)).astype(f32) for _ in range(m)]
    gb = [(rng.standard_normal(2) * 10.0 ** rng.integers(-3, 6)).astype(f32) for _ in range(m)]
    w0, b0 = kv.get("fc0.weights"), kv.get("fc0.bias")
    msgs = []
    for j in range(m):
        msgs += [("fc0.weights", gw[j]), ("fc0.bias", gb[j])]
    kv.push_update(msgs)

    def mean(xs):
        s = xs[0].copy()
        for x in xs[1:]:
            s = (x + s).astype(f32)
        return (s / f32(len(xs))).astype(f32)

    z = np.zeros
    np.testing.assert_array_equal(kv.get("fc0.weights"), orc.adam_update(w0, mean(gw), z(6, f32), z(6, f32))[0])
    np.testing.assert_array_equal(kv.get("fc0.bias"), orc.adam_update(b0, mean(gb), z(2, f32), z(2, f32))[0])


def test_push_update_validates_every_message_before_touching_the_store(shard):
    """one bad message (unknown key, wrong length, row of another shard) fails the call with the store
    untouched -- not half-updated."""
    kv, _ = shard
    import ps_amd
    from ps_amd import native as N
    before = (kv.get_rows(0, np.arange(10)), kv.get("fc0.weights"), kv.get_wide(np.arange(7)))
    good = [("emF0.2.0", np.ones(4, f32)), ("wide.weights.4.0", np.ones(1, f32)), ("fc0.weights", np.ones(6, f32))]
    for bad in (("fc0.weights", np.ones(5, f32)), ("wide.weights.99.0", np.ones(1, f32)), ("emF0.2.0", np.ones(3, f32)),
                ("emF9.2.0", np.ones(4, f32)), ("nosuch", np.ones(1, f32))):
        with pytest.raises(N.PsError):
            kv.push_update(good + [bad])
        np.testing.assert_array_equal(kv.get_rows(0, np.arange(10)), before[0])
        np.testing.assert_array_equal(kv.get("fc0.weights"), before[1])
        np.testing.assert_array_equal(kv.get_wide(np.arange(7)), before[2])
