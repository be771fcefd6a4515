"""Known-answer tests that pin the oracle: every expected
value here is derivable by hand from the reference source, no reference run
needed.  PARITY UNPINNED otherwise (the reference holds no golden vectors)."""
import numpy as np
import pytest

f32 = np.float32


def test_java_string_hashcode(orc):
    # h = 31*h + c, int32 wrap
    assert orc.java_hashcode("") == 0
    assert orc.java_hashcode("a") == 97
    assert orc.java_hashcode("ab") == 3105
    assert orc.java_hashcode("hello") == 99162322
    # wraps negative
    assert orc.java_hashcode("emF13.28305.0") == _py_hash("emF13.28305.0")
    assert orc.java_hashcode("fc0.weights") == _py_hash("fc0.weights")


def _py_hash(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h - (1 << 32) if h >= 1 << 31 else h


@pytest.mark.parametrize("v,s", [
    (0.0, "0.0"), (1.0, "1.0"), (28305.0, "28305.0"), (9999999.0, "9999999.0"),
    (1.0e7, "1.0E7"), (1.5, "1.5"), (0.001, "0.001"), (1.0e-4, "1.0E-4"),
    (123456.0, "123456.0"), (0.005, "0.005"), (1.0e-8, "1.0E-8"), (0.999, "0.999"),
    (16777216.0, "1.6777216E7"), (-3.0, "-3.0"), (100.0, "100.0"),
])
def test_float_to_string(orc, v, s):
    assert orc.float_to_string(v) == s


def test_key_format(orc):
    # debug literals in store/KVStore.java:204, net/PServer.java:93
    assert orc.emb_key(13, 28305.0) == "emF13.28305.0"
    assert orc.wide_key(23456.0) == "wide.weights.23456.0"


def test_matrixutil_hash(orc):
    assert orc.matrixutil_hash(123456.0, 100000) == 23456.0
    assert orc.matrixutil_hash(99999.0, 100000) == 99999.0


def test_mod_shard_sign(orc):
    # Java % keeps the dividend's sign (net/Mod.java:13-15): a negative hash
    # gives a negative shard in the reference; floorMod is the documented fix.
    keys = ["emF%d.%d.0" % (f, i) for f in range(4) for i in range(50)]
    neg = [k for k in keys if orc.java_hashcode(k) < 0 and orc.java_hashcode(k) % 8 != 0]
    assert neg, "need a negative-hash key"
    k = neg[0]
    assert orc.mod_shard(k, 8, False) < 0
    assert 0 <= orc.mod_shard(k, 8, True) < 8
    assert (orc.mod_shard(k, 8, True) - orc.mod_shard(k, 8, False)) == 8
    for k in keys:
        assert orc.mod_shard(k, 8, True) == orc.java_hashcode(k) % 8   # python % is floorMod


def test_clipped_sigmoid(orc):
    assert orc.sigmoid_clip(0.0) == f32(0.001 + 0.998 / 2) == f32(0.5)
    assert orc.sigmoid_clip(200.0) == f32(0.999)
    assert orc.sigmoid_clip(-200.0) == f32(0.001)


def test_ce_backward_through_sigmoid(orc):
    # p = 0.5, l = 1: delta_L = -2 ; after sigmoid' (x 0.25) = -0.5
    d = orc.ce_backward([0.5], [1.0])
    assert d[0] == f32(-2.0)
    assert f32(d[0] * f32(0.5) * f32(1 - 0.5)) == f32(-0.5)
    assert abs(orc.ce_forward([0.5, 0.5], [1.0, 0.0]) - np.log(2)) < 1e-6


def test_adam_first_step(orc):
    # zero state: mhat ~ g, vhat ~ g^2 -> dw ~ -alfa * sign(g)
    g = np.array([0.3, -2.0, 1e-3, 5.0], f32)
    w, M, V = orc.adam_update(np.zeros(4), g, np.zeros(4), np.zeros(4))
    np.testing.assert_allclose(w, -0.005 * np.sign(g), rtol=2e-5)
    c1 = f32(1) - f32(0.9)
    c2 = f32(1) - f32(0.999)
    assert c1 == f32(0.100000024) and c2 == f32(0.0009999871)
    np.testing.assert_array_equal(M, g * c1)
    np.testing.assert_array_equal(V, (g * g) * c2)
    # untouched rows do not decay: state is per key (lazy Adam for free)
    w2, M2, V2 = orc.adam_update(w, g, M, V)
    assert np.all(np.abs(w2) > np.abs(w))


def test_ftrl_two_steps(orc):
    # step 1, zero state, g = 0.5 : w -> 0 ; s = 0.5 ; z = 0.5 ; n = 0.25
    w, z, n, did = orc.ftrl_update([0.7], [0.5], [0.0], [0.0])
    assert did == 1 and w[0] == 0 and z[0] == f32(0.5) and n[0] == f32(0.25)
    # step 2 : w = -(0.5-0.001)/((0.001+1+0.5)/0.005) before z,n move
    w2, z2, n2, _ = orc.ftrl_update(w, [0.25], z, n)
    np.testing.assert_allclose(w2[0], -0.499 / 300.2, rtol=1e-6)
    # g[0] == 0 -> untouched  (update/FtrlUpdater.java:52-54)
    w3, z3, n3, did3 = orc.ftrl_update([0.7], [0.0], [0.1], [0.2])
    assert did3 == 0 and w3[0] == f32(0.7) and z3[0] == f32(0.1) and n3[0] == f32(0.2)


@pytest.mark.parametrize("n,fac", [(1, 1.0), (2, 0.75), (3, 2.0 / 3.0), (4, 0.625)])
def test_duplicate_factor(orc, n, fac):
    # one key seen n times with equal g: g_eff = n*g*(n+1)/(2n^2)  (App. A.6)
    g = np.array([0.5, -1.0, 2.0], f32)
    gk = np.tile(g, (n, 1))
    out = orc.emb_geff(gk, orc.GRAD_COMPAT)
    np.testing.assert_allclose(out, g * fac, rtol=3e-7)
    if n == 1:
        np.testing.assert_array_equal(out, g)      # n = 1 is exact
    np.testing.assert_allclose(orc.emb_geff(gk, orc.GRAD_INTENDED), g, rtol=3e-7)


def test_geff_chunked_order_equals_sequential_for_short(orc):
    rng = np.random.default_rng(0)
    gk = rng.standard_normal((7, 4)).astype(f32)
    np.testing.assert_array_equal(orc.emb_geff(gk, 0, 0), orc.emb_geff(gk, 0, 8))
    a, b = orc.emb_geff(gk, 0, 0), orc.emb_geff(gk, 0, 2)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def test_init_distribution(orc):
    s = orc.xavier_scale(1, 16)
    assert s == f32(4 * np.sqrt(6) / np.sqrt(17))
    rows = orc.init_rows(7, 3, range(200), 16, s)
    assert np.all(np.abs(rows) < s) and 0.4 < (rows > 0).mean() < 0.6
    assert abs(np.abs(rows).mean() - s / 2) < 0.1 * s
    # pure function of the key
    assert rows[5, 2] == orc.init_value(7, 3, 5, 2, s)
