"""The stand-alone Layer operators of the boundary (what a GpuEmbeddingLayer / GpuFcLayer JNI shim binds):
ps_emb_forward (layer/EmbeddingLayer.java:25-48 + EmbeddingField.java:66-78) and ps_fc_forward
(layer/FcLayer.java:74-91), on caller-owned device buffers."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED


class Dev:
    def __init__(self, kv, a=None, nbytes=None):
        from ps_amd import native as N
        self.kv, self.N = kv, N
        self.p = C.c_void_p()
        self.nbytes = a.nbytes if a is not None else nbytes
        N.check(N.lib().ps_dev_alloc(kv.h, max(self.nbytes, 4), C.byref(self.p)))
        if a is not None and a.size:
            a = np.ascontiguousarray(a)
            N.check(N.lib().ps_dev_upload(kv.h, self.p, a.ctypes.data, a.nbytes))

    def get(self, shape, dtype):
        a = np.empty(shape, dtype)
        self.N.check(self.N.lib().ps_dev_download(self.kv.h, a.ctypes.data, self.p, a.nbytes))
        return a

    def free(self):
        self.N.lib().ps_dev_free(self.kv.h, self.p)


@pytest.mark.parametrize("D", [4, 10, 16, 64])
def test_emb_forward_single_hot_and_bags(D):
    import ps_amd
    from ps_amd import native as N
    F, V, B = 3, 37, 29
    rng = np.random.default_rng(D)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    tabs = [kv.get_rows(f, np.arange(V)) for f in range(F)]
    ld = ((F * D + 5 + 15) // 16) * 16
    out = Dev(kv, nbytes=B * ld * 4)
    # single-hot, relu (the reference's EmbeddingField) and no activation
    E = rng.integers(0, V, size=(B, F)).astype(np.int64)
    ids = Dev(kv, E)
    for act in (N.PS_ACT_RELU, N.PS_ACT_NONE):
        N.check(N.lib().ps_emb_forward(kv.h, ids.p, None, B, act, out.p, ld))
        kv.sync()
        got = out.get((B, ld), f32)
        for f in range(F):
            want = tabs[f][E[:, f]]
            np.testing.assert_array_equal(got[:, f * D:(f + 1) * D], np.maximum(want, 0) if act == N.PS_ACT_RELU else want)   # rcopy: bit-exact
    # bags (CSR over (sample, field)), incl. empty ones: sum in bag order, then relu
    lens = rng.integers(0, 6, size=B * F); lens[1] = 0
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bid = rng.integers(0, V, size=int(offs[-1])).astype(np.int64)
    dids, doffs = Dev(kv, bid), Dev(kv, offs)
    N.check(N.lib().ps_emb_forward(kv.h, dids.p, doffs.p, B, N.PS_ACT_RELU, out.p, ld))
    kv.sync()
    got = out.get((B, ld), f32)
    for bag in range(B * F):
        b, f = divmod(bag, F)
        s = np.zeros(D, f32)
        for k, p in enumerate(range(offs[bag], offs[bag + 1])):
            s = tabs[f][bid[p]].copy() if k == 0 else (tabs[f][bid[p]] + s).astype(f32)
        np.testing.assert_array_equal(got[b, f * D:(f + 1) * D], np.maximum(s, 0))
    # an id outside its table is reported (204), not a crash
    bad = E.copy(); bad[0, 0] = V + 3
    dbad = Dev(kv, bad)
    rc = N.lib().ps_emb_forward(kv.h, dbad.p, None, B, N.PS_ACT_RELU, out.p, ld)
    rc2 = N.lib().ps_store_sync(kv.h)
    assert N.PS_MISSING in (rc, rc2)
    for d in (out, ids, dids, doffs, dbad):
        d.free()
    kv.close()


@pytest.mark.parametrize("in_dims,out_dims,B", [(13, 7, 5), (429, 512, 300), (256, 1, 64)])
def test_fc_forward(in_dims, out_dims, B):
    import ps_amd
    from ps_amd import native as N
    rng = np.random.default_rng(in_dims)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_fc(0, in_dims, out_dims)
    W = kv.get("fc0.weights").astype(np.float64).reshape(in_dims, out_dims); b = kv.get("fc0.bias").astype(np.float64)
    ldx = ((in_dims + 1 + 15) // 16) * 16
    ldy = ((out_dims + 15) // 16) * 16
    x = np.zeros((B, ldx), f32)
    x[:, :in_dims] = rng.standard_normal((B, in_dims)).astype(f32)
    x[:, in_dims] = 1.0                                        # the ones column that carries the bias
    dx, dy = Dev(kv, x), Dev(kv, nbytes=B * ldy * 4)
    z = x[:, :in_dims].astype(np.float64) @ W + b
    mag = np.abs(x[:, :in_dims].astype(np.float64)) @ np.abs(W) + np.abs(b)
    for act, ref in ((N.PS_ACT_NONE, z), (N.PS_ACT_RELU, np.maximum(z, 0)), (N.PS_ACT_SIGMOID, 0.001 + 0.998 / (1 + np.exp(-z)))):
        N.check(N.lib().ps_fc_forward(kv.h, 0, act, dx.p, ldx, B, dy.p, ldy))
        kv.sync()
        got = dy.get((B, ldy), f32)[:, :out_dims].astype(np.float64)
        err = np.abs(got - ref) - 1e-5 * np.abs(ref) - 8 * 2.0 ** -24 * mag
        assert err.max() <= 0, (act, err.max())
    # wrong leading dimension is an argument error
    assert N.lib().ps_fc_forward(kv.h, 0, N.PS_ACT_NONE, dx.p, ldx + 16, B, dy.p, ldy) != 0
    assert N.lib().ps_fc_forward(kv.h, 3, N.PS_ACT_NONE, dx.p, ldx, B, dy.p, ldy) == N.PS_MISSING
    dx.free(); dy.free(); kv.close()
