"""The layer-granular backward operators of the C ABI -- what a Java GpuFcLayer / GpuEmbeddingLayer binds for
Layer.backward() (layer/Layer.java:39-45) -- against the oracle:

  ps_fc_backward          FcLayer.backward (layer/FcLayer.java:93-110): act' in place, db, dW -> KVStore.sum, W^T delta
  ps_dense_update         KVStore.update(Map) + clear for the dense tensors (store/KVStore.java:240-277)
  ps_emb_backward_update  EmbeddingLayer/Field.backward twice + sum + update (layer/EmbeddingField.java:86-104)

FP32 contractions: float64 on the GPU's own inputs, 1e-5 relative + f32 roundoff floor.  Everything that is a sequence
of individually rounded f32 ops (per-key reduction, Adam, Ftrl, the mean over pending sums): bit-exact."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import close64

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED


class Dev:
    """A device buffer behind the store's ps_dev_* calls."""

    def __init__(self, kv, arr):
        from ps_amd import native as N
        self.kv, self.N = kv, N
        self.shape, self.dtype = arr.shape, arr.dtype
        self.p = C.c_void_p()
        a = np.ascontiguousarray(arr)
        N.check(N.lib().ps_dev_alloc(kv.h, max(a.nbytes, 16), C.byref(self.p)))
        if a.nbytes:
            N.check(N.lib().ps_dev_upload(kv.h, self.p, a.ctypes.data, a.nbytes))

    def get(self):
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            self.N.check(self.N.lib().ps_dev_download(self.kv.h, out.ctypes.data, self.p, out.nbytes))
        return out

    def free(self):
        self.N.lib().ps_dev_free(self.kv.h, self.p)


def rup(x, m):
    return (x + m - 1) // m * m


def test_fc_backward_and_dense_update(orc):
    import ps_amd
    from ps_amd import native as N
    L = N.lib()
    B, K0, N0, N1 = 200, 37, 24, 5
    rng = np.random.default_rng(3)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_fc(0, K0, N0)
    kv.create_fc(1, N0, N1)
    kv.set_updater("fc1.weights", ps_amd.AdamUpdater(alfa=0.01))        # exact key wins over "default"
    kp0, kp1, ld0, ld1 = rup(K0 + 1, 16), rup(N0 + 1, 16), rup(N0, 16), rup(N1, 16)
    x0h = np.zeros((B, kp0), f32); x0h[:, :K0] = rng.standard_normal((B, K0)); x0h[:, K0] = 1
    x1h = np.zeros((B, kp1), f32); x1h[:, N0] = 1
    x0, x1 = Dev(kv, x0h), Dev(kv, x1h)
    y1 = Dev(kv, np.zeros((B, ld1), f32))
    # forward: x1 = relu(fc0(x0)) written into the next layer's input buffer, y1 = fc1(x1)
    N.check(L.ps_fc_forward(kv.h, 0, N.PS_ACT_RELU, x0.p, kp0, B, x1.p, kp1))
    N.check(L.ps_fc_forward(kv.h, 1, N.PS_ACT_SIGMOID, x1.p, kp1, B, y1.p, ld1))
    W0 = kv.get("fc0.weights").reshape(K0, N0).astype(np.float64); W1 = kv.get("fc1.weights").reshape(N0, N1).astype(np.float64)
    b0 = kv.get("fc0.bias"); b1 = kv.get("fc1.bias")
    A1 = x1.get()[:, :N0].astype(np.float64); Y1 = y1.get()[:, :N1].astype(np.float64)
    sums = {0: [0, 0], 1: [0, 0]}
    for rnd in range(2):                                                # two replicas' worth of KVStore.sum
        d1h = np.zeros((B, ld1), f32); d1h[:, :N1] = rng.standard_normal((B, N1))
        d1 = Dev(kv, d1h)
        d0 = Dev(kv, np.zeros((B, ld0), f32))                           # fc1.delta = next.delta of fc0
        N.check(L.ps_fc_backward(kv.h, 1, N.PS_ACT_SIGMOID, x1.p, kp1, y1.p, ld1, d1.p, ld1, B, d0.p, ld0))
        dm = d1.get()[:, :N1].astype(np.float64)                        # act' was applied in place
        want_dm = d1h[:, :N1].astype(np.float64) * (Y1 * (1 - Y1))
        assert np.abs(dm - want_dm).max() <= 1e-6 * np.abs(want_dm).max()
        got_d0 = d0.get()
        assert not got_d0[:, N0:].any()
        close64(got_d0[:, :N0], dm @ W1.T, np.abs(dm) @ np.abs(W1).T, "fc1.delta = W^T delta")
        N.check(L.ps_fc_backward(kv.h, 0, N.PS_ACT_RELU, x0.p, kp0, x1.p, kp1, d0.p, ld0, B, None, 0))
        d0m = d0.get()[:, :N0].astype(np.float64)
        np.testing.assert_array_equal(d0m, got_d0[:, :N0].astype(np.float64) * (A1 > 0))      # relu' in place
        for layer, A, dlt, K, Nn in ((1, A1, dm, N0, N1), (0, x0h[:, :K0].astype(np.float64), d0m, K0, N0)):
            sums[layer][0] = sums[layer][0] + A.T @ dlt / B
            sums[layer][1] = sums[layer][1] + dlt.mean(0)
        d1.free(); d0.free()
    for layer, K, Nn, A in ((0, K0, N0, x0h[:, :K0]), (1, N0, N1, A1)):
        gw = np.empty(K * Nn, f32); gb = np.empty(Nn, f32); cnt = C.c_int()
        N.check(L.ps_fc_pending_grad(kv.h, layer, 0, gw.ctypes.data_as(C.POINTER(C.c_float)), gw.size, C.byref(cnt)))
        N.check(L.ps_fc_pending_grad(kv.h, layer, 1, gb.ctypes.data_as(C.POINTER(C.c_float)), gb.size, None))
        assert cnt.value == 2
        mag = 2 * np.abs(sums[layer][0]).max()
        assert np.abs(gw.reshape(K, Nn) - sums[layer][0]).max() <= 1e-5 * mag + 1e-6
        assert np.abs(gb - sums[layer][1]).max() <= 1e-5 * np.abs(sums[layer][1]).max() + 1e-6
        # KVStore.update: g = sum / cnt, then the layer's updater -- bit-exact given the GPU's own sums
        w_before = kv.get("fc%d.weights" % layer); b_before = kv.get("fc%d.bias" % layer)
        N.check(L.ps_dense_update(kv.h, layer))
        alfa = 0.01 if layer == 1 else 0.005
        z = np.zeros_like(gw)
        we, _, _ = orc.adam_update(w_before, (gw / f32(2)).astype(f32), z, z, alfa=alfa)
        np.testing.assert_array_equal(kv.get("fc%d.weights" % layer), we)
        zb = np.zeros_like(gb)
        be, _, _ = orc.adam_update(b_before, (gb / f32(2)).astype(f32), zb, zb, alfa=alfa)
        np.testing.assert_array_equal(kv.get("fc%d.bias" % layer), be)
        assert L.ps_dense_update(kv.h, layer) == N.PS_MISSING            # cleared
    # the transposed copy the forward reads was rewritten too: forward again == float64 on the NEW weights
    N.check(L.ps_fc_forward(kv.h, 0, N.PS_ACT_RELU, x0.p, kp0, B, x1.p, kp1))
    W0n = kv.get("fc0.weights").reshape(K0, N0).astype(np.float64); b0n = kv.get("fc0.bias").astype(np.float64)
    z0 = x0h[:, :K0].astype(np.float64) @ W0n + b0n
    close64(x1.get()[:, :N0], np.maximum(z0, 0), np.abs(x0h[:, :K0]).astype(np.float64) @ np.abs(W0n) + np.abs(b0n), "fc0 forward after the update")
    assert kv.global_step() == 2
    for d in (x0, x1, y1):
        d.free()
    kv.close()


@pytest.mark.parametrize("multi_hot,updater", [(False, "adam"), (True, "ftrl")])
def test_emb_backward_update(orc, multi_hot, updater):
    import ps_amd
    from ps_amd import native as N
    L = N.lib()
    F, D, V, B = 3, 8, 40, 96
    rng = np.random.default_rng(6)
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D)
    if updater == "ftrl":
        kv.set_updater("emF", ps_amd.FtrlUpdater())
    if multi_hot:
        lens = rng.integers(0, 5, size=B * F); lens[3] = 0; lens[7] = 45       # an empty bag, a run above one chunk
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
        ids[offsets[7]:offsets[8]] = 2
    else:
        offsets = None
        ids = np.minimum(rng.zipf(1.4, size=(B, F)) - 1, V - 1).astype(np.int64)
        ids[:, 2] = 9                                                          # n = B > 32: a long-key wave
    nnz = ids.size
    ld = rup(F * D, 16)
    ids_d = Dev(kv, ids.reshape(-1)); off_d = Dev(kv, offsets) if multi_hot else None
    a_d = Dev(kv, np.zeros((B, ld), f32))
    N.check(L.ps_emb_forward(kv.h, ids_d.p, off_d.p if multi_hot else None, B, N.PS_ACT_RELU, a_d.p, ld))
    A = a_d.get()
    delta = np.zeros((B, ld), f32); delta[:, :F * D] = rng.standard_normal((B, F * D))
    dl_d = Dev(kv, delta)
    allid = np.arange(V)
    before = [[kv.get_rows(f, allid, w) for w in range(3)] for f in range(F)]
    N.check(L.ps_emb_backward_update(kv.h, ids_d.p, off_d.p if multi_hot else None, nnz, B, N.PS_ACT_RELU, a_d.p, ld, dl_d.p, ld,
                                     N.PS_GRAD_COMPAT, N.PS_SUM_AUTO, 1))
    kv.sync()
    masked = delta * (A > 0)
    flat = ids.reshape(-1)
    bag_of = np.repeat(np.arange(B * F), np.diff(offsets)) if multi_hot else np.arange(B * F)
    nrows = C.c_int64()
    N.check(L.ps_emb_last_grads(kv.h, None, None, 0, C.byref(nrows)))
    rows = np.empty(nrows.value, np.int64); grads = np.empty((nrows.value, D), f32)
    N.check(L.ps_emb_last_grads(kv.h, rows.ctypes.data_as(C.POINTER(C.c_int64)), grads.ctypes.data_as(C.POINTER(C.c_float)), nrows.value, C.byref(nrows)))
    seen = 0
    for f in range(F):
        ent_f = np.nonzero(bag_of % F == f)[0]
        for idv in np.unique(flat[ent_f]):
            ents = ent_f[flat[ent_f] == idv]                                   # the key's entries in batch order
            gk = np.stack([masked[bag_of[e] // F, f * D:(f + 1) * D] for e in ents])
            g = orc.emb_geff(gk, orc.GRAD_COMPAT, 32 if multi_hot else 0)      # AUTO: reference order when single-hot
            k = int(np.searchsorted(rows, f * V + idv))
            assert rows[k] == f * V + idv
            np.testing.assert_array_equal(grads[k], g, err_msg="emF%d.%d n=%d" % (f, idv, len(ents)))
            w0, s1, s2 = before[f][0][idv], before[f][1][idv], before[f][2][idv]
            if updater == "adam":
                we, ae, be = orc.adam_update(w0, g, s1, s2)
            else:
                we, ae, be, _ = orc.ftrl_update(w0, g, s1, s2)
            np.testing.assert_array_equal(kv.get_rows(f, [idv])[0], we)
            np.testing.assert_array_equal(kv.get_rows(f, [idv], 1)[0], ae); np.testing.assert_array_equal(kv.get_rows(f, [idv], 2)[0], be)
            seen += 1
        untouched = np.setdiff1d(allid, np.unique(flat[ent_f]))
        np.testing.assert_array_equal(kv.get_rows(f, untouched), before[f][0][untouched])
    assert seen == nrows.value and kv.global_step() == 1
    # error paths: wrong nnz for single-hot, missing layer output
    assert L.ps_emb_backward_update(kv.h, ids_d.p, None, nnz + 1, B, N.PS_ACT_RELU, a_d.p, ld, dl_d.p, ld, 0, 0, 1) == (N.PS_E_BAD_ARG if not multi_hot else N.PS_E_BAD_ARG)
    assert L.ps_emb_backward_update(kv.h, ids_d.p, off_d.p if multi_hot else None, nnz, B, N.PS_ACT_RELU, None, ld, dl_d.p, ld, 0, 0, 1) == N.PS_E_BAD_ARG
    for d in (ids_d, a_d, dl_d) + ((off_d,) if multi_hot else ()):
        d.free()
    kv.close()
