"""The ingest pipeline on the GPU (ps_ingest_*: host threads parse batch k+1 into pinned memory and copy it to
HBM while batch k trains): the device batches are the parser's arrays bit for bit, and training from the
pipeline == training from the same arrays handed over as host batches (CTR.java's C1 shape)."""
import ctypes as C

import numpy as np
import pytest

from test_ingest_cpu import make_text

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED


def _down(kv, ptr, shape, dtype):
    from ps_amd import native as N
    a = np.empty(shape, dtype)
    if a.size:
        N.check(N.lib().ps_dev_download(kv.h, a.ctypes.data, ptr, a.nbytes))
    return a


def test_device_batches_are_the_parsed_arrays(orc):
    import ps_amd
    rng = np.random.default_rng(3)
    F, X, WS, B = 23, 45, 1000, 64
    text = make_text(rng, 300, F, X)
    E, Xd, Y, W = orc.parse_libsvm(text, F, X, WS)
    kv = ps_amd.KVStore(0, SEED)
    ds = ps_amd.DataSet(kv, text.encode(), F, X, B, wide_size=WS, threads=3)
    assert ds.lines() == 300
    for epoch in range(2):
        seen = 0
        for b in ds:
            n = b.B
            assert n == min(B, 300 - seen)
            kv.sync()
            np.testing.assert_array_equal(_down(kv, b.c.ids, (n, F), np.int64), E[seen:seen + n].astype(np.int64))
            np.testing.assert_array_equal(_down(kv, b.c.wide_ids, (n, F), np.int64), W[seen:seen + n].astype(np.int64))
            np.testing.assert_array_equal(_down(kv, b.c.dense, (n, X), f32), Xd[seen:seen + n])
            np.testing.assert_array_equal(_down(kv, b.c.labels, (n,), f32), Y[seen:seen + n])
            seen += n
        assert seen == 300
        assert ds.next() is None                     # FileSource returns null at the end
        ds.reset()
    ds.close(); kv.close()


def test_training_from_the_pipeline_equals_host_batches():
    import ps_amd
    rng = np.random.default_rng(4)
    F, D, X, fc, V, B = 23, 10, 45, [150, 10, 1], 50, 100          # DNN.buildModel(23, 10, 45, {150,10,1}) (CTR.java:91)
    lines = []
    for i in range(730):
        cols = [str(int(rng.random() < 0.4))] + ["%d:1" % int(rng.integers(0, V)) for _ in range(F)]
        cols += ["%d:%.5f" % (F + 1 + j, rng.standard_normal()) for j in range(X)]
        lines.append(" ".join(cols))
    text = "\n".join(lines).encode()
    parsed = ps_amd.LibsvmParser(F, X).parse(text)
    res = []
    for mode in ("host", "pipeline"):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
        losses = []
        if mode == "host":
            for s in range(0, 730, B):
                losses.append(gm.train(ps_amd.Batch(parsed["E"][s:s + B], parsed["X"][s:s + B], parsed["Y"][s:s + B])))
        else:
            ds = ps_amd.DataSet(kv, text, F, X, B, threads=4)
            for b in ds:
                losses.append(gm.train(b))
            ds.close()
        res.append((losses, [kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)]))
        gm.close(); kv.close()
    assert len(res[0][0]) == 8 and res[0][0] == res[1][0]
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        np.testing.assert_array_equal(a, b)


def test_two_readers_split_the_file_like_datasource_offset_step(orc):
    import ps_amd
    rng = np.random.default_rng(6)
    text = make_text(rng, 41, 4, 3)
    kv = ps_amd.KVStore(0, SEED)
    for offset in (0, 1):
        E, Xd, Y, _ = orc.parse_libsvm(text, 4, 3, 0, offset, 2)
        ds = ps_amd.DataSet(kv, text.encode(), 4, 3, 8, offset=offset, step=2, threads=2)
        got = []
        for b in ds:
            kv.sync()
            got.append(_down(kv, b.c.ids, (b.B, 4), np.int64))
        np.testing.assert_array_equal(np.concatenate(got), E.astype(np.int64))
        ds.close()
    kv.close()


def test_the_epoch_loop_in_c_and_a_ring_that_wraps():
    """ps_ingest_train (CTR.java:84-100's loop in C) over three epochs of 41 batches through a ring of 4 slots (threads = 1: every
    slot is reused ten times per epoch, every reuse waits for the kernels that read it) == the same batches handed over one by
    one as host batches, bit for bit."""
    import ps_amd
    rng = np.random.default_rng(8)
    F, D, X, fc, V, B = 5, 8, 3, [16, 1], 300, 64
    n = 41 * B - 17                                              # a short last batch
    lines = [" ".join([str(int(rng.random() < 0.4))] + ["%d:1" % int(rng.integers(0, V)) for _ in range(F)] +
                      ["%d:%.5f" % (F + 1 + j, rng.standard_normal()) for j in range(X)]) for _ in range(n)]
    text = "\n".join(lines).encode()
    parsed = ps_amd.LibsvmParser(F, X).parse(text)
    res = []
    for mode in ("host", "c loop, 1 parser thread", "c loop, 7 parser threads"):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
        if mode == "host":
            for _ in range(3):
                for s in range(0, n, B):
                    gm.train_async(ps_amd.Batch(parsed["E"][s:s + B], parsed["X"][s:s + B], parsed["Y"][s:s + B]))
        else:
            ds = ps_amd.DataSet(kv, text, F, X, B, threads=1 if "1 parser" in mode else 7)
            for ep in range(3):
                if ep == 1:                                      # an epoch in two calls
                    assert ds.train(gm, 10) == 10 and ds.train(gm) == 31
                else:
                    assert ds.train(gm) == 41
                assert ds.train(gm) == 0                         # dry: nothing trained, no error
                ds.reset()
            ds.close()
        kv.sync()
        res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(2)], kv.global_step()))
        gm.close(); kv.close()
    for got in res[1:]:
        assert got[2] == res[0][2] == 3 * 41
        for a, b in zip(res[0][0] + res[0][1], got[0] + got[1]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("via_float", [True, False])
def test_compact_and_full_size_batches_hold_the_same_arrays(via_float):
    """Round 6: a batch whose ids all fit an int32 crosses the link as [dense | labels | ids32] and a kernel rebuilds the two int64
    arrays (ids; wide ids = the parser's own function of the id, through a float or not); a batch with one id beyond int32 crosses
    as the int64 arrays.  Both kinds, mixed in one epoch, equal the host parser's arrays bit for bit -- negative ids included."""
    import ps_amd
    rng = np.random.default_rng(21)
    F, X, WS, B, n = 4, 2, 1000, 32, 32 * 6
    ids = rng.integers(-5000, 1 << 24, size=(n, F))
    ids[40, 2] = (1 << 40) + 12345                               # batch 1: beyond int32 -> the full-size copy
    ids[100, 0] = -(1 << 33)                                     # batch 3 too
    lines = [" ".join(["1"] + ["%d:1" % v for v in ids[i]] + ["%d:%.4f" % (F + 1 + j, rng.standard_normal()) for j in range(X)]) for i in range(n)]
    text = "\n".join(lines).encode()
    want = ps_amd.LibsvmParser(F, X, WS, ids_via_float=via_float).parse(text)
    kv = ps_amd.KVStore(0, SEED)
    ds = ps_amd.DataSet(kv, text, F, X, B, wide_size=WS, threads=3, ids_via_float=via_float)
    seen = 0
    for b in ds:
        kv.sync()
        np.testing.assert_array_equal(_down(kv, b.c.ids, (b.B, F), np.int64), want["E"][seen:seen + b.B])
        np.testing.assert_array_equal(_down(kv, b.c.wide_ids, (b.B, F), np.int64), want["W"][seen:seen + b.B])
        np.testing.assert_array_equal(_down(kv, b.c.dense, (b.B, X), f32), want["X"][seen:seen + b.B])
        seen += b.B
    assert seen == n
    ds.close(); kv.close()
