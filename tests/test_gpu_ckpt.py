"""Shard checkpoint / resume: train k steps, save, load into a FRESH store of the same geometry, continue --
bit-identical to the uninterrupted run (weights, Adam / Ftrl state, wide table, globalStep, losses)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED


def _mk(F, D, X, fc, V, B, WS, shard=0, nshards=1):
    import ps_amd
    kv = ps_amd.KVStore(0, SEED)
    kv.create_embedding([V] * F, D, shard=shard, nshards=nshards)
    gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
    return kv, gm


def _state(kv, F, V, WS, nfc):
    ids = np.arange(V)
    return ([kv.get_rows(f, ids, w) for f in range(F) for w in (0, 1, 2)] + [kv.get("fc%d.%s" % (l, k)) for l in range(nfc) for k in ("weights", "bias")]
            + [kv.get_wide(np.arange(WS), w) for w in (0, 1, 2)] + [kv.get("wide.bias")])


def test_resume_is_exact(tmp_path):
    import ps_amd
    F, D, X, fc, V, B, WS = 5, 8, 3, [32, 8, 1], 60, 128, 41
    rng = np.random.default_rng(8)
    data = []
    for _ in range(7):
        E = np.minimum(rng.zipf(1.3, size=(B, F)) - 1, V - 1).astype(np.int64)
        data.append({"E": E, "X": rng.standard_normal((B, X)).astype(f32), "Y": (rng.random(B) < 0.3).astype(f32), "W": E % WS})
    kv, gm = _mk(F, D, X, fc, V, B, WS)
    kv.set_updater("emF", ps_amd.AdamUpdater(alfa=0.01))                # a non-default updater travels with the checkpoint
    la = [gm.train(d) for d in data[:4]]
    path = str(tmp_path / "shard0.ck")
    kv.save(path)
    la += [gm.train(d) for d in data[4:]]
    want = _state(kv, F, V, WS, 3); step_a = kv.global_step()
    gm.close(); kv.close()
    kv2, gm2 = _mk(F, D, X, fc, V, B, WS)
    kv2.load(path)
    assert kv2.global_step() == 4
    lb = [gm2.train(d) for d in data[4:]]
    assert lb == la[4:]
    assert kv2.global_step() == step_a == 7
    for a, b in zip(want, _state(kv2, F, V, WS, 3)):
        np.testing.assert_array_equal(a, b)
    gm2.close(); kv2.close()


def test_load_rejects_another_geometry(tmp_path):
    import ps_amd
    from ps_amd import native as N
    kv, gm = _mk(3, 8, 2, [8, 1], 30, 16, 11)
    path = str(tmp_path / "a.ck")
    kv.save(path)
    gm.close(); kv.close()
    for args in ((3, 8, 2, [8, 1], 31, 16, 11), (3, 4, 2, [8, 1], 30, 16, 11), (3, 8, 2, [16, 1], 30, 16, 11), (3, 8, 2, [8, 1], 30, 16, 12)):
        kv2, gm2 = _mk(*args)
        with pytest.raises(N.PsError):
            kv2.load(path)
        gm2.close(); kv2.close()
    kv3, gm3 = _mk(3, 8, 2, [8, 1], 30, 16, 11)
    with pytest.raises(N.PsError):
        kv3.load(str(tmp_path / "missing.ck"))
    # a truncated / padded file is refused BEFORE anything reaches the device: the store is exactly what it was
    rng = np.random.default_rng(2)
    E = rng.integers(0, 30, size=(16, 3)).astype(np.int64)
    gm3.train({"E": E, "X": rng.standard_normal((16, 2)).astype(f32), "Y": (rng.random(16) < 0.5).astype(f32), "W": E % 11})
    before = _state(kv3, 3, 30, 11, 2); step = kv3.global_step()
    blob = open(path, "rb").read()
    for name, data in (("trunc.ck", blob[:-100]), ("padded.ck", blob + b"x" * 64), ("noend.ck", blob[:-8] + b"PSAMDXXX")):
        open(str(tmp_path / name), "wb").write(data)
        with pytest.raises(N.PsError):
            kv3.load(str(tmp_path / name))
        for a, b in zip(before, _state(kv3, 3, 30, 11, 2)):
            np.testing.assert_array_equal(a, b)
        assert kv3.global_step() == step
    # saving is atomic: an existing checkpoint is replaced only by a complete new one, no .tmp is left behind
    kv3.save(path)
    import os
    assert not os.path.exists(path + ".tmp") and os.path.getsize(path) == len(blob)
    gm3.close(); kv3.close()
