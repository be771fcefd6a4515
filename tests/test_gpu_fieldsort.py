"""The one-launch field sort of single-hot batches (kernels_sort.hip k_field_sort_segments) against the general radix
sort + segment builder it replaces: the same batch trained through both (ps_tune_set("field_sort", 0) selects the
general path) must leave bit-identical tables, gradients and unique-key lists -- the two produce the same sorted
pairs and segments by construction, and the long-key role of k_emb_reduce_update then walks the sort's list of long
runs instead of searching tiles for them.  Shapes: batch sizes on both sides of every workgroup layout (1, < 64,
non powers of two, 1024/4096/8192 exactly), fields of different vocabularies (1 row .. 2^20), all-equal ids (one run
of B entries per field), all-distinct ids, and more fields than fit one look-back sweep is not reachable (F < 65536)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED

CASES = [
    # B, vocab per field, id pattern
    (1, [5, 7], "rand"),
    (37, [3, 1, 50], "rand"),
    (64, [10, 10], "same"),
    (1000, [100000, 17, 1, 4096], "zipf"),
    (1024, [1 << 20, 9], "rand"),
    (1500, [300, 70000], "same"),
    (4096, [100000] * 5, "zipf"),
    (4096, [100000] * 5, "zipf_trunc"),
    (5000, [2000, 100000, 33], "zipf"),
    (8192, [50000, 8], "distinct"),
]


def make_ids(rng, B, vocab, pattern):
    cols = []
    for V in vocab:
        if pattern == "same":
            c = np.full(B, V - 1, np.int64)
        elif pattern == "distinct" and V >= B:
            c = rng.permutation(V)[:B].astype(np.int64)
        elif pattern == "zipf":            # rounds 1-2's generator: the unbounded tail piles up on id V - 1
            c = np.minimum(rng.zipf(1.05, B) - 1, V - 1).astype(np.int64)
        elif pattern == "zipf_trunc":      # Zipf(1.05) over V by inverse CDF
            from ps_amd import synth
            c = synth.zipf_truncated(rng, 1.05, V, B)
        else:
            c = rng.integers(0, V, B).astype(np.int64)
        cols.append(c)
    return np.stack(cols, 1)


def run(field_sort, B, vocab, E, Xd, Y, steps=2):
    import ps_amd
    from ps_amd import native as N
    N.lib().ps_tune_set(b"field_sort", field_sort)
    try:
        F, D = len(vocab), 8
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding(vocab, D)
        gm = ps_amd.DNN.buildModel(F, D, Xd.shape[1], [16, 1], store=kv, max_batch=B)
        for _ in range(steps):
            gm.train({"E": E, "X": Xd, "Y": Y})
        rows = [kv.get_rows(f, np.arange(vocab[f])) for f in range(F)]
        gm.forward({"E": E, "X": Xd, "Y": Y}); gm.backward()
        grads = [gm.emb_grads(f) for f in range(F)]
        gm.close(); kv.close()
        return rows, grads
    finally:
        N.lib().ps_tune_set(b"field_sort", 1)


@pytest.mark.parametrize("B,vocab,pattern", CASES)
def test_field_sort_equals_general_sort(B, vocab, pattern):
    rng = np.random.default_rng(B * 31 + len(vocab))
    E = make_ids(rng, B, vocab, pattern)
    Xd = rng.standard_normal((B, 2)).astype(f32)
    Y = (rng.random(B) < 0.3).astype(f32)
    rows_f, grads_f = run(1, B, vocab, E, Xd, Y)
    rows_g, grads_g = run(0, B, vocab, E, Xd, Y)
    for f in range(len(vocab)):
        np.testing.assert_array_equal(rows_f[f], rows_g[f], err_msg="field %d rows" % f)
        np.testing.assert_array_equal(grads_f[f][0], grads_g[f][0], err_msg="field %d unique ids" % f)
        np.testing.assert_array_equal(grads_f[f][0], np.unique(E[:, f]))
        np.testing.assert_array_equal(grads_f[f][1], grads_g[f][1], err_msg="field %d gradients" % f)
    # something was trained at all
    assert any(np.abs(g[1]).max() > 0 for g in grads_f)


def test_long_run_list_many_runs(orc):
    """More long runs than the long-key grid has workgroups (512): every id repeated 20 times, 1000 ids per field."""
    import ps_amd
    B, vocab = 8000, [400, 400]
    rng = np.random.default_rng(5)
    E = np.stack([rng.permutation(np.repeat(np.arange(400), 20)).astype(np.int64) for _ in vocab], 1)
    Xd = rng.standard_normal((B, 2)).astype(f32)
    Y = (rng.random(B) < 0.3).astype(f32)
    rows_f, grads_f = run(1, B, vocab, E, Xd, Y)
    rows_g, grads_g = run(0, B, vocab, E, Xd, Y)
    for f in range(2):
        np.testing.assert_array_equal(rows_f[f], rows_g[f])
        np.testing.assert_array_equal(grads_f[f][1], grads_g[f][1])


# ---------------------------------------------------------------------------------------------------------------------------
# multi-hot batches: the segmented two-pass sort (k_bag_scan + k_emb_keys_seg + 2 x (k_seg_hist, k_seg_scatter): the partition by
# field costs no radix pass) against the three-pass radix sort on (field, id) keys it replaces -- ps_tune_set("mh_seg_sort", 0)
# selects the latter.  The two must produce the same sorted pairs, so the same tables bit for bit after training.
MH_CASES = [
    # B, vocab per field, mean bag length, id pattern
    (7, [5, 7], 3, "rand"),                       # a handful of entries: one tile per field, most of it padding
    (300, [50, 1, 2000], 4, "rand"),              # a field of one row (one run of all its entries), empty bags
    (512, [262144, 9], 6, "rand"),                # 18-bit ids: both passes' digits fully used
    (700, [100000, 100000, 33], 30, "zipf"),      # hot keys, fields of ~21 000 entries (6 tiles each)
    (1024, [3000], 160, "zipf"),                  # ONE field of ~164 000 entries: 41 tiles (the scatter's look-back beyond 32 tiles)
    (2048, [100000] * 6, 12, "zipf"),
    (256, [40, 40, 40, 40], 0, "rand"),           # mean 0: most bags empty, some fields with no entry at all
]


def mh_batch(rng, B, vocab, mean_len, pattern):
    F = len(vocab)
    lens = rng.poisson(mean_len, size=B * F) if mean_len > 0 else (rng.random(B * F) < 0.02).astype(np.int64)
    if mean_len == 0:
        lens.reshape(B, F)[:, 1] = 0              # field 1: nothing
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    field = np.repeat(np.tile(np.arange(F), B), lens)
    V = np.asarray(vocab)[field]
    if pattern == "zipf":
        ids = np.minimum(rng.zipf(1.1, field.size) - 1, V - 1).astype(np.int64)
    else:
        ids = (rng.random(field.size) * V).astype(np.int64)
    return ids, offsets


def mh_run(seg, B, vocab, batches, Xd, Y, nnz_max):
    import ps_amd
    from ps_amd import native as N
    N.lib().ps_tune_set(b"mh_seg_sort", seg)
    try:
        F, D = len(vocab), 8
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding(vocab, D)
        kv.set_updater("emF", ps_amd.FtrlUpdater())
        gm = ps_amd.DNN.buildModel(F, D, Xd.shape[1], [16, 1], store=kv, max_batch=B, max_nnz=nnz_max)
        losses = [gm.train(ps_amd.Batch(ids, Xd, Y, None, offsets)) for ids, offsets in batches]
        rows = [kv.get_rows(f, np.arange(vocab[f])) for f in range(F)]
        w = [kv.get("fc%d.weights" % i) for i in range(2)]
        gm.close(); kv.close()
        return losses, rows, w
    finally:
        N.lib().ps_tune_set(b"mh_seg_sort", 1)


@pytest.mark.parametrize("B,vocab,mean_len,pattern", MH_CASES)
def test_segmented_sort_equals_three_pass_sort(B, vocab, mean_len, pattern):
    rng = np.random.default_rng(B * 7 + len(vocab) + mean_len)
    batches = [mh_batch(rng, B, vocab, mean_len, pattern) for _ in range(3)]
    nnz_max = max(max(b[0].size for b in batches), 1)
    Xd = rng.standard_normal((B, 2)).astype(f32)
    Y = (rng.random(B) < 0.3).astype(f32)
    a = mh_run(1, B, vocab, batches, Xd, Y, nnz_max)
    b = mh_run(0, B, vocab, batches, Xd, Y, nnz_max)
    assert a[0] == b[0], (a[0], b[0])
    for f in range(len(vocab)):
        np.testing.assert_array_equal(a[1][f], b[1][f], err_msg="field %d rows" % f)
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    if sum(bt[0].size for bt in batches) > 0:
        assert any(np.abs(r).max() > 0 for r in a[1])
