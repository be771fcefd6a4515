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
