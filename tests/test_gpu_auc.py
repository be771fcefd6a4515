"""ps_auc_compute (device: stable radix sort of the float bits + exact pair count) against the restated
evaluate/AUC.java on the reference's own TestAuc vector, ties, degenerate label sets, and a large case."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "auc_testauc.npz")
TOL = 1e-12          # the exact pair count / (P*N) vs the reference's running double sum


def test_testauc_vector_on_device():
    import ps_amd
    z = np.load(GOLD)
    kv = ps_amd.KVStore(0, 1)
    a = ps_amd.AUC(z["p"], z["y"], store=kv)
    got = a.calculate()
    assert abs(got - z["expected"][0]) <= TOL
    assert a.posNum == int((z["y"] > 0).sum()) and a.negNum == 1000 - a.posNum
    kv.close()


@pytest.mark.parametrize("n,quant", [(1, 0), (2, 0), (257, 2), (5000, 2), (5000, 0), (70001, 3)])
def test_against_restatement(orc, n, quant):
    import ps_amd
    rng = np.random.default_rng(n)
    p = rng.random(n).astype(np.float32)
    if quant:
        p = np.round(p, quant).astype(np.float32)                # ties: resolved by input order (stable sort)
    if n > 2:
        p[::7] *= -1                                              # negative scores order below positive ones
    y = (rng.random(n) < 0.3).astype(np.float32)
    kv = ps_amd.KVStore(0, 1)
    got, want = ps_amd.AUC(p, y, store=kv).calculate(), orc.auc(p, y)
    assert (np.isnan(got) and np.isnan(want)) or abs(got - want) <= TOL, (got, want)
    kv.close()


def test_degenerate_and_device_resident(orc):
    import ps_amd
    from ps_amd import native as N
    kv = ps_amd.KVStore(0, 1)
    assert ps_amd.AUC([0.2, 0.3], [1, 1], store=kv).calculate() == 0.0
    assert np.isnan(ps_amd.AUC([0.2, 0.3], [0, 0], store=kv).calculate())
    assert ps_amd.AUC([0.5, 0.5], [1, 0], store=kv).calculate() == 0.0
    assert ps_amd.AUC([0.5, 0.5], [0, 1], store=kv).calculate() == 1.0
    # device-resident p / y (what a predict loop leaves in HBM)
    rng = np.random.default_rng(5)
    n = 1 << 20
    p = rng.random(n).astype(np.float32); y = (rng.random(n) < 0.25).astype(np.float32)
    dp, dy = C.c_void_p(), C.c_void_p()
    for d, a in ((dp, p), (dy, y)):
        N.check(N.lib().ps_dev_alloc(kv.h, a.nbytes, C.byref(d))); N.check(N.lib().ps_dev_upload(kv.h, d, a.ctypes.data, a.nbytes))
    got = ps_amd.AUC(dp, dy, store=kv, n=n, on_device=True).calculate()
    # closed form on the host: mean rank of the positives (no ties at this size matter beyond 1e-6)
    order = np.argsort(p, kind="stable"); r = np.empty(n, np.int64); r[order] = np.arange(n)
    P = int((y > 0).sum()); Nn = n - P
    want = (float(r[y > 0].sum()) - P * (P - 1) / 2) / (float(P) * Nn)
    assert abs(got - want) <= 1e-12
    N.lib().ps_dev_free(kv.h, dp); N.lib().ps_dev_free(kv.h, dy)
    kv.close()


@pytest.mark.parametrize("n,levels", [(6_000_000, 0), (4_300_000, 1000)])
def test_sort_above_one_superblock_level(n, levels):
    """> 4.2 M pairs = more than 32 superblocks of 32 tiles: the radix scatter's prefix then also walks the loop over the
    superblock sums beyond the first 32 (kernels_sort.hip), 4 passes of 32-bit keys; exact against the stable-rank form."""
    import ps_amd
    from ps_amd import native as N
    kv = ps_amd.KVStore(0, 1)
    rng = np.random.default_rng(n)
    p = rng.random(n).astype(np.float32)
    if levels:
        p = (np.floor(p * levels) / levels).astype(np.float32)         # long runs of equal keys
    y = (rng.random(n) < 0.3).astype(np.float32)
    dp, dy = C.c_void_p(), C.c_void_p()
    for d, a in ((dp, p), (dy, y)):
        N.check(N.lib().ps_dev_alloc(kv.h, a.nbytes, C.byref(d))); N.check(N.lib().ps_dev_upload(kv.h, d, a.ctypes.data, a.nbytes))
    got = ps_amd.AUC(dp, dy, store=kv, n=n, on_device=True).calculate()
    order = np.argsort(p, kind="stable"); r = np.empty(n, np.int64); r[order] = np.arange(n)
    P = int((y > 0).sum()); Nn = n - P
    want = (float(r[y > 0].sum()) - P * (P - 1) / 2) / (float(P) * Nn)
    assert abs(got - want) <= 1e-12
    N.lib().ps_dev_free(kv.h, dp); N.lib().ps_dev_free(kv.h, dy)
    kv.close()
