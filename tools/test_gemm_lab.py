"""The rejected GEMM variants of the LAB build (kernels_gemm.hip, PS_GEMM_LAB; tools/gemm_lab_build.sh) against the plain loop on
the same tiles -- moved out of the product's test suite with the kernels themselves.
    PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_lab.so python -m pytest tools/test_gemm_lab.py -m gpu -q"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_schedule import batches, run      # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lab_only():
    from ps_amd import native as N
    if b"+gemm_lab" not in N.lib().ps_version():
        pytest.skip("not the lab build: PS_AMD_LIB=ps_amd/lib/libps_amd_lab.so (tools/gemm_lab_build.sh)")


@pytest.mark.parametrize("knob,plain,piped", [("gemm_nt_cfg", 5, 45), ("gemm_nt_cfg", 5, 85), ("gemm_nt_cfg", 5, 105), ("gemm_nt_cfg", 6, 46),
                                              ("gemm_nt_cfg", 7, 47), ("gemm_nt_cfg", 8, 48), ("gemm_nt_cfg", 13, 113), ("gemm_nt_cfg", 20, 120), ("gemm_nt_cfg", 5, 125), ("gemm_nt_cfg", 13, 133), ("gemm_nt_cfg", 3, 140), ("gemm_nt_cfg", 3, 141),
                                              ("gemm_tn_cfg", 2, 12), ("gemm_tn_cfg", 6, 16), ("gemm_tn_cfg", 7, 17),
                                              ("gemm_tn_cfg", 2, 22), ("gemm_tn_cfg", 6, 26), ("gemm_tn_cfg", 7, 27)])
def test_pipelined_gemm_loops_are_bit_identical(knob, plain, piped):
    """The software-pipelined slab loops of k_gemm_nt / k_gemm_tn (fragment prefetch, three LDS buffers and register sets,
    LDS writes dealt out over the slab) multiply the same products in the same order per accumulator as the plain loop on the
    same tiles: six training steps leave identical tables, on shapes with ragged M, N and K."""
    for kind, F, D, X, fc, V, B in (("widedeep", 6, 16, 5, [64, 32, 1], 3000, 2048), ("dnn", 9, 8, 1, [130, 70, 1], 40, 1000),
                                    ("dnn", 3, 4, 2, [5, 3, 1], 7, 6)):
        rng = np.random.default_rng(F * 100 + B)
        WS = 97
        data = batches(rng, 6, B, F, X, V, WS)
        ref = run(kind, {knob: plain}, False, data, F, D, X, fc, V, B, WS)
        got = run(kind, {knob: piped}, False, data, F, D, X, fc, V, B, WS)
        assert got[0] == ref[0], "%s %d vs %d: losses %s vs %s" % (knob, plain, piped, got[0], ref[0])
        for a, b in zip(ref[1:], got[1:]):
            if isinstance(a, list):
                for x, y in zip(a, b):
                    np.testing.assert_array_equal(x, y)
            else:
                np.testing.assert_array_equal(a, b)




def _close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    tol = 2e-5 * (np.abs(b) + np.sqrt(np.mean(b * b)) + 1e-30)
    bad = np.abs(a - b) > tol
    assert not bad.any(), "%s: %d of %d beyond 2e-5 (|x| + rms), worst %.3g at |x| = %.3g" % (what, bad.sum(), bad.size, np.abs(a - b).max(), np.abs(b).flat[np.argmax(np.abs(a - b))])


def test_row_panel_forward_matches_the_gemm_launches():
    """k_fwd_panel (kernels_panel.hip; ps_tune_set("fwd_panel")): FcLayer.forward x 2 of a 16-row panel in one launch at configs[1]'s FC
    shape (429 -> 512 -> 256 -> 1), weights streamed in fragment order (Wp, written by k_dense_update beside W' and Wt).  With and
    without the head in the launch the step is bit-identical (the head's arithmetic is one function: kernels_head.inc head_one_t);
    against the k_gemm_nt launches the products are summed in another order: tables, losses and P agree to f32 rounding over three
    training steps, on a batch whose last panel is ragged."""
    kind, F, D, X, fc, V, WS = "widedeep", 26, 16, 13, [512, 256, 1], 500, 997
    for B in (4096, 1000):
        rng = np.random.default_rng(B)
        data = batches(rng, 3, B, F, X, V, WS)
        gemm = run(kind, {"fwd_panel": 0}, False, data, F, D, X, fc, V, B, WS)
        p1 = run(kind, {"fwd_panel": 1}, False, data, F, D, X, fc, V, B, WS)
        p2 = run(kind, {"fwd_panel": 2}, False, data, F, D, X, fc, V, B, WS)
        assert p1[0] == p2[0], "losses with / without the head in the launch: %s vs %s" % (p1[0], p2[0])
        for a, b in zip(p1[1:], p2[1:]):
            for x, y in zip(a if isinstance(a, list) else [a], b if isinstance(b, list) else [b]):
                np.testing.assert_array_equal(x, y)
        flat = lambda o: [np.asarray(x) for a in o[1:] for x in (a if isinstance(a, list) else [a])]
        assert any(not np.array_equal(x, y) for x, y in zip(flat(p2), flat(gemm))), "the panel kernel did not run (every table has the GEMM launches' bits)"
        _close(p2[0], gemm[0], "losses")
        for i, (a, b) in enumerate(zip(p2[1:], gemm[1:])):
            for j, (x, y) in enumerate(zip(a if isinstance(a, list) else [a], b if isinstance(b, list) else [b])):
                _close(x, y, "output %d.%d at B = %d" % (i, j, B))
