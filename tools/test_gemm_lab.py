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


