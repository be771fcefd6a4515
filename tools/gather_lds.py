"""The LDS-staged single-hot gather (global_load_lds) vs plain 16-byte vector loads on the 256 GB table: checks the
staged kernel's outputs against the table's definition, then times both (measurement only)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ps_amd
from ps_amd import native as N
from test_gpu_configs import table_rows
kv = ps_amd.KVStore(0, 1); L = N.lib()
rows, D, n, ns, seed = 1000 * 1000 * 1000, 64, 1 << 22, 1024, 0x5EED
for lds in (1, 0):
    L.ps_tune_set(b"gather_lds", lds)
    bi = np.zeros(ns, np.int64); ids = np.zeros((ns, 1), np.int64); out = np.zeros((ns, D), np.float32)
    N.check(L.ps_bench_gather_check(kv.h, rows, D, n, 1, seed, ns, bi.ctypes.data_as(C.POINTER(C.c_int64)),
                                    ids.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_float))))
    ok = np.array_equal(out, np.maximum(table_rows(seed, ids[:, 0], D), 0))
    ms, br, bw = C.c_double(), C.c_double(), C.c_double()
    N.check(L.ps_bench_gather(kv.h, rows, D, n, 1, 20, seed, C.byref(ms), C.byref(br), C.byref(bw)))
    print("gather_lds=%d: outputs %s, %.1f us, read %.3f, read+write %.3f of 8 TB/s" % (
        lds, "bit-exact" if ok else "WRONG", ms.value * 1e3, br.value / ms.value / 1e6 / 8000, (br.value + bw.value) / ms.value / 1e6 / 8000))
L.ps_tune_set(b"gather_lds", 0)
