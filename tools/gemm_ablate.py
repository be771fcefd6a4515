"""Ablation of k_gemm_nt's loop (measurement only): which part of a slab costs what."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
for cfg in (5, 6):
    L.ps_tune_set(b"gemm_nt_cfg", cfg)
    for (M, Nn, K) in ((4096, 512, 432), (4096, 512, 1744), (4096, 4096, 4096)):
        for ab, name in ((0, "full"), (1, "no gload"), (3, "no gload, no ds_write"), (7, "mfma + ds_read only")):
            L.ps_tune_set(b"gemm_ablate", ab)
            ms = C.c_double()
            N.check(L.ps_bench_gemm(kv.h, 0, M, Nn, K, 1, 30, C.byref(ms)))
            print("cfg%d M%d N%d K%-5d %-26s %9.2f us  %6.1f TF/s" % (cfg, M, Nn, K, name, ms.value * 1e3, 2.0 * M * Nn * K / ms.value / 1e9))
L.ps_tune_set(b"gemm_ablate", 0)
