"""GEMM micro-benchmark sweep over the FC shapes of BASELINE configs[1] (measurement only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N

kv = ps_amd.KVStore(0, 1)
L = N.lib()
NT = {5: "64x64/32", 6: "64x128/32", 9: "128x64/32", 10: "64x64/64", 11: "128x64w41", 12: "64x128w14", 8: "128x32/32"}
TN = {1: "64x64/16", 2: "64x64/32", 3: "128x128/16"}
shapes = [("fwd0", 4096, 512, 432), ("fwd1", 4096, 256, 528), ("bwd_data0", 4096, 416, 512),
          ("bwd_data1", 4096, 512, 256), ("big", 4096, 4096, 4096)]
quick = "--quick" in sys.argv
for xcd in (1,):
    L.ps_tune_set(b"gemm_xcd", xcd)
    for name, M, Nn, K in shapes:
        for cfg in ([5] if quick else NT):
            L.ps_tune_set(b"gemm_nt_cfg", cfg)
            ms = C.c_double()
            N.check(L.ps_bench_gemm(kv.h, 0, M, Nn, K, 1, 50, C.byref(ms)))
            print("xcd%d %-10s %-11s M%d N%d K%d  %8.2f us  %6.1f TF/s" % (xcd, name, NT[cfg], M, Nn, K, ms.value * 1e3, 2.0 * M * Nn * K / ms.value / 1e9))
    L.ps_tune_set(b"gemm_nt_cfg", 0)
    for name, M, Nn, K in (("dW0", 4096, 512, 430), ("dW1", 4096, 256, 513)):
        for cfg in ([1, 2] if quick else TN):
            L.ps_tune_set(b"gemm_tn_cfg", cfg)
            for ns in (8, 16, 32):
                ms = C.c_double()
                N.check(L.ps_bench_gemm(kv.h, 1, M, Nn, K, ns, 50, C.byref(ms)))
                print("xcd%d %-10s %-11s split%-3d %8.2f us  %6.1f TF/s" % (xcd, name, TN[cfg], ns, ms.value * 1e3, 2.0 * M * Nn * K / ms.value / 1e9))
    L.ps_tune_set(b"gemm_tn_cfg", 0)
