import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
def run(kind, M, Nn, K, ns, it=20):
    ms = C.c_double(); N.check(L.ps_bench_gemm(kv.h, kind, M, Nn, K, ns, it, C.byref(ms))); return ms.value * 1e3
for (M, Nn, K) in ((4096, 4096, 4096), (4096, 512, 4096), (8192, 512, 432), (16384, 512, 432), (4096, 512, 432), (4096, 512, 1728)):
    for cfg in (5, 10, 7, 6, 9):
        L.ps_tune_set(b"gemm_nt_cfg", cfg)
        us = run(0, M, Nn, K, 1)
        print("NT cfg%-2d M=%d N=%d K=%d: %.1f us  %.1f TF  %.0f%%" % (cfg, M, Nn, K, us, 2.0 * M * Nn * K / us / 1e6, 100 * 2.0 * M * Nn * K / us / 1e6 / 157.3))
