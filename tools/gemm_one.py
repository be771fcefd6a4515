"""Run a few GEMM shapes once each config (for PMC passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
for cfg, (M, Nn, K) in ((5, (4096, 512, 432)), (5, (4096, 4096, 4096)), (7, (4096, 4096, 4096))):
    L.ps_tune_set(b"gemm_nt_cfg", cfg)
    ms = C.c_double()
    N.check(L.ps_bench_gemm(kv.h, 0, M, Nn, K, 1, 10, C.byref(ms)))
    print(cfg, M, Nn, K, ms.value * 1e3, "us")
