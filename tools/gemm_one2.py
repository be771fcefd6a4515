"""The FC forward GEMM shape and a 16x longer K through k_gemm_nt with round 2's slab loop (gemm_pipe 0) and the pipelined
default, a few launches each -- for a counter pass (tools/pmc_gemm2.sh: MFMA busy, LDS waits, bank conflicts)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
for pipe in (0, 5):
    L.ps_tune_set(b"gemm_pipe", pipe)
    for (M, Nn, K) in ((4096, 512, 432), (4096, 512, 6912)):
        ms = C.c_double()
        N.check(L.ps_bench_gemm(kv.h, 0, M, Nn, K, 1, 10, C.byref(ms)))
        print("gemm_pipe", pipe, M, Nn, K, "%.2f us" % (ms.value * 1e3))
