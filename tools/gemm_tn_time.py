"""The dW GEMMs (k_gemm_tn, split over the batch) alone, 200 back-to-back launches per config: dW0 (430 x 512 over 4096 rows,
4 splits), dW1 (513 x 256, 7 splits), and a long one (430 x 512 over 65536 rows, 4 splits) for the steady state."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
for cfg in [int(x) for x in sys.argv[1:]] or [0]:
    L.ps_tune_set(b"gemm_tn_cfg", cfg)
    out = []
    for (K, Nn, M, ns) in ((430, 512, 4096, 4), (513, 256, 4096, 7), (430, 512, 65536, 4)):
        ms = C.c_double()
        N.check(L.ps_bench_gemm(kv.h, 1, M, Nn, K, ns, 200, C.byref(ms)))
        out.append("%7.2f us %5.1f TF" % (ms.value * 1e3, 2.0 * M * Nn * K / ms.value / 1e9))
    print("tn cfg %2d: " % cfg + " | ".join(out))
