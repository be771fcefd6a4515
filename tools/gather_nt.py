"""configs[3] gather on the 256 GB table: non-temporal loads / stores on or off (measurement)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000 * 1000 * 1000
for n, bag in ((1 << 22, 1), (1 << 17, 32)):
    for nt in (0, 1, 2, 3):
        L.ps_tune_set(b"gather_nt", nt)
        ms, br, bw = C.c_double(), C.c_double(), C.c_double()
        N.check(L.ps_bench_gather(kv.h, rows, 64, n, bag, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw)))
        print("bag %2d nt=%d: %.1f us  read %.0f GB/s (%.3f of 8 TB/s)  read+write %.0f GB/s (%.3f)" % (
            bag, nt, ms.value * 1e3, br.value / ms.value / 1e6, br.value / ms.value / 1e6 / 8000, (br.value + bw.value) / ms.value / 1e6, (br.value + bw.value) / ms.value / 1e6 / 8000))
L.ps_tune_set(b"gather_nt", -1)
