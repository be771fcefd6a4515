import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1)
for ilp in (1, 2, 4):
    N.check(N.lib().ps_tune_set(b"mh_ilp16", ilp))
    for bag in (32, 8):
        ms, br, bw = C.c_double(), C.c_double(), C.c_double()
        N.check(N.lib().ps_bench_gather(kv.h, 1000 * 1000 * 1000, 64, (1 << 22) // bag, bag, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw)))
        print("ilp %d bag %2d: %.1f us  read %.0f GB/s" % (ilp, bag, ms.value * 1e3, br.value / ms.value / 1e6))
