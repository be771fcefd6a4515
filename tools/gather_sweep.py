"""configs[3]'s gather on the 256 GB table under the multi-hot knobs: row loads in flight per 16-lane group (mh_ilp16) and
non-temporal loads / stores (gather_nt bit 0 / bit 1).  Three rounds per setting, interleaved (one table allocation each)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
rows = 1000 * 1000 * 1000
res = {}
for rnd in range(3):
    for ilp in (1, 2, 4):
        for nt in (1, 3):
            L.ps_tune_set(b"mh_ilp16", ilp); L.ps_tune_set(b"gather_nt", nt)
            for n, bag in ((1 << 17, 32), (1 << 22, 1)):
                if bag == 1 and ilp != 1:
                    continue
                ms, br, bw = C.c_double(), C.c_double(), C.c_double()
                N.check(L.ps_bench_gather(kv.h, rows, 64, n, bag, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw)))
                res.setdefault((bag, ilp, nt), []).append(ms.value * 1e3)
for k in sorted(res):
    bag, ilp, nt = k
    by = (1 << 22) * (264 if bag > 1 else 264)
    print("bag %2d  mh_ilp16 %d  gather_nt %d : %s us   best read frac %.4f" % (bag, ilp, nt, " ".join("%.1f" % x for x in res[k]),
          ((1 << 22) * 264 + (8 * ((1 << 17) + 1) if bag > 1 else 0)) / (min(res[k]) * 1e-6) / 8e12))
