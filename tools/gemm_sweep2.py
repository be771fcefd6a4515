"""Sweep the tile configurations of k_gemm_nt / k_gemm_tn over the FC shapes of BASELINE configs[1] (measurement only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
def run(kind, M, Nn, K, ns, it=200):
    ms = C.c_double(); N.check(L.ps_bench_gemm(kv.h, kind, M, Nn, K, ns, it, C.byref(ms))); return ms.value * 1e3
B = 4096
NT = [("fwd0", B, 512, 432), ("fwd1", B, 256, 528), ("data1", B, 512, 256), ("data0", B, 416, 512)]
names = {1: "128x128/16", 2: "64x128/16", 3: "64x64/16", 4: "128x32/16", 5: "64x64/32", 6: "64x128/32", 7: "128x128/32", 8: "128x32/32",
         9: "128x64/32", 10: "64x64/64", 11: "128x64w41/32", 12: "64x128w14/32", 13: "128x64 8w", 14: "64x128 8w", 15: "128x128 8w",
         16: "64x32 2w", 17: "32x64 2w"}
cfgs = [int(x) for x in os.environ.get("NT_CFGS", "5,13,14,15,16,17,9,6").split(",")]
for name, M, Nn, K in NT:
    fl = 2.0 * M * Nn * K
    res = []
    for cfg in cfgs:
        L.ps_tune_set(b"gemm_nt_cfg", cfg)
        us = run(0, M, Nn, K, 1)
        res.append((us, cfg))
    res.sort()
    print("NT %-6s M=%d N=%d K=%d : " % (name, M, Nn, K) + "  ".join("%s %.1fus %.0f%%" % (names.get(c, str(c)), us, 100 * fl / us / 1e6 / 157.3) for us, c in res[:6]))
L.ps_tune_set(b"gemm_nt_cfg", 0)
TN = [("dw0", 433, 512), ("dw1", 513, 256)]
for name, K, Nn in TN:
    fl = 2.0 * B * K * Nn
    res = []
    for cfg in (1, 2, 3, 4, 5):
        L.ps_tune_set(b"gemm_tn_cfg", cfg)
        for ns in (2, 4, 6, 8, 10, 14, 20, 28):
            res.append((run(1, B, Nn, K, ns), cfg, ns))
    res.sort()
    print("TN %-4s Kout=%d N=%d M=%d : " % (name, K, Nn, B) + "  ".join("cfg%d/s%d %.1fus %.0f%%" % (c, ns, us, 100 * fl / us / 1e6 / 157.3) for us, c, ns in res[:8]))
