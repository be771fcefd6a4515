"""BASELINE config 4 gather roofline at full size: one table of 1e9 rows x 64 f32 (256 GB in 288 GB HBM),
2^22 random ids per launch (measurement only; weights-only table, filled on device)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
out = []
for rows in [int(x) for x in (sys.argv[1:] or ["64000000", "1000000000"])]:
    for n, bag in ((1 << 22, 1), (1 << 17, 32)):
        ms, br, bw = C.c_double(), C.c_double(), C.c_double()
        rc = L.ps_bench_gather(kv.h, rows, 64, n, bag, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw))
        if rc != 0:
            out.append({"rows": rows, "error": L.ps_last_error().decode()}); break
        out.append({"rows": rows, "table_GB": rows * 256 / 1e9, "lookups": n * bag, "bag": bag, "avg_us": round(ms.value * 1e3, 1),
                    "read_GBs": round(br.value / ms.value / 1e6, 1), "read_plus_write_GBs": round((br.value + bw.value) / ms.value / 1e6, 1)})
print(json.dumps(out))
