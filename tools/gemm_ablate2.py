"""The FC shapes through k_gemm_nt, 200 back-to-back launches each, for one build of the library (PS_AMD_LIB): used with
tools/gemm_lab_build.sh <bits> to price the parts of a slab (LDS reads, MFMAs, barriers, global loads, LDS writes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_amd
from ps_amd import native as N
kv = ps_amd.KVStore(0, 1); L = N.lib()
cfgs = [int(x) for x in sys.argv[1:]] or [0]
out = []
for cfg in cfgs:
    L.ps_tune_set(b"gemm_nt_cfg", cfg)
    for (M, Nn, K) in ((4096, 512, 432), (4096, 256, 512), (4096, 512, 256), (4096, 416, 512), (4096, 512, 6912)):
        ms = C.c_double()
        N.check(L.ps_bench_gemm(kv.h, 0, M, Nn, K, 1, 200, C.byref(ms)))
        out.append("%7.2f us %5.1f TF" % (ms.value * 1e3, 2.0 * M * Nn * K / ms.value / 1e9))
print("%-28s cfg %s: " % (os.path.basename(os.environ.get("PS_AMD_LIB", "libps_amd.so")), cfgs) + " | ".join(out))
