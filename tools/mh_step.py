"""Measurement only: the multi-hot step (configs[4]'s shape on one GPU, bench.py's multi_hot leg) alone.
python tools/mh_step.py [steps] [repeats]   (PS_TUNE=knob=v,... and PS_AMD_LIB=<other build> are honoured)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ps_amd import native as N

for kv_ in os.environ.get("PS_TUNE", "").split(","):
    if "=" in kv_:
        N.lib().ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    r = bench.multi_hot_step(dict(bench.C2), steps)
    print("multi-hot %.4f ms/step  %.2f G ids/s  loss %.4f" % (r["ms_per_step"], r["ids_per_s"] * 1e-9, r["final_loss"]))
