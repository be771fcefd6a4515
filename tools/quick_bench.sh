#!/bin/bash
# quick look at the fused step on the GPU box: ms/step and the per-group kernel times (no CPU baseline, no big gather)
mkdir -p gpurun_out/r2
python bench.py --steps ${1:-300} --warmup 30 --no-cpu --gather 0 --multi-hot ${2:-0} ${3:-} > gpurun_out/r2/quick.json 2> gpurun_out/r2/quick.err || tail -5 gpurun_out/r2/quick.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/quick.json"))
print("ms/step %.4f  ex/s %.3e  roofline %s %.3f (%.1f us)" % (d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"]))
g = d["kernel_groups_us"]
print("  ".join("%s=%.1f" % kv for kv in sorted(g.items())))
print("sum of groups %.1f us" % sum(g.values()))
if "multi_hot" in d: print(d["multi_hot"])
PY
