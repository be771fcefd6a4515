#!/bin/bash
# round 5: the fused step under knob sets / CU masks, interleaved on one box (tools/step_time.py: best of 3 x 500 steps)
#   bash tools/r05_fused_ab.sh <out dir under gpurun_out> <rounds>
O=gpurun_out/$1; mkdir -p $O
for r in $(seq 1 $2); do
  for k in "" "emb_short_grid=1024" "emb_short_grid=2048" "seq_long_grid=1024" "emb_short_grid=1024,seq_long_grid=1024"; do
    printf '%-44s ' "[$k]"; PS_TUNE="$k" python tools/step_time.py 64 2>&1 | tail -1
  done
  for n in 64 96 128; do
    printf '%-44s ' "[PS_CU_MASK_DW=$n]"; PS_CU_MASK_DW=$n python tools/step_time.py 64 2>&1 | tail -1
    printf '%-44s ' "[PS_CU_MASK_DW=$n PS_CU_MASK_MAIN=1]"; PS_CU_MASK_DW=$n PS_CU_MASK_MAIN=1 python tools/step_time.py 64 2>&1 | tail -1
  done
done 2>&1 | tee $O/fused_ab.txt
python tools/gpu_timeline.py 64 > $O/c2_gpu_timeline.txt 2>&1; tail -25 $O/c2_gpu_timeline.txt
PS_CU_MASK_DW=96 PS_CU_MASK_MAIN=1 python tools/gpu_timeline.py 64 > $O/c2_gpu_timeline_cumask96.txt 2>&1; tail -25 $O/c2_gpu_timeline_cumask96.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --gather 0 > $O/bench20.json 2> $O/bench20.err
python - <<PY
import json
d = json.load(open("$O/bench20.json"))
print("20-step line:", d["ms_per_step"], "after ramp", d["after_clock_ramp"]["ms_per_step"], "frac", d["roofline"]["frac"], "multi_hot", d["multi_hot"]["ms_per_step"], d["multi_hot"].get("stream_joins"), "sharded", d["sharded_n1"].get("ms_per_step"))
PY
