#!/bin/bash
# A/B of ps_tune_set knob sets on ONE GPU box, interleaved: bash tools/ab_knobs.sh <rounds> "" "tail_fused=0" "tail_fused=0,tn_start_wait=0" ...
# (tools/step_time.py: 64 rotating batches, best of 3 x 500 steps per run)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for k in "$@"; do
    printf '%-40s ' "[$k]"
    PS_TUNE="$k" python tools/step_time.py 64 2>&1 | tail -1
  done
done
