#!/bin/bash
# round 5: what short timed regions cost -- the multi-hot leg (60 steps) against long runs, presort on / off; the driver's 20-step line
O=gpurun_out/$1; mkdir -p $O
for r in 1 2; do for k in "mh_presort=3" "mh_presort=0"; do for n in 60 200; do printf '%-18s %4d steps: ' "[$k]" $n; PS_TUNE="$k" python tools/mh_step.py $n 1 2>&1 | tail -1; done; done; done | tee $O/mh_short.txt
for r in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --gather 0 --sharded-leg 0 --clamped 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20-step line', d['ms_per_step'], 'from idle', d['from_idle_queue']['ms_per_step'], 'after', d['after_clock_ramp']['ms_per_step'], 'multi-hot leg', d['multi_hot']['ms_per_step'])"; done | tee $O/bench20.txt
