#!/bin/bash
# round 5: EmbSplit (short keys' reduce + update beside the chunk partials) -- tests, A/B, timeline
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_sumorder.py -m gpu -x -q -k "multi_hot or bags or config4 or hot_keys or emb_backward or sumorder or split" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2; do for k in "" "emb_split=0" "emb_long_grid=512" "emb_long_grid=256"; do printf '%-28s ' "[$k]"; PS_TUNE="$k" timeout 120 python tools/mh_step.py 200 1 2>&1 | tail -1; done; done | tee $O/mh_ab.txt
MULTI_HOT=1 timeout 200 python tools/gpu_timeline.py 8 > $O/c4_gpu_timeline.txt 2>&1; tail -24 $O/c4_gpu_timeline.txt
