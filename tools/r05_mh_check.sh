#!/bin/bash
# round 5: the multi-hot step after a change -- its tests, the step under knob sets (tools/mh_step.py), its GPU-side timeline
#   bash tools/r05_mh_check.sh <out dir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_parity.py tests/test_gpu_fieldsort.py tests/test_gpu_sumorder.py tests/test_gpu_layer_ops.py tests/test_gpu_operators.py tests/test_gpu_configs.py tests/test_gpu_auc.py tests/test_gpu_ps_server.py -m gpu -x -q -k "multi_hot or bags or segment or sort or config4 or hot_keys or emb_backward or auc or server" > $O/pytest.log 2>&1; [ -n "$SKIP_TESTS" ] || tail -5 $O/pytest.log
for r in 1 2; do for k in "" "keys_grid=256" "mh_presort=0"; do printf '%-28s ' "[$k]"; PS_TUNE="$k" python tools/mh_step.py 200 1 2>&1 | tail -1; done; done | tee $O/mh_ab.txt
MULTI_HOT=1 python tools/gpu_timeline.py 8 > $O/c4_gpu_timeline.txt 2>&1; tail -32 $O/c4_gpu_timeline.txt
