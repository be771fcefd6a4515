#!/bin/bash
# round 5: XCD-affine order of the embedding backward (emb_vblock) -- tests, A/B on the single-hot and the multi-hot step, timelines
O=gpurun_out/$1; mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_schedule.py tests/test_gpu_sumorder.py tests/test_gpu_layer_ops.py tests/test_gpu_operators.py tests/test_gpu_multirank.py tests/test_gpu_rccl_wire.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log; }
for r in 1 2 3; do for k in "" "emb_xcd=2" "emb_xcd=0"; do
  printf '%-14s ' "[$k]"; PS_TUNE="$k" timeout 120 python tools/mh_step.py 200 1 2>&1 | tail -1
  printf '%-14s ' "[$k]"; PS_TUNE="$k" timeout 120 python tools/step_time.py 64 2>&1 | tail -1
done; done | tee $O/ab.txt
MULTI_HOT=1 timeout 200 python tools/gpu_timeline.py 8 > $O/c4_gpu_timeline.txt 2>&1; tail -22 $O/c4_gpu_timeline.txt
timeout 200 python tools/gpu_timeline.py 64 > $O/c2_gpu_timeline.txt 2>&1; tail -16 $O/c2_gpu_timeline.txt
