#!/bin/bash
# round 5: the sharded N = 1 step through RCCL under environment sets (what of RCCL's +36 us per step is configuration?)
#   bash tools/r05_rccl_env.sh <out dir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
run() {  # label, env assignments...
  local label="$1"; shift
  printf '%-60s ' "[$label]"
  env "$@" timeout 300 python bench.py --leg sharded_n1 --steps 1000 2>/dev/null | python -c '
import json, sys
d = json.loads(sys.stdin.readline())
c = d["collective_device_us"]["rccl_with_own_keys_in_place"]
print("copies %.4f  rccl %.4f  | ids %.1f rows %.1f grads %.1f allreduce %.1f" % (d["ms_per_step"]["device_copies"], d["ms_per_step"]["rccl_with_own_keys_in_place"], c["id_blocks"], c["rows"], c["gradients"], c["allreduce"]))' 2>&1 | tail -1
}
if [ -n "$SWEEP2" ]; then
for r in 1 2 3; do
  run "baseline" PS_X=0
  run "NCCL_NCHANNELS_PER_PEER=2" NCCL_NCHANNELS_PER_PEER=2
  run "NCCL_NCHANNELS_PER_PEER=4" NCCL_NCHANNELS_PER_PEER=4
  run "NCCL_NCHANNELS_PER_PEER=8" NCCL_NCHANNELS_PER_PEER=8
  run "NCCL_MIN_NCHANNELS=16" NCCL_MIN_NCHANNELS=16
  run "NCCL_MIN_NCHANNELS=32" NCCL_MIN_NCHANNELS=32
  run "NCCL_MIN_NCHANNELS=64" NCCL_MIN_NCHANNELS=64
  run "NCCL_MIN_NCHANNELS=32 NCCL_NCHANNELS_PER_PEER=8" NCCL_MIN_NCHANNELS=32 NCCL_NCHANNELS_PER_PEER=8
  run "NCCL_DEBUG=VERSION" NCCL_DEBUG=VERSION
done 2>&1 | tee $O/rccl_env2.txt
exit 0
fi
for r in 1 2; do
  run "baseline" PS_X=0
  run "HSA_NO_SCRATCH_RECLAIM=1" HSA_NO_SCRATCH_RECLAIM=1
  run "NCCL_MAX_NCHANNELS=4" NCCL_MAX_NCHANNELS=4
  run "NCCL_MAX_NCHANNELS=2" NCCL_MAX_NCHANNELS=2
  run "NCCL_MIN_NCHANNELS=1 NCCL_MAX_NCHANNELS=1" NCCL_MIN_NCHANNELS=1 NCCL_MAX_NCHANNELS=1
  run "NCCL_NCHANNELS_PER_PEER=1" NCCL_NCHANNELS_PER_PEER=1
  run "RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0" RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
  run "NCCL_PROTO=LL" NCCL_PROTO=LL
  run "NCCL_BUFFSIZE=1048576" NCCL_BUFFSIZE=1048576
  run "HSA_NO_SCRATCH_RECLAIM=1 NCCL_MAX_NCHANNELS=4" HSA_NO_SCRATCH_RECLAIM=1 NCCL_MAX_NCHANNELS=4
done 2>&1 | tee $O/rccl_env.txt
