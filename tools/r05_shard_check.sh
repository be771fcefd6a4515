#!/bin/bash
# round 5: the sharded step after a change -- its tests, host timing, GPU-side timeline, A/B of knob sets (one gpurun call)
#   bash tools/r05_shard_check.sh <out dir under gpurun_out> ["knob=v" ...]
O=gpurun_out/$1; shift
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_multirank.py tests/test_gpu_multiproc.py tests/test_gpu_rccl_wire.py tests/test_gpu_parity.py tests/test_gpu_rehearse_n8.py -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
PS_HOST_TIMING=1 timeout 300 python bench.py --sharded --steps 2000 --no-cpu --gather 0 --multi-hot 0 > $O/shard_n1_line.json 2> $O/shard.err; grep "host:" $O/shard.err | head -3
python - <<PY
import json
d = json.load(open("$O/shard_n1_line.json")); print(d["ms_per_step"], d.get("wire_cost_ms_per_step"))
PY
PS_STAMPS=$O/shard_stamps.json timeout 200 python bench.py --sharded --wire-cost 0 --steps 300 --no-cpu --gather 0 --multi-hot 0 > /dev/null 2>&1
python tools/shard_timeline.py $O/shard_stamps.json > $O/shard_gpu_timeline.txt 2>&1; rm -f $O/shard_stamps.json; cat $O/shard_gpu_timeline.txt
if [ $# -gt 0 ]; then STEPS=1500 timeout 500 bash tools/shard_ab.sh 2 "" "$@" 2>&1 | tee $O/ab.txt; fi
# the driver's view: the 20-step sharded_n1 leg of the default bench line (child process), under the same knob sets
for r in 1 2; do for k in "" "$@"; do printf '%-24s ' "[leg20 $k]"; PS_TUNE="$k" timeout 200 python bench.py --leg sharded_n1 --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('collective_device_us'))"; done; done 2>&1 | tee $O/leg20.txt
