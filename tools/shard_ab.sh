#!/bin/bash
# the sharded N = 1 step under several knob sets, interleaved on one box: bash tools/shard_ab.sh <rounds> "" "gemm_pipe=0" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do for k in "$@"; do printf '%-24s ' "[$k]"; PS_TUNE="$k" python bench.py --sharded --steps ${STEPS:-1500} --no-cpu --gather 0 --multi-hot 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
