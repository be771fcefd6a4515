#!/bin/bash
# A/B of two builds of libps_amd.so on ONE GPU box (boxes differ by several percent, runs on one box by ~1%):
#   cp ps_amd/lib/libps_amd.so ps_amd/lib/libps_amd_A.so   # before the change
#   ... edit, python -m ps_amd.build ...
#   gpurun -- 'bash tools/ab_bench.sh [steps] [rounds]'
# prints ms/step of A (libps_amd_A.so) and B (libps_amd.so), interleaved
steps=${1:-1000}; rounds=${2:-3}
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then export PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_A.so; else unset PS_AMD_LIB; fi
    python bench.py --steps $steps --warmup 50 --no-cpu --gather 0 --multi-hot 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['kernel_groups_us']
print('$v ms/step %.4f  head %.1f emb_bwd %.1f dense %.1f' % (d['ms_per_step'], g.get('head_last_bwd',0), g['emb_bwd_update'], g['dense_update']))"
  done
done
