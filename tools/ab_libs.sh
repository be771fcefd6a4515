#!/bin/bash
# compare several builds of the library on ONE GPU box: bash tools/ab_libs.sh <steps> <rounds> libA.so libB.so ...
steps=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for l in "$@"; do
    PS_AMD_LIB=$PWD/$l python bench.py --steps $steps --warmup 50 --no-cpu --gather 0 --multi-hot 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['kernel_groups_us']
print('%-34s ms/step %.4f  head %.1f emb_bwd %.1f dense %.1f' % ('$l', d['ms_per_step'], g.get('head_last_bwd',0), g['emb_bwd_update'], g['dense_update']))"
  done
done
