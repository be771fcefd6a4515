#!/bin/bash
set -e
cd "$(dirname "$0")/.."
touch ps_amd/csrc/kernels_emb.hip
PS_AMD_EXTRA_FLAGS=-DPS_HEAD_TIMING python -m ps_amd.build > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'python tools/head_timing.py' 2>&1 | tail -4
touch ps_amd/csrc/kernels_emb.hip
python -m ps_amd.build > /dev/null
