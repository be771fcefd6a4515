#!/bin/bash
set -e
cd "$(dirname "$0")/.."
touch ps_amd/csrc/kernels_sort.hip
PS_AMD_EXTRA_FLAGS=-DPS_FS_TIMING python -m ps_amd.build > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'python tools/fs_timing.py' 2>&1 | tail -7
touch ps_amd/csrc/kernels_sort.hip
python -m ps_amd.build > /dev/null
