#!/bin/bash
# where k_gemm_nt_head's workgroups spend their time: a measurement build (-DPS_HD_TIMING), one GPU call, the product build again
set -e
cd "$(dirname "$0")/.."
touch ps_amd/csrc/kernels_gemm.hip
PS_AMD_EXTRA_FLAGS=-DPS_HD_TIMING python -m ps_amd.build > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout 300 python tools/hd_timing.py' 2>&1 | tail -12
touch ps_amd/csrc/kernels_gemm.hip
python -m ps_amd.build > /dev/null
