#!/bin/bash
# start / end of every workgroup of k_emb_reduce_update: build kernels_emb.hip with -DPS_EMB_TIMING, run tools/emb_timing.py on a GPU box
# (both shapes), rebuild the product library.   bash tools/emb_timing.sh ["knob=v,..."]
set -e
cd "$(dirname "$0")/.."
touch ps_amd/csrc/kernels_emb.hip
PS_AMD_EXTRA_FLAGS=-DPS_EMB_TIMING python -m ps_amd.build > /dev/null 2>&1
/usr/local/graft/bin/gpurun --timeout 600 -- "PS_TUNE=$1 MULTI_HOT=1 python tools/emb_timing.py 2>&1 | tee gpurun_out/emb_timing_mh.txt; PS_TUNE=$1 python tools/emb_timing.py 2>&1 | tee gpurun_out/emb_timing_sh.txt" 2>&1 | grep -v '^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl' | tail -34
touch ps_amd/csrc/kernels_emb.hip
python -m ps_amd.build > /dev/null 2>&1
