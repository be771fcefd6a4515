#!/bin/bash
# stamps of k_fwd_panel's phases per workgroup (lab kernel, csrc/lab/kernels_panel.hip): the lab library with -DPS_PANEL_TIMING
# (tools/gemm_lab_build.sh), tools/panel_timing.py on a GPU box against it, then the lab library without the stamps again
set -e
cd "$(dirname "$0")/.."
PS_AMD_EXTRA_FLAGS="-DPS_PANEL_TIMING" bash tools/gemm_lab_build.sh > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_lab.so python tools/panel_timing.py 2>&1 | tee gpurun_out/panel_timing.txt' 2>&1 | tail -14
bash tools/gemm_lab_build.sh > /dev/null
