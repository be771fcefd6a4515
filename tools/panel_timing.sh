#!/bin/bash
# stamps of k_fwd_panel's phases per workgroup (lab kernel, kernels_panel.hip): build the translation units that look at PS_GEMM_LAB with
# -DPS_GEMM_LAB=1 -DPS_PANEL_TIMING, run tools/panel_timing.py on a GPU box, rebuild the product library
set -e
cd "$(dirname "$0")/.."
T="ps_amd/csrc/kernels_panel.hip ps_amd/csrc/kernels_emb.hip ps_amd/csrc/ps_store.hip ps_amd/csrc/kernels_gemm.hip ps_amd/csrc/ps_ops.hip"
touch $T
PS_AMD_EXTRA_FLAGS="-DPS_PANEL_TIMING -DPS_GEMM_LAB=1" python -m ps_amd.build > /dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- 'python tools/panel_timing.py 2>&1 | tee gpurun_out/panel_timing.txt' 2>&1 | tail -14
touch $T
python -m ps_amd.build > /dev/null
