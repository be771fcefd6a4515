#!/bin/bash
# Measurement builds of the library with parts of k_gemm_nt's slab loop compiled out (PS_GEMM_ABLATE bits, kernels_gemm.hip):
#   bash tools/gemm_ablate_build.sh 1 2 3 4 ...   ->  ps_amd/lib/libps_amd_ab<N>.so   (run here; the .so files travel to the GPU box)
# then on the box:  for n in 0 1 2 ...; do PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_ab$n.so python tools/gemm_ablate2.py; done
set -e
cd "$(dirname "$0")/.."
python -m ps_amd.build >/dev/null
objs=$(ls ps_amd/build/*.o | grep -v kernels_gemm | grep -v _ab)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPS_GEMM_ABLATE=$n -c ps_amd/csrc/kernels_gemm.hip -o ps_amd/build/kernels_gemm_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ps_amd/lib/libps_amd_ab$n.so $objs ps_amd/build/kernels_gemm_ab$n.o -ldl
  echo ps_amd/lib/libps_amd_ab$n.so
done
