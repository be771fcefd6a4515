#!/bin/bash
# The MEASUREMENT build of the library: every GEMM tile shape / slab loop / kernel that rounds 2-3 built, measured and rejected
# (kernels_gemm.hip, PS_GEMM_LAB) compiled in, selectable by ps_tune_set("gemm_nt_cfg" / "gemm_tn_cfg" / "gemm_pipe" /
# "gemm_8w" / "gemm_ks" / "fwd_pair").  The product library (python -m ps_amd.build) has none of them.
#   bash tools/gemm_lab_build.sh [ablate bits ...]    ->  ps_amd/lib/libps_amd_lab.so  (+ libps_amd_lab_ab<N>.so with PS_GEMM_ABLATE=N)
# then on the GPU box:  PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_lab.so python -m pytest tools/test_gemm_lab.py -m gpu -q
#                       PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_lab.so python tools/gemm_sweep2.py
set -e
cd "$(dirname "$0")/.."
python -m ps_amd.build >/dev/null
mkdir -p ps_amd/build_lab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPS_GEMM_LAB=1 -Ips_amd/csrc ${PS_AMD_EXTRA_FLAGS:-}"
# only the translation units that look at PS_GEMM_LAB are rebuilt (kernels_emb / ps_ops: the LDS-staged gather; lab/kernels_panel, ps_store: the
# row-panel forward and its fragment-order weights); the rest are the product's objects
for src in kernels_gemm ps_store kernels_emb ps_ops; do
  /opt/rocm/bin/hipcc $FLAGS -c ps_amd/csrc/$src.hip -o ps_amd/build_lab/$src.o
done
/opt/rocm/bin/hipcc $FLAGS -c ps_amd/csrc/lab/kernels_panel.hip -o ps_amd/build_lab/kernels_panel.o      # (the lab-only sources live in csrc/lab/)
objs=$(ls ps_amd/build/*.o | grep -v "/kernels_gemm.o" | grep -v "/ps_store.o" | grep -v "/kernels_emb.o" | grep -v "/ps_ops.o" | grep -v "/kernels_panel.o" | grep -v _ab)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ps_amd/lib/libps_amd_lab.so $objs ps_amd/build_lab/kernels_gemm.o ps_amd/build_lab/ps_store.o ps_amd/build_lab/kernels_emb.o ps_amd/build_lab/ps_ops.o ps_amd/build_lab/kernels_panel.o -ldl
echo ps_amd/lib/libps_amd_lab.so
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DPS_GEMM_ABLATE=$n -c ps_amd/csrc/kernels_gemm.hip -o ps_amd/build_lab/kernels_gemm_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ps_amd/lib/libps_amd_lab_ab$n.so $objs ps_amd/build_lab/kernels_gemm_ab$n.o ps_amd/build_lab/ps_store.o ps_amd/build_lab/kernels_emb.o ps_amd/build_lab/ps_ops.o ps_amd/build_lab/kernels_panel.o -ldl
  echo ps_amd/lib/libps_amd_lab_ab$n.so
done
