#!/bin/bash
# gpurun_out/prof_<tag> (written by tools/profile_<tag>.sh on the GPU box) -> profiles/<tag>_*      usage: bash tools/copy_evidence.sh r06
T=${1:-r06}
P=gpurun_out/prof_$T
head -24 $P/trace/c2_kernel_stats.csv > profiles/${T}_c2_bench_kernel_stats.csv
grep "^{" $P/c2_bench_line.json | tail -1 > profiles/${T}_c2_bench_line.json; cp $P/summary.txt profiles/${T}_c2_bench_summary.txt
grep "^{" $P/c2_bench_line_20steps.json | tail -1 > profiles/${T}_c2_bench_line_20steps.json
cp $P/c2_gpu_timeline.txt profiles/${T}_c2_gpu_timeline.txt; cp $P/c4_gpu_timeline.txt profiles/${T}_c4_gpu_timeline.txt
cp $P/pmc_traffic.txt profiles/${T}_pmc_traffic.txt; cp $P/pmc_traffic.json profiles/pmc_traffic.json
grep "passed\|failed" $P/pytest_gpu_full.log > profiles/${T}_pytest_gpu_full.log; grep "passed\|failed" $P/pytest_gemm_lab.log > profiles/${T}_pytest_gemm_lab.log
cp $P/shard_gpu_timeline.txt profiles/${T}_shard_gpu_timeline.txt; grep "^{" $P/shard_n1_line.json | tail -1 > profiles/${T}_shard_n1_line.json
for f in shard_n1_modes_line shard_n1_modes_line_20steps; do grep "^{" $P/$f.json | tail -1 > profiles/${T}_$f.json; done
cp $P/shard_host_timing.txt profiles/${T}_shard_host_timing.txt
for f in rehearse_n8 rehearse_n8_mapped rehearse_c4_n8 rehearse_c4_n8_mapped; do cp $P/$f.log profiles/${T}_$f.log 2>/dev/null; done
if [ -d gpurun_out/adam_$T ]; then
  cp gpurun_out/adam_$T/summary.txt profiles/${T}_c3_adam_summary.txt; cp gpurun_out/adam_$T/trace/adam_kernel_stats.csv profiles/${T}_c3_adam_kernel_stats.csv
  grep "^{" gpurun_out/adam_$T/line.json | tail -1 > profiles/${T}_c3_adam_line.json
fi
