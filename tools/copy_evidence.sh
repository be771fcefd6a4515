#!/bin/bash
# gpurun_out/prof_<tag> (written by tools/profile_<tag>.sh on the GPU box) -> profiles/<tag>_*      usage: bash tools/copy_evidence.sh r05
T=${1:-r05}
P=gpurun_out/prof_$T
head -22 $P/trace/c2_kernel_stats.csv > profiles/${T}_c2_bench_kernel_stats.csv
cp $P/c2_bench_line.json profiles/${T}_c2_bench_line.json; cp $P/summary.txt profiles/${T}_c2_bench_summary.txt
cp $P/c2_bench_line_20steps.json profiles/${T}_c2_bench_line_20steps.json 2>/dev/null
cp $P/c2_gpu_timeline.txt profiles/${T}_c2_gpu_timeline.txt; cp $P/c4_gpu_timeline.txt profiles/${T}_c4_gpu_timeline.txt
cp $P/pmc_traffic.txt profiles/${T}_pmc_traffic.txt; cp $P/pmc_traffic.json profiles/pmc_traffic.json
grep "passed\|failed" $P/pytest_gpu_full.log > profiles/${T}_pytest_gpu_full.log; grep "passed\|failed" $P/pytest_gemm_lab.log > profiles/${T}_pytest_gemm_lab.log
cp $P/shard_gpu_timeline.txt profiles/${T}_shard_gpu_timeline.txt; cp $P/shard_n1_line.json profiles/${T}_shard_n1_line.json; cp $P/shard_n1_rccl_line.json profiles/${T}_shard_n1_rccl_line.json
cp $P/shard_n1_line_20steps.json profiles/${T}_shard_n1_line_20steps.json 2>/dev/null
cp $P/shard_host_timing.txt profiles/${T}_shard_host_timing.txt; cp $P/rehearse_n8.log profiles/${T}_rehearse_n8.log 2>/dev/null; cp $P/rehearse_c4_n8.log profiles/${T}_rehearse_c4_n8.log 2>/dev/null
