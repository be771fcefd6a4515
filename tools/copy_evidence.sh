#!/bin/bash
# gpurun_out/prof_r04 (written by tools/profile_r04.sh on the GPU box) -> profiles/r04_*
P=gpurun_out/prof_r04
head -22 $P/trace/c2_kernel_stats.csv > profiles/r04_c2_bench_kernel_stats.csv
cp $P/c2_bench_line.json profiles/r04_c2_bench_line.json; cp $P/summary.txt profiles/r04_c2_bench_summary.txt
cp $P/c2_gpu_timeline.txt profiles/r04_c2_gpu_timeline.txt; cp $P/c4_gpu_timeline.txt profiles/r04_c4_gpu_timeline.txt
cp $P/pmc_traffic.txt profiles/r04_pmc_traffic.txt; cp $P/pmc_traffic.json profiles/pmc_traffic.json
grep "passed\|failed" $P/pytest_gpu_full.log > profiles/r04_pytest_gpu_full.log; grep "passed\|failed" $P/pytest_gemm_lab.log > profiles/r04_pytest_gemm_lab.log
cp $P/shard_gpu_timeline.txt profiles/r04_shard_gpu_timeline.txt; cp $P/shard_n1_line.json profiles/r04_shard_n1_line.json; cp $P/shard_n1_rccl_line.json profiles/r04_shard_n1_rccl_line.json
cp $P/shard_host_timing.txt profiles/r04_shard_host_timing.txt; cp $P/rehearse_n8.log profiles/r04_rehearse_n8.log 2>/dev/null
