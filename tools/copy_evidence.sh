#!/bin/bash
# gpurun_out/prof_r03 (written by tools/profile_r03.sh on the GPU box) -> profiles/r03_*
P=gpurun_out/prof_r03
head -22 $P/trace/c2_kernel_stats.csv > profiles/r03_c2_bench_kernel_stats.csv
cp $P/c2_bench_line.json profiles/r03_c2_bench_line.json; cp $P/summary.txt profiles/r03_c2_bench_summary.txt
cp $P/c2_gpu_timeline.txt profiles/r03_c2_gpu_timeline.txt; cp $P/c4_gpu_timeline.txt profiles/r03_c4_gpu_timeline.txt
cp $P/pmc_traffic.txt profiles/r03_pmc_traffic.txt; cp $P/pmc_traffic.json profiles/pmc_traffic.json
tail -3 $P/pytest_gpu_full.log > profiles/r03_pytest_gpu_full.log
cp $P/shard_gpu_timeline.txt profiles/r03_shard_gpu_timeline.txt; cp $P/shard_n1_line.json profiles/r03_shard_n1_line.json; cp $P/shard_n1_stage1_line.json profiles/r03_shard_n1_stage1_line.json
(head -3 $P/soak_multirank.log; echo ...; tail -17 $P/soak_multirank.log) > profiles/r03_soak_multirank.log
