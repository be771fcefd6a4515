#!/bin/bash
# Round-6 evidence run on the GPU box (one gpurun call): full GPU suite, the bench lines (default and the driver's 20-step command),
# kernel-trace stats + PMC traffic of the headline, the configs[3] fused-Adam leg with its own kernel stats + PMC passes, GPU-side
# timelines (fused / multi-hot / sharded), the sharded N = 1 lines, the lab build's tests.
set -u
OUT=gpurun_out/prof_r06
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu_full.log 2>&1; tail -2 $OUT/pytest_gpu_full.log
python bench.py > $OUT/c2_bench_line.json 2> $OUT/c2_bench.err; cut -c1-300 $OUT/c2_bench_line.json
T0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/c2_bench_line_20steps.json 2> $OUT/c2_bench_20.err; echo "driver command wall seconds: $(( $(date +%s) - T0 ))" | tee $OUT/driver_command_seconds.txt; cut -c1-300 $OUT/c2_bench_line_20steps.json
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
PMC=1 bash tools/r06_adam.sh r06 > $OUT/adam.log 2>&1; tail -12 $OUT/adam.log; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/gpu_timeline.py 64 > $OUT/c2_gpu_timeline.txt 2>&1
MULTI_HOT=1 python tools/gpu_timeline.py 8 > $OUT/c4_gpu_timeline.txt 2>&1
PS_HOST_TIMING=1 python bench.py --sharded --steps 2000 --no-cpu --gather 0 --multi-hot 0 > $OUT/shard_n1_line.json 2> $OUT/shard.err
grep "host:" $OUT/shard.err | head -3 > $OUT/shard_host_timing.txt
python bench.py --leg sharded_n1 --steps 1000 > $OUT/shard_n1_modes_line.json 2>> $OUT/shard.err
python bench.py --leg sharded_n1 --steps 20 > $OUT/shard_n1_modes_line_20steps.json 2>> $OUT/shard.err
PS_STAMPS=$OUT/shard_stamps.json python bench.py --sharded --wire-cost 0 --steps 300 --no-cpu --gather 0 --multi-hot 0 > /dev/null 2>&1
python tools/shard_timeline.py $OUT/shard_stamps.json > $OUT/shard_gpu_timeline.txt 2>&1; rm -f $OUT/shard_stamps.json
if [ -f ps_amd/lib/libps_amd_lab.so ]; then PS_AMD_LIB=$PWD/ps_amd/lib/libps_amd_lab.so python -m pytest tools/test_gemm_lab.py tests/test_gpu_schedule.py -m gpu -q > $OUT/pytest_gemm_lab.log 2>&1; tail -1 $OUT/pytest_gemm_lab.log; fi
for f in rehearse_n8 rehearse_n8_mapped rehearse_c4_n8 rehearse_c4_n8_mapped; do cp gpurun_out/$f.log $OUT/$f.log 2>/dev/null; done
echo done
