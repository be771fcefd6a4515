#!/bin/bash
# Round-3 evidence run on the GPU box: bench line, kernel-trace stats, PMC traffic, GPU-side timelines, sharded lines, soak.
set -u
OUT=gpurun_out/prof_r03
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu_full.log 2>&1; tail -2 $OUT/pytest_gpu_full.log
python bench.py > $OUT/c2_bench_line.json 2> $OUT/c2_bench.err; cut -c1-300 $OUT/c2_bench_line.json
bash tools/profile_round.sh r03 > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
python tools/pmc_traffic.py gpurun_out/prof_r03 r03 > $OUT/pmc_traffic.txt 2>&1; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/gpu_timeline.py 64 > $OUT/c2_gpu_timeline.txt 2>&1
MULTI_HOT=1 python tools/gpu_timeline.py 8 > $OUT/c4_gpu_timeline.txt 2>&1
python bench.py --sharded --steps 2000 --no-cpu --gather 0 --multi-hot 0 > $OUT/shard_n1_line.json 2> $OUT/shard.err
PS_BENCH_STAGE=1 python bench.py --sharded --steps 1000 --no-cpu --gather 0 --multi-hot 0 > $OUT/shard_n1_stage1_line.json 2>> $OUT/shard.err
PS_STAMPS=$OUT/shard_stamps.json python bench.py --sharded --steps 300 --no-cpu --gather 0 --multi-hot 0 > /dev/null 2>&1
python tools/shard_timeline.py $OUT/shard_stamps.json > $OUT/shard_gpu_timeline.txt 2>&1; rm -f $OUT/shard_stamps.json
timeout 1500 python tools/soak_multirank.py ${SOAK:-1000} > $OUT/soak_multirank.log 2>&1; tail -2 $OUT/soak_multirank.log
echo done
