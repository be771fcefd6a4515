#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats of the default bench command, then two PMC passes
# (FETCH_SIZE, WRITE_SIZE -- separate passes: TCC has 4 slots, FETCH_SIZE costs 3) for HBM traffic.
# Output: gpurun_out/prof_<tag>/  (copy the summaries into profiles/).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 200 --warmup 20 --no-cpu --gather 1 --multi-hot 0 --sharded-leg 0 --clamped 0 --fused-adam 0 --ingest-fed 0"
if [ -z "${PMC_ONLY:-}" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c2 -- $BENCH > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-400
fi
# (counter collection runs one kernel at a time: libps_amd sees ROCPROF_COUNTER_COLLECTION and replaces every device-side
# flag wait by its event form -- a spinner could otherwise wait for a kernel the profiler has not let run yet)
BENCH2="python bench.py --steps 40 --warmup 5 --priming 0 --no-cpu --gather 1 --multi-hot 0 --sharded-leg 0 --clamped 0 --fused-adam 0 --ingest-fed 0"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o c2 -- $BENCH2 > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o c2 -- $BENCH2 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# HBM bytes per launch of every kernel group from the two PMC passes (before the raw tables are dropped)
python tools/pmc_traffic.py $OUT $TAG > $OUT/pmc_traffic.txt 2>&1; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
# keep the merged output small: drop the raw per-dispatch tables
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
