#!/bin/bash
# configs[3]'s fused backward + Adam on the 320 M-row table: the leg alone, its rocprofv3 kernel stats, and two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs).  Output: gpurun_out/adam_<tag>/
set -u
TAG=${1:-r06}
OUT=gpurun_out/adam_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LEG="python bench.py --leg fused_adam"
timeout 300 $LEG > $OUT/line.json 2> $OUT/line.err; tail -c 2000 $OUT/line.json
if [ -z "${NO_PROF:-}" ]; then
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o adam -- $LEG > $OUT/trace.log 2>&1
python tools/adam_profile.py $OUT > $OUT/summary.txt 2>&1
fi
if [ -n "${PMC:-}" ]; then
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o adam -- $LEG > $OUT/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o adam -- $LEG > $OUT/pmc_write.log 2>&1
python tools/adam_profile.py $OUT pmc >> $OUT/summary.txt 2>&1
fi
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
