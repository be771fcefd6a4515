#!/bin/bash
# kernel trace of a short fused-step run; prints one step's timeline (run on the GPU box)
TAG=${1:-a}
OUT=gpurun_out/r2/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --steps 80 --warmup 10 --no-cpu --gather 0 --multi-hot 0 > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $F > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -size +3M -delete
