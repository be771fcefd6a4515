#!/bin/bash
# kernel trace of the sharded step at N=1 (ps_shard_step, 1-rank communicator); prints one step's timeline
OUT=gpurun_out/r2/trace_sh_${1:-a}
mkdir -p $OUT
python bench.py --sharded --steps 300 --warmup 20 --priming 50 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('untraced: ms/step %.4f' % d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --sharded --steps 60 --warmup 10 --priming 20 > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n).strip()[:50]
starts = [i for i, r in enumerate(rows) if "k_shard_keys" in r["Kernel_Name"]]
i0, i1 = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = int(rows[i0]["Start_Timestamp"]); qs = {}
for r in rows[i0:i1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    print("%8.1f -> %8.1f (%6.1f)  q%d %s%s" % (s, e, e - s, q, "      " * q, short(r["Kernel_Name"])))
print("step span %.1f us, %d launches" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, i1 - i0))
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
