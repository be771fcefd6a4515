#!/bin/bash
# effective shader clock per kernel = GRBM_GUI_ACTIVE / duration (same profiled dispatch)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_clock -o g -- python bench.py --steps 40 --warmup 5 --no-cpu --gather 0 > /dev/null 2>&1
python - <<PY
import csv, collections, glob
cc=glob.glob("gpurun_out/pmc_clock/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    k=r["Kernel_Name"]; i=k.find("k_")
    name=k[i:k.find("(",i)] if i>=0 else k[:30]
    agg[(name, r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
    agg[(name, r["Dispatch_Id"])]["dur"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
per=collections.defaultdict(list)
for (name,_),c in agg.items():
    if "GRBM_GUI_ACTIVE" in c and c["dur"]>0:
        per[name].append((c["GRBM_GUI_ACTIVE"]/c["dur"], c["dur"]/1e3, c.get("SQ_VALU_MFMA_BUSY_CYCLES",0)))
for name,v in sorted(per.items(), key=lambda kv:-sum(x[1] for x in kv[1])):
    n=len(v); print("%-34s n=%4d  clk %.2f GHz  dur %.1f us  mfma_busy/SIMD-cycle %.2f" % (name, n, sum(x[0] for x in v)/n, sum(x[1] for x in v)/n, sum(x[2] for x in v)/max(1e-9,sum(x[0]*x[1]*1e3 for x in v))/1024))
PY
