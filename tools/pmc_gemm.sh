#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_gemm -o g -- python tools/gemm_one.py 2>&1 | grep -v "^W2026\|^E2026" | tail -5
python - <<PY
import csv, collections, glob
f=glob.glob("gpurun_out/pmc_gemm/**/*counter_collection.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if "gemm" not in k: continue
    i=k.find("k_gemm")
    key=(k[i:k.find("(",i)], r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key,c in agg.items():
    print(key, {n: round(sum(v)/len(v)) for n,v in c.items()})
PY
