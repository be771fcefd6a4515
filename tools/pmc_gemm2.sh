#!/bin/bash
# MFMA / LDS counters of k_gemm_nt's two slab loops (tools/gemm_one2.py); output: gpurun_out/pmc_gemm2.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gemm2
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_gemm2 -o g -- python tools/gemm_one2.py 2>&1 | grep -v "^W2026\|^E2026" | tail -5
python - <<PY | tee gpurun_out/pmc_gemm2.txt
import csv, collections, glob
f=glob.glob("gpurun_out/pmc_gemm2/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_gemm_nt" not in k: continue
    i=k.find("k_gemm_nt"); name=k[i:k.find("(",i)]
    d=r["Dispatch_Id"]
    agg[(name, r["Grid_Size"], d)][r["Counter_Name"]]=float(r["Counter_Value"])
    agg[(name, r["Grid_Size"], d)]["dur_us"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
per=collections.defaultdict(list)
for (name, grid, d), c in agg.items(): per[(name, grid, round(c["dur_us"], -1) > 100)].append(c)
for key, v in per.items():
    n=len(v); avg=lambda k: sum(c.get(k, 0) for c in v)/n
    busy=avg("SQ_VALU_MFMA_BUSY_CYCLES"); act=avg("GRBM_GUI_ACTIVE")
    print("%-44s grid %-8s %s  n=%2d  dur %7.1f us  SQ_VALU_MFMA_BUSY_CYCLES / (4 x GRBM_GUI_ACTIVE) %.3f  LDS bank conflict cycles / busy %.4f  wait-LDS / wait-any %.3f" % (
        key[0], key[1], "long K" if key[2] else "fwd0  ", n, avg("dur_us"), busy / max(act, 1) / 4.0, avg("SQ_LDS_BANK_CONFLICT") / max(avg("SQ_BUSY_CYCLES"), 1), avg("SQ_WAIT_INST_LDS") / max(avg("SQ_WAIT_INST_ANY"), 1)))
PY
