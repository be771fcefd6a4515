#!/usr/bin/env python
"""Summary of tools/r06_adam.sh's rocprofv3 output (the configs[3] fused backward + Adam leg, `bench.py --leg fused_adam`).

    python tools/adam_profile.py gpurun_out/adam_<tag>        kernel stats of the run (per template instantiation)
    python tools/adam_profile.py gpurun_out/adam_<tag> pmc    FETCH_SIZE / WRITE_SIZE per launch of k_emb_reduce_update ->
                                                              profiles/pmc_traffic.json["fused_adam_hbm"] (read by bench.py)

Counter corrections as tools/pmc_traffic.py (MI355X_MICROARCH.md, HBM section): KiB -> bytes, FETCH_SIZE x 2 on gfx950."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n).strip()[:70]


def role(name):
    # the leg runs the kernel in two instantiations: <4, false, true> single-hot (reference order), <4, true, false> bags of 32
    if not name.startswith("k_emb_reduce_update"):
        return None
    return "bag1" if "false, true" in name.replace(" ", "").replace(",", ", ") else "bag32"


def main():
    out = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "pmc":
        acc = {}
        for which in ("fetch", "write"):
            cc = glob.glob(os.path.join(out, "pmc_" + which, "**", "*counter_collection.csv"), recursive=True)
            per = defaultdict(list)
            for row in csv.DictReader(open(cc[0])):
                r = role(short(row["Kernel_Name"]))
                if r:
                    per[r].append(float(row["Counter_Value"]))
            for r, v in per.items():
                acc.setdefault(r, {})[which] = (len(v), sum(v) / len(v))
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
        pmc = json.load(open(path))
        pmc["fused_adam_hbm"] = {}
        line = None
        try:
            line = json.loads([l for l in open(os.path.join(out, "line.json")) if l.startswith("{")][-1])["fused_adam_hbm"]
        except Exception:
            pass
        print("\n== HBM traffic per launch of k_emb_reduce_update on the 320 M-row table (FETCH_SIZE x 2, WRITE_SIZE; KiB -> bytes) ==")
        for r, v in sorted(acc.items()):
            f, w = v.get("fetch", (0, 0.0)), v.get("write", (0, 0.0))
            rec = {"dispatches": f[0], "fetch_size_KiB_raw": round(f[1], 1), "write_size_KiB_raw": round(w[1], 1),
                   "read_bytes": 2.0 * f[1] * 1024, "write_bytes": w[1] * 1024}
            rec["hbm_bytes_per_launch"] = rec["read_bytes"] + rec["write_bytes"]
            pmc["fused_adam_hbm"][r] = rec
            algo = None
            if line:
                algo = [x for x in line if "bag%d" % x["bag"] == r][0]["algorithmic_bytes"]
            print("%-6s read %9.1f MB  write %9.1f MB  total %9.1f MB%s" % (r, rec["read_bytes"] / 1e6, rec["write_bytes"] / 1e6, rec["hbm_bytes_per_launch"] / 1e6,
                  "  = %.3f x algorithmic (%.1f MB)" % (rec["hbm_bytes_per_launch"] / algo, algo / 1e6) if algo else ""))
        json.dump(pmc, open(path, "w"), indent=1)
        return
    tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
    st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        print("== rocprofv3 --kernel-trace --stats -- python bench.py --leg fused_adam ==")
        print("%-72s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
        for row in csv.DictReader(open(st[0])):
            print("%-72s %7d %11.1f %9.2f %9.2f %9.2f %6.1f" % (short(row["Name"]), int(row["Calls"]), float(row["TotalDurationNs"]) / 1e3,
                  float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, float(row["Percentage"])))
    if tr:
        per = defaultdict(list)
        for row in csv.DictReader(open(tr[0])):
            n = short(row["Kernel_Name"])
            if role(n):
                per[n].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        print("\n== k_emb_reduce_update per instantiation: launches 4..15 are the leg's bracketed pass (one stream at a time, what the bench line")
        print("   reports); 0..3 warm-up and 16.. the step as it runs, beside the dW GEMMs and the dense update of the other two streams ==")
        for k, v in per.items():
            for name, w in (("bracketed pass", v[4:16]), ("beside the other streams", v[16:])):
                if w:
                    print("%-52s %-26s n %3d  avg %9.2f us  min %9.2f  max %9.2f" % (k, name, len(w), sum(w) / len(w), min(w), max(w)))


if __name__ == "__main__":
    main()
