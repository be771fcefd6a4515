#!/usr/bin/env python
"""Summarise rocprofv3 CSV output of tools/profile_round.sh: per-kernel duration stats and per-kernel
average FETCH_SIZE / WRITE_SIZE per dispatch (PMC passes)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n.strip()[:60]


def main():
    out = sys.argv[1]
    st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        print("== rocprofv3 --kernel-trace --stats (whole bench run incl. warm-up and gather micro-bench) ==")
        print("%-62s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
        for row in csv.DictReader(open(st[0])):
            print("%-62s %7d %11.1f %9.2f %9.2f %9.2f %6.1f" % (short(row["Name"]), int(row["Calls"]), float(row["TotalDurationNs"]) / 1e3,
                  float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, float(row["Percentage"])))
        print()
    tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        per = defaultdict(list)
        for row in csv.DictReader(open(tr[0])):
            per[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        tot = sum(sum(v) for v in per.values())
        print("== kernel trace (us) ==")
        print("%-62s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total", "avg", "min", "max", "%"))
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print("%-62s %7d %11.1f %9.2f %9.2f %9.2f %6.1f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    for name in ("fetch", "write"):
        cc = glob.glob(os.path.join(out, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True)
        if not cc:
            continue
        per = defaultdict(list)
        for row in csv.DictReader(open(cc[0])):
            per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        cname = "FETCH_SIZE" if name == "fetch" else "WRITE_SIZE"
        print("\n== %s per dispatch (counter units as reported by rocprofv3; KB) ==" % cname)
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print("%-62s %7d avg %14.1f" % (k, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
