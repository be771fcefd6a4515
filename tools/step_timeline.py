#!/usr/bin/env python
"""One training step out of a rocprofv3 --kernel-trace CSV, as a timeline by HIP queue:
    python tools/step_timeline.py <kernel_trace.csv> [which_step]
A step starts at a k_emb_fwd dispatch and ends before the next one."""
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n).strip()[:52]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_emb_fwd" in r["Kernel_Name"]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
    i0, i1 = starts[which], starts[which + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    qs = {}
    busy = 0.0
    for r in rows[i0:i1]:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3
        e = (int(r["End_Timestamp"]) - t0) / 1e3
        q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
        busy += e - s
        print("%8.1f -> %8.1f (%6.1f)  q%d %s%s" % (s, e, e - s, q, "      " * q, short(r["Kernel_Name"])))
    span = (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3
    print("step span (emb_fwd to next emb_fwd) %.1f us, %d launches, sum of kernel durations %.1f us" % (span, i1 - i0, busy))
    # median span over all steps
    spans = sorted((int(rows[starts[k + 1]]["Start_Timestamp"]) - int(rows[starts[k]]["Start_Timestamp"])) / 1e3 for k in range(len(starts) - 1))
    print("median step span over %d steps: %.1f us" % (len(spans), spans[len(spans) // 2]))


if __name__ == "__main__":
    main()
