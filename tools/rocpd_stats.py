#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as the --stats table:
kernel name, calls, total/avg/min/max duration.  Usage: rocpd_stats.py results.db [skip_first_n]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    kcols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in kcols else "display_name"
    q = "select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (namecol, kd, ks)
    rows = list(db.execute(q))
    per = {}
    seen = {}
    for name, st, en in rows:
        name = re.sub(r"\(.*", "", name)
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        seen[name] = seen.get(name, 0) + 1
        if seen[name] <= skip:
            continue
        per.setdefault(name, []).append((en - st) / 1e3)
    tot = sum(sum(v) for v in per.values())
    print("%-44s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("%-44s %7d %12.1f %10.2f %10.2f %10.2f %6.1f" % (name[:44], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    print("total kernel time %.1f us over %d dispatches" % (tot, sum(len(v) for v in per.values())))


if __name__ == "__main__":
    main()
