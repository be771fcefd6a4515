"""Print a window of a rocprofv3 kernel-trace CSV as a timeline: start offset (us), duration, queue, name."""
import csv, sys
path, t_from, t_len = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
qs = {}
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    if s < t_from or s > t_from + t_len:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    name = r["Kernel_Name"].split("(")[0][-60:]
    print("%10.1f %7.1f  q%d %s%s" % (s - t_from, d, q, "    " * q, name))
