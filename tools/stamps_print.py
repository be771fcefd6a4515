"""Print one step from a stamps dump (PS_STAMPS=<file> python bench.py --sharded ...): python tools/stamps_print.py <file> [first kernel name]"""
import json, sys
import numpy as np
d = json.load(open(sys.argv[1]))
first = sys.argv[2] if len(sys.argv) > 2 else "emb_fwd"
nm = d["names"]; v = np.array(d["vals"], np.int64).reshape(-1, 2) / 100.0
starts = [i for i, x in enumerate(nm) if x == first]
per = starts[1] - starts[0]
spans = np.diff([v[i, 0] for i in starts])
print("%d launches, %d steps of %d stamped launches; span median %.1f us" % (len(nm), len(starts) - 1, per, np.median(spans)))
sel = [k for k in range(3, len(starts) - 1) if starts[k + 1] - starts[k] == per and abs(spans[k] - np.median(spans)) < 0.03 * np.median(spans)]
T = np.mean([v[starts[k]:starts[k] + per + 1] - v[starts[k], 0] for k in sel], axis=0)
prev = None
for i in range(per + 1):
    print("%8.1f -> %8.1f (%5.1f)  %-16s%s" % (T[i, 0], T[i, 1], T[i, 1] - T[i, 0], nm[starts[sel[0]] + i], "" if prev is None else "   since previous end %.1f" % (T[i, 0] - prev)))
    prev = T[i, 1]
