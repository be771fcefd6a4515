"""Distribution of the sharded step's period (emb_fwd start to emb_fwd start) over one stamped run: bimodal or not?
PS_STAMPS=gpurun_out/x.json python bench.py --sharded --steps 300 --no-cpu --gather 0 --multi-hot 0; python tools/shard_spans.py gpurun_out/x.json"""
import json, sys
import numpy as np
d = json.load(open(sys.argv[1]))
nm = d["names"]; v = np.array(d["vals"], np.int64).reshape(-1, 2) / 100.0
starts = [i for i in range(len(nm)) if nm[i] == "emb_fwd" and v[i, 1] > 0 and v[i, 0] < 1e15]
spans = np.diff([v[i, 0] for i in starts])
q = np.percentile(spans, [5, 25, 50, 75, 95])
print("%d steps: p5 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p95 %.1f us; mean %.1f" % (len(spans), *q, spans.mean()))
h, e = np.histogram(spans, bins=np.arange(140, 200, 4))
print(" ".join("%d:%d" % (int(a), b) for a, b in zip(e[:-1], h)))
# where the long steps lose their time: mean start of every stamped launch in the slow steps minus the same in the typical ones
per = starts[1] - starts[0]
med = np.median(spans)
typ = [k for k in range(3, len(starts) - 1) if starts[k + 1] - starts[k] == per and abs(spans[k] - med) < 0.03 * med]
slow = [k for k in range(3, len(starts) - 1) if starts[k + 1] - starts[k] == per and spans[k] > 1.15 * med]
if typ and slow:
    T = np.mean([v[starts[k]:starts[k] + per + 1] - v[starts[k], 0] for k in typ], axis=0)
    S = np.mean([v[starts[k]:starts[k] + per + 1] - v[starts[k], 0] for k in slow], axis=0)
    print("%d slow steps (> 1.15 x median); launch: typical start -> slow start (delay)" % len(slow))
    for i in np.argsort(T[:, 0], kind="stable"):
        print("  %-18s %7.1f -> %7.1f  (%+6.1f)   duration %5.1f -> %5.1f" % (nm[starts[typ[0]] + i] if i < per else "emb_fwd (next)", T[i, 0], S[i, 0], S[i, 0] - T[i, 0], T[i, 1] - T[i, 0], S[i, 1] - S[i, 0]))
