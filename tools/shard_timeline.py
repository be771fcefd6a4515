"""The sharded step (ps_shard_step) as the GPU ran it, from the in-kernel stamps bench.py --sharded writes with
PS_STAMPS=<file>:   PS_STAMPS=gpurun_out/x.json python bench.py --sharded --steps 300 --no-cpu; python tools/shard_timeline.py gpurun_out/x.json"""
import json, sys
import numpy as np
d = json.load(open(sys.argv[1]))
nm = d["names"]; v = np.array(d["vals"], np.int64).reshape(-1, 2) / 100.0
ok = [i for i in range(len(nm)) if v[i, 1] > 0 and v[i, 0] < 1e15]
starts = [i for i in ok if nm[i] == "emb_fwd"]
per = starts[1] - starts[0]
spans = np.diff([v[i, 0] for i in starts])
med = np.median(spans)
sel = [k for k in range(3, len(starts) - 1) if starts[k + 1] - starts[k] == per and abs(spans[k] - med) < 0.03 * med]
print("%d launches, %d steps of %d stamped launches; span median %.1f us, mean %.1f, p10 %.1f, p90 %.1f, max %.1f" % (len(nm), len(starts) - 1, per, med, spans.mean(), np.percentile(spans, 10), np.percentile(spans, 90), spans.max()))
T = np.mean([v[starts[k]:starts[k] + per + 1] - v[starts[k], 0] for k in sel], axis=0)
order = np.argsort(T[:, 0], kind="stable")
for i in order:
    print("%8.1f -> %8.1f (%5.1f)  %s" % (T[i, 0], T[i, 1], T[i, 1] - T[i, 0], nm[starts[sel[0]] + i] if i < per else "emb_fwd (next step)"))
