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
