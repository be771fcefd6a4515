#!/usr/bin/env python
"""tests/golden/auc_testauc.npz: the ONLY vector the reference's tests hold for this build's scope --
the 1000 (p, y) pairs of src/test/java/TestAuc.java:10-11 (data: two comma-separated float lists).  The
reference prints AUC.calculate() without asserting a value, so `expected` is produced by the restatement
(oracle.auc, evaluate/AUC.java:32-82 op for op in double).  Run in the build container only:

    python tests/golden/make_auc_fixture.py
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402

src = open("/root/reference/src/test/java/TestAuc.java").read()
ps = re.search(r'String ps = "([^"]*)"', src).group(1)
ys = re.search(r'String ys = "([^"]*)"', src).group(1)
p = np.array([np.float32(t) for t in ps.split(", ")], np.float32)      # Float.parseFloat
y = np.array([np.float32(t) for t in ys.split(", ")], np.float32)
assert p.size == y.size
np.savez_compressed(os.path.join(HERE, "auc_testauc.npz"), p=p, y=y, expected=np.array([orc.auc(p, y)], np.float64))
print(p.size, "pairs, positives", int((y > 0).sum()), "AUC (restatement)", repr(orc.auc(p, y)))
