"""Soak of the N-ranks-as-threads exchange test that hung once in round 1 (test_n_ranks_on_one_gpu[4-False-True]):
N iterations in ONE process, each under a deadline, a line per iteration.
    python tools/soak_multirank.py [iterations] > profiles/rNN_soak_multirank.log"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_multirank as T
from oracle import oracle as orc
from ps_amd import native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
world, is_async, pipelined = 4, False, True
emb, fcW, fcb, ww, wb = T.expected(world, is_async)
t_all = time.time()
worst = 0.0
for it in range(n):
    shared = T.Shared(world)
    out, errs = [None] * world, []
    t0 = time.time()
    try:
        T.run_ranks(T.rank_main, [(r, world, shared, is_async, pipelined, out, errs) for r in range(world)], deadline_s=60)
    except BaseException as e:      # noqa: BLE001 -- pytest.fail raises
        print("iteration %d: STUCK / FAILED after %.1f s: %s" % (it, time.time() - t0, e), flush=True)
        sys.exit(1)
    if errs:
        print("iteration %d: rank error\n%s" % (it, errs[0][1]), flush=True)
        sys.exit(1)
    dt = time.time() - t0
    worst = max(worst, dt)
    ok = all(np.abs(out[r][1][0] - fcW[0]).max() <= 2e-5 * T.STEPS for r in range(world)) and all(np.array_equal(out[r][1][0], out[0][1][0]) for r in range(world))
    if it % 50 == 0 or not ok:
        print("iteration %4d: %.2f s  %s" % (it, dt, "ok" if ok else "WRONG RESULT"), flush=True)
    if not ok:
        sys.exit(1)
print("%d iterations of test_n_ranks_on_one_gpu[4-False-True] in one process: all returned and agreed with the PS simulation; "
      "%.0f s in total, slowest iteration %.2f s" % (n, time.time() - t_all, worst))
