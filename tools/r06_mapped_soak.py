"""Soak of the mapped-peer exchanges between rank PROCESSES on one GPU (tests/test_gpu_multiproc.py's processes, many more steps):
  (1) `short` steps, mapped against the plain run (table's collectives through the host): every row, replicated tensor and the step count bit for bit;
  (2) `long` steps, mapped only: the replicated tensors bit-identical between the ranks, one put launch per exchange and step, no wait that ran
      into its bound, nothing through the table but the set-up.
    python tools/r06_mapped_soak.py [world] [short] [long] > profiles/r06_mapped_soak.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("PS_MULTIPROC_TIMEOUT", "600")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
short = int(sys.argv[2]) if len(sys.argv) > 2 else 200
long_ = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
MAPPED = dict(own_in_place=True, tune=dict(mapped_peer=1, spin_timeout_ms=30000))


def run(steps, opts):
    os.environ["PS_MULTIPROC_STEPS"] = str(steps)
    for m in [k for k in sys.modules if k.startswith("test_gpu_multiproc")]:
        del sys.modules[m]
    import test_gpu_multiproc as T
    assert T.STEPS == steps
    t0 = time.time()
    print("... %d processes x %d steps, %s" % (world, steps, "mapped" if "tune" in opts else "plain"), flush=True)
    out = T.run_processes(world, False, True, opts)
    print("    %.1f s" % (time.time() - t0), flush=True)
    return T, out, time.time() - t0


def main():
    T, ref, dt0 = run(short, dict(own_in_place=True))
    T, got, dt1 = run(short, MAPPED)
    for r in range(world):
        T._same(ref[r], got[r], "rank %d" % r)
        assert got[r][9] == 0 and got[r][11][0] == 1 and got[r][11][2] == short and got[r][11][3] == short, (got[r][9], got[r][11])
    print("%d rank processes x %d steps: mapped peer == the table's collectives, every row / replicated tensor / step count bit for bit (%.1f s plain, %.1f s mapped)"
          % (world, short, dt0, dt1))
    T, got, dt = run(long_, MAPPED)
    for r in range(world):
        rows, fcw, fcb, wide, wbias, gstep, mode, why, calls, timeouts, xstats, mapped = got[r]
        assert timeouts == 0 and xstats[0] == long_ and mapped[0] == 1 and mapped[2] == long_ and mapped[3] == long_, (timeouts, xstats, mapped)
        assert calls["all_to_all_v"] == 1 and calls["all_reduce"] == 1 and calls["all_gather"] == 5, calls
        for a, b in zip(fcw + fcb + [wide, wbias], got[0][1] + got[0][2] + [got[0][3], got[0][4]]):
            assert np.array_equal(a, b), "rank %d's replicated tensors differ from rank 0's" % r
        assert all(np.isfinite(x).all() for x in fcw)
    print("%d rank processes x %d steps over mapped peer memory in %.1f s: replicated tensors bit-identical between the ranks, %d put launches per kind and rank, "
          "0 waits ran into their bound, the table carried the set-up only (%s); join mode %s" % (world, long_, dt, got[0][11][2], got[0][8], got[0][6]))


if __name__ == "__main__":      # (the rank processes are SPAWNED: they import this file)
    main()
