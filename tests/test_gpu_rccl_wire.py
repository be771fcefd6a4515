"""The RCCL wire of ps_shard_step, EXECUTED on the one GPU a test box has (VERDICT r3 next #1a).

At N = 1 a table made by ps_comm_rccl_create needs no wire and loads none.  ps_tune_set("rccl_force", 1 | 2) makes it
load librccl anyway, create the three 1-rank communicators (rows + gradients | key lists | all-reduce) and send EVERY
collective of the step through RCCL on the stream it belongs to: ncclAllGather (the one-time agreement on block sizes),
grouped ncclSend + ncclRecv to this rank itself for the three all-to-all-v of a step (id blocks on side chain 0, rows and
gradients on the training stream), ncclAllReduce of the flat dense / wide gradient on side chain 1.  That is the code
that replaces net/PSRouterClient.java:60-151 <-> net/PServer.java:102-283 -- the dlopen, the hand-declared enum values,
the 128-byte ids, three communicators driven from three streams -- and it had never run.

  force = 1  everything the step reads came off the wire (a rank's own keys too)
  force = 2  the wire runs and a rank's own keys are read in place, as at N > 1 (PS_COMM_OWN_IN_PLACE)

Both must leave the tables of the fused single-GPU step, bit for bit, after 200 pipelined steps; ps_comm_selfcheck runs
its patterns through all three communicators first.  Each case runs in a child process (its own HIP + RCCL runtime).

Round 5: one force = 2 case on MULTI-HOT batches (bags of 0..5 ids, Ftrl on the embedding rows, async push: configs[4]'s
shape of step -- the plan takes the radix sort, the backward the chunked per-key sums, the push one Ftrl step per push)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
f32 = np.float32
SEED = 0x5EED
STEPS = 200


def child(force, is_async, q, bags=False):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
        import ctypes as C
        import ps_amd
        from ps_amd import native as N
        from ps_amd.sharded import NativeWorker
        from test_gpu_schedule import batches
        L = N.lib()
        F, D, X, fc, V, B, WS = 5, 16, 3, [32, 16, 1], 500, 512, 61
        rng = np.random.default_rng(21)
        data = batches(rng, 9, B, F, X, V, WS)
        if bags:            # (ids, offsets) in place of one id per (sample, field); empty bags included
            bagged = []
            for E, Xd, Y, W in data:
                lens = rng.integers(0, 6, size=B * F)
                offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                bagged.append((rng.integers(0, V, size=int(offsets[-1])).astype(np.int64), Xd, Y, W, offsets))
            data = bagged
        res, info = [], {}
        for native in (False, True):
            kv = ps_amd.KVStore(0, SEED)
            kv.create_embedding([V] * F, D)
            if bags:
                kv.set_updater("emF", ps_amd.FtrlUpdater())
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B, max_nnz=B * F * 5 if bags else 0)
            bs = [ps_amd.DeviceBatch(kv, *d) for d in data]
            if native:
                assert "librccl" not in open("/proc/self/maps").read(), "RCCL was mapped before anybody asked for a wire"
                N.check(L.ps_tune_set(b"rccl_force", force))
                wk = NativeWorker([gm], 1, 0, is_async=is_async)          # id = None: the 1-rank ids are made inside
                N.check(L.ps_tune_set(b"rccl_force", 0))
                wk.selfcheck()                                            # all three communicators, known patterns
                wk.run(bs, STEPS)                                         # the one-model pipeline bench.py --gpus N runs
                kv.sync()
                cc, ur, hs = C.c_int(), C.c_int(), C.c_int()
                N.check(L.ps_comm_rccl_info(C.byref(wk.ops), C.byref(cc), C.byref(ur), C.byref(hs)))
                calls = (C.c_int64 * 5)()
                N.check(L.ps_comm_rccl_calls(C.byref(wk.ops), calls))
                why = C.create_string_buffer(256)
                st10 = (C.c_int64 * 10)()
                N.check(L.ps_shard_exchange_stats(gm.h, st10, 10))
                info = {"ncclCommCount": cc.value, "ncclCommUserRank": ur.value, "extra_communicators": hs.value, "calls": [int(x) for x in calls],
                        "join_mode": L.ps_store_join_mode(kv.h, why, 256), "why": why.value.decode(), "timeouts": int(L.ps_store_wait_timeouts(kv.h)),
                        "librccl_mapped": "librccl" in open("/proc/self/maps").read(), "stats": [int(x) for x in st10]}
                wk.close()
            else:
                for i in range(STEPS):
                    gm.train_async(bs[i % len(bs)])
            kv.sync()
            res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get_rows(f, np.arange(V), 1) for f in range(F)],
                        [kv.get("fc%d.weights" % i) for i in range(3)], [kv.get("fc%d.bias" % i) for i in range(3)],
                        kv.get_wide(np.arange(WS)), kv.get("wide.bias"), kv.global_step()))
            for b in bs:
                b.close()
            gm.close(); kv.close()
        q.put(("ok", res, info))
    except BaseException:       # noqa: BLE001
        import traceback
        q.put(("fail", traceback.format_exc(), None))


@pytest.mark.parametrize("force,is_async,bags", [(1, False, False), (2, False, False), (2, True, False), (2, True, True)])
def test_the_rccl_wire_runs_on_one_gpu(force, is_async, bags):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(force, is_async, q, bags), daemon=True)
    p.start()
    try:
        status, res, info = q.get(timeout=170)       # (the first load of librccl's 570 MB on a cold box takes a while)
    finally:
        p.join(30)
        if p.is_alive():
            p.kill()                                 # exactly the process this test started
    assert status == "ok", res
    fused, wired = res
    assert fused[6] == wired[6] == STEPS
    for a, b in zip(fused[:4], wired[:4]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(fused[4], wired[4]); np.testing.assert_array_equal(fused[5], wired[5])
    # RCCL's own view: one rank, index 0, and both extra communicators exist
    assert info["ncclCommCount"] == 1 and info["ncclCommUserRank"] == 0 and info["extra_communicators"] == 2, info
    assert info["librccl_mapped"]
    ag, groups, ar, p2p, wire = info["calls"]
    # selfcheck: 3 x (all-gather, all-to-all-v, all-reduce); the model's agreement: 1 all-gather; per step: id blocks, rows,
    # gradients (a self send + a self receive each) and one all-reduce
    assert wire == 1 and ag == 3 + 1 and ar == 3 + STEPS, info
    assert groups == 3 + 3 * STEPS and p2p == 2 * groups, info
    # ... on three streams: the joins are device-side flags and the overlap mode is on (the step did not fall back)
    assert info["join_mode"] == 1 and info["why"] == "" and info["timeouts"] == 0, info
    assert info["stats"][0] == STEPS and info["stats"][8] == 0, info
