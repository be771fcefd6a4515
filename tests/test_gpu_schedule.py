"""The step's scheduling is not allowed to change a single bit.

The fused step runs on three HIP streams, carries its cross-stream events on the launches themselves
(hipExtLaunchKernelGGL stop events), sorts single-hot batches with the one-launch field sort and runs the dense update last
on the main chain.  Every one of those is a switch; whatever the switches say, K training steps must
leave identical tables:
  * default                                   (three streams, launch-carried events, field sort)
  * ps_tune_set("ext_events", 0)              (plain hipEventRecord / hipStreamWaitEvent)
  * ps_tune_set("field_sort", 0)              (general radix sort + segment builder + long-run list by k_long_runs)
  * ps_tune_set("dev_wait", 0)                (the dW chain waits for the head by event, not behind a device-side spinner)
  * ps_tune_set("end_wait", 0)                (the main chain joins side chain 0 behind a spinner launch, not inside the last delta GEMM)
  * ps_tune_set("tail_dev", 0)                (the dense update last on the main chain, not beside the embedding update)
  * ps_tune_set("tail_fused", 0)              (the dense update behind a spinner launch and followed by a flag setter, the main chain ends behind a spinner)
  * ps_tune_set("tn_start_wait", 0)           (a spinner launch in front of every dW GEMM, not only the first)
  * ps_tune_set("tail_defer", 0)              (the step ends behind its dense update; default: the NEXT use of the store's stream pays the join)
  * ps_tune_set("dw_split", 1)                (the first dW GEMM on side chain 0 -- measured slower, off)
  * ps_tune_set("dw_late", 1)                 (the first dW GEMM released with the next delta GEMM)
  * ps_tune_set("gemm_pipe", 0) / gemm_8w   (other slab loops / tiles of the SAME contraction order: k_gemm_nt / k_gemm_tn)
  * profile mode                              (everything on ONE stream: the serial order is the definition)
Also: the sharded plan's presence map is stamped with an 8-bit epoch (no clearing between steps): more than 256
consecutive plans must still equal the fused step (the map is re-zeroed when the epoch wraps)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
SEED = 0x5EED


def batches(rng, n, B, F, X, V, WS):
    out = []
    for _ in range(n):
        E = np.minimum(rng.zipf(1.2, (B, F)) - 1, V - 1).astype(np.int64)
        out.append((E, rng.standard_normal((B, X)).astype(f32), (rng.random(B) < 0.3).astype(f32), E % WS))
    return out


DEFAULTS = {"dw_split": 0, "fwd_pair": 0, "fwd_panel": 0, "dw_late": 0, "gemm_pipe": 5, "gemm_tn_cfg": 0, "gemm_nt_cfg": 0, "gemm_ks": 0, "gemm_8w": 0}      # (every other knob: 1)


def run(kind, knobs, profile, data, F, D, X, fc, V, B, WS):
    import ps_amd
    from ps_amd import native as N
    for k, v in knobs.items():
        N.lib().ps_tune_set(k.encode(), v)
    try:
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        if kind == "widedeep":
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
        else:
            gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
        if profile:
            gm.set_profile(True)
        losses = [gm.train(ps_amd.Batch(E, Xd, Y, W if kind == "widedeep" else None)) for E, Xd, Y, W in data]
        out = (losses, [kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(len(fc))],
               [kv.get("fc%d.bias" % i) for i in range(len(fc))])
        if kind == "widedeep":
            out += (kv.get_wide(np.arange(WS)), kv.get("wide.bias"))
        gm.close(); kv.close()
        return out
    finally:
        for k in knobs:
            N.lib().ps_tune_set(k.encode(), DEFAULTS.get(k, 1))


@pytest.mark.parametrize("kind,F,D,X,fc,V,B", [("dnn", 4, 8, 3, [16, 1], 50, 200), ("widedeep", 6, 16, 5, [64, 32, 1], 3000, 2048),
                                               ("dnn", 3, 8, 2, [24, 12, 6, 1], 40, 130), ("dnn", 5, 16, 3, [48, 1], 60, 300),
                                               ("widedeep", 4, 16, 2, [96, 40, 1], 500, 700)])
def test_schedules_agree(kind, F, D, X, fc, V, B):
    rng = np.random.default_rng(F * 100 + B)
    WS = 97
    data = batches(rng, 6, B, F, X, V, WS)
    ref = run(kind, {}, False, data, F, D, X, fc, V, B, WS)
    variants = {"plain events": ({"ext_events": 0}, False), "general sort": ({"field_sort": 0}, False),
                "dW chain released by event": ({"dev_wait": 0}, False), "dense update on the main chain": ({"tail_dev": 0}, False),
                "side chain 0 joined behind a spinner": ({"end_wait": 0}, False),
                "dense update between a spinner and a flag setter": ({"tail_fused": 0}, False),
                "a spinner in front of every dW GEMM": ({"tn_start_wait": 0}, False),
                "round 2's tail": ({"tail_fused": 0, "tn_start_wait": 0}, False),
                "first dW GEMM on side chain 0": ({"dw_split": 1}, False),
                "first two forward GEMMs in one launch": ({"fwd_pair": 1}, False),
                "first dW GEMM held back to the next delta GEMM": ({"dw_late": 1}, False),
                "round 2's GEMM slab loop": ({"gemm_pipe": 0}, False),
                "8-wave 128 x 64 tiles": ({"gemm_8w": 1}, False),
                "the embedding update holds the join with the dense update": ({"tail_defer": 0}, False),
                "general sort with scan launches": ({"field_sort": 0, "radix_scan_free": 0}, False),
                "the head walks the wide ids itself (no LRLayer.forward role in the gather's launch)": ({"wide_in_gather": 0}, False),
                "no raised wave priority": ({"main_prio": 0}, False), "all plain": ({"dev_wait": 0, "ext_events": 0, "field_sort": 0}, False),
                "general sort + plain events": ({"field_sort": 0, "ext_events": 0}, False), "one stream": ({}, True)}
    from ps_amd import native as N
    if b"+gemm_lab" not in N.lib().ps_version():      # the rejected GEMM variants exist in the lab build only (tools/gemm_lab_build.sh)
        for name in ("first two forward GEMMs in one launch", "round 2's GEMM slab loop", "8-wave 128 x 64 tiles"):
            del variants[name]
    for name, (knobs, profile) in variants.items():
        got = run(kind, knobs, profile, data, F, D, X, fc, V, B, WS)
        assert got[0] == ref[0], "%s: losses %s vs %s" % (name, got[0], ref[0])
        for a, b in zip(ref[1:], got[1:]):
            if isinstance(a, list):
                for x, y in zip(a, b):
                    np.testing.assert_array_equal(x, y, err_msg=name)
            else:
                np.testing.assert_array_equal(a, b, err_msg=name)


@pytest.mark.parametrize("kind,F,D,X,fc,V,B", [("widedeep", 6, 16, 5, [64, 32, 1], 3000, 2048), ("dnn", 3, 8, 2, [24, 12, 6, 1], 40, 130)])
def test_graph_replay_is_bit_identical(kind, F, D, X, fc, V, B):
    """ps_model_config_t.use_graph (ps_model.hip train_graph: the step captured once per (batch pointers, B, nnz) and replayed as
    a hipGraph -- streams joined by events, no device-side flags) leaves the eager three-stream step's tables, bit for bit:
    host batches (staged into the model's fixed buffers: ONE graph) and device-resident batches (one graph per batch, replayed
    on the second round)."""
    import ps_amd
    rng = np.random.default_rng(F * 10 + B)
    WS = 97
    data = batches(rng, 4, B, F, X, V, WS)
    res = []
    for graph, resident in ((0, False), (1, False), (1, True)):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        if kind == "widedeep":
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B, use_graph=graph)
        else:
            gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, use_graph=graph)
        mk = (lambda *a: ps_amd.DeviceBatch(kv, *a)) if resident else ps_amd.Batch
        bs = [mk(E, Xd, Y, W if kind == "widedeep" else None) for E, Xd, Y, W in data]
        losses = [gm.train(bs[i % len(bs)]) for i in range(3 * len(bs))]
        out = (losses, [kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get_rows(f, np.arange(V), 1) for f in range(F)],
               [kv.get("fc%d.weights" % i) for i in range(len(fc))], [kv.get("fc%d.bias" % i) for i in range(len(fc))])
        if kind == "widedeep":
            out += ([kv.get_wide(np.arange(WS))], [kv.get("wide.bias")])
        if resident:
            for b in bs:
                b.close()
        gm.close(); kv.close()
        res.append(out)
    for name, got in (("graph replay, host batches", res[1]), ("graph replay, resident batches", res[2])):
        assert got[0] == res[0][0], "%s: losses %s vs %s" % (name, got[0], res[0][0])
        for a, b in zip(res[0][1:], got[1:]):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y, err_msg=name)


def test_graph_replay_multi_hot_look_back_words():
    """build_segments' one-launch look-back (k_seg_fused) and the field sort's tag their per-workgroup words with a
    launch number that a captured graph freezes.  Multi-hot batches above one 8192-pair tile, two resident batches (two graphs),
    each replayed twice in a row, behind an eager forward that already advanced the number: the tables equal the eager step's."""
    import ps_amd
    F, D, X, fc, V, B = 3, 8, 2, [16, 1], 5000, 512
    rng = np.random.default_rng(77)
    data = []
    for _ in range(2):
        lens = np.clip(rng.poisson(10, size=B * F), 1, 40)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
        assert ids.size > 8192 + 1024
        data.append((ids, rng.standard_normal((B, X)).astype(np.float32), (rng.random(B) < 0.3).astype(np.float32), None, offsets))
    nnz_max = max(d[0].size for d in data)
    res = []
    for graph in (0, 1):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, max_nnz=nnz_max, use_graph=graph)
        bs = [ps_amd.DeviceBatch(kv, *d) for d in data]
        gm.forward({"E": data[1][0], "X": data[1][1], "Y": data[1][2], "offsets": data[1][4]})     # eager: the launch number moves on
        losses = [gm.train(bs[i]) for i in (0, 0, 1, 1, 0, 1, 1, 0)]
        res.append((losses, [kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(len(fc))]))
        for b in bs:
            b.close()
        gm.close(); kv.close()
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        np.testing.assert_array_equal(a, b)


def test_plan_epoch_wraps():
    """300 consecutive sharded steps (the plan's 8-bit epoch wraps after 255) == 300 fused steps, bit for bit."""
    import ps_amd
    from ps_amd.sharded import NativeWorker
    F, D, X, fc, V, B, WS = 3, 8, 2, [8, 1], 30, 48, 17
    rng = np.random.default_rng(9)
    data = batches(rng, 7, B, F, X, V, WS)
    res = []
    for native in (False, True):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
        wk = NativeWorker([gm], 1, 0) if native else None
        bs = [ps_amd.Batch(E, Xd, Y, W) for E, Xd, Y, W in data]
        for i in range(300):
            if native:
                wk.step(bs[i % len(bs)], want_loss=False)
            else:
                gm.train_async(bs[i % len(bs)])
        kv.sync()
        res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(2)],
                    kv.get_wide(np.arange(WS)), kv.global_step()))
        if wk:
            wk.close()
        gm.close(); kv.close()
    a, b = res
    assert a[3] == b[3] == 300
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2])


@pytest.mark.parametrize("early,extra", [(1, {}), (0, {}), (1, {"shard_sort_defer": 0}), (1, {"plan_fused": 0}), (1, {"shard_sort_defer": 0, "plan_fused": 0}),
                                         (1, {"tn_start_wait": 0}), (1, {"tn_start_wait": 0, "main_prio": 0})])
def test_pipelined_sharded_steps_on_device_batches(early, extra):
    """ps_shard_step's one-model pipeline on device-resident batches (what bench.py --sharded / --gpus N runs): with the next
    step's plan on side chain 0 while the step trains (early = 1, the default) and with the plan in the step's tail (0); with the
    plan's field sort launched by the forward (default) or right behind the slots; count / emit / pack in one launch or three --
    120 steps each, equal to 120 fused steps bit for bit (the 8-bit plan epoch does not wrap here; the run count's two
    buffers alternate 120 times).  tn_start_wait = 0 (a spinner launch in front of every dW GEMM) is here for what it does to the
    timing: it delays side chain 1, and at these toy sizes the NEXT step's gather then overwrote the first layer's input while the
    last dW GEMM still read it -- in overlap mode nothing on the training stream waits for that chain (6 of 10 runs on some boxes,
    1 of 100 with the default knobs; round 4: the gather's workgroups now wait for the flat-gradient launch's start)."""
    import ps_amd
    from ps_amd import native as N
    from ps_amd.sharded import NativeWorker
    F, D, X, fc, V, B, WS = 5, 16, 3, [32, 16, 1], 500, 512, 61
    rng = np.random.default_rng(21)
    data = batches(rng, 9, B, F, X, V, WS)
    res = []
    N.lib().ps_tune_set(b"plan_early", early)
    for k, v in extra.items():          # round 3's forms of round 4's changes: the sort right behind the slots; count / emit / pack as three launches
        N.lib().ps_tune_set(k.encode(), v)
    try:
        for native in (False, True):
            kv = ps_amd.KVStore(0, SEED)
            kv.create_embedding([V] * F, D)
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
            bs = [ps_amd.DeviceBatch(kv, E, Xd, Y, W) for E, Xd, Y, W in data]
            if native:
                wk = NativeWorker([gm], 1, 0)
                wk.run(bs, 120)
                wk.close()
            else:
                for i in range(120):
                    gm.train_async(bs[i % len(bs)])
            kv.sync()
            res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)],
                        kv.get_wide(np.arange(WS)), kv.global_step()))
            for b in bs:
                b.close()
            gm.close(); kv.close()
    finally:
        N.lib().ps_tune_set(b"plan_early", 1)
        for k in extra:
            N.lib().ps_tune_set(k.encode(), 1)
    a, b = res
    assert a[3] == b[3] == 120
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2])


def test_counter_collecting_profiler_guard():
    """rocprofv3 --pmc runs one kernel at a time in an order of its own; a kernel waiting for another stream's flag then
    deadlocks (round 2 lost two 600 s PMC passes to it).  libps_amd reads ROCPROF_COUNTER_COLLECTION when it is loaded and
    uses the event form of every join: a fresh process with the variable set trains the same steps to the same tables."""
    import subprocess, sys, os, json
    code = r'''
import sys, json, hashlib
import numpy as np
sys.path.insert(0, %r)
import ps_amd
F, D, X, fc, V, B, WS = 6, 16, 5, [64, 32, 1], 3000, 2048, 97
rng = np.random.default_rng(5)
kv = ps_amd.KVStore(0, 0x5EED)
kv.create_embedding([V] * F, D)
gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
h = hashlib.sha256()
for _ in range(5):
    E = np.minimum(rng.zipf(1.2, (B, F)) - 1, V - 1).astype(np.int64)
    loss = gm.train({"E": E, "X": rng.standard_normal((B, X)).astype(np.float32), "Y": (rng.random(B) < 0.3).astype(np.float32), "W": E %% WS})
    h.update(np.float32(loss).tobytes())
for f in range(F):
    h.update(kv.get_rows(f, np.arange(V)).tobytes())
for i in range(3):
    h.update(kv.get("fc%%d.weights" %% i).tobytes())
import ctypes as C
why = C.create_string_buffer(256)
mode = ps_amd.native.lib().ps_store_join_mode(kv.h, why, 256)
print(json.dumps({"digest": h.hexdigest(), "mode": mode, "why": why.value.decode()}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    guards = [None, "ROCPROF_COUNTER_COLLECTION", "HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "GPU_MAX_HW_QUEUES"]
    for guard in guards:
        env = dict(os.environ)
        for g in guards[1:]:
            env.pop(g, None)
        if guard:
            env[guard] = "2" if guard == "GPU_MAX_HW_QUEUES" else ("3" if guard == "AMD_SERIALIZE_KERNEL" else "1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        rep = json.loads(r.stdout.strip().splitlines()[-1])
        out.append(rep["digest"])
        # the serialising environments are recognised when the library is loaded: joins by events, and the reason is reported
        assert rep["mode"] == (0 if guard else 1), rep
        assert (guard in rep["why"]) if guard else rep["why"] == "", rep
    assert all(d == out[0] for d in out[1:])


def test_a_wait_that_is_never_released_times_out_with_an_error():
    """Every device-side wait is bounded: a spinner launch and a GEMM's end wait on a flag nobody
    raises come back after the timeout with PS_E_STATE -- not a hang -- and the store then joins its streams by events;
    training on that store still gives the tables of a store that never timed out."""
    import ctypes as C
    import time
    import ps_amd
    from ps_amd import native as N
    L = N.lib()
    L.ps_dbg_stuck_wait.argtypes = [C.c_void_p, C.c_int]
    L.ps_dbg_stuck_wait.restype = C.c_int
    F, D, X, fc, V, B, WS = 6, 16, 5, [64, 32, 1], 3000, 1024, 97
    rng = np.random.default_rng(8)
    data = batches(rng, 4, B, F, X, V, WS)

    def train(kv):
        gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
        for E, Xd, Y, W in data:
            gm.train({"E": E, "X": Xd, "Y": Y, "W": W})
        r = [kv.get_rows(f, np.arange(V)) for f in range(F)] + [kv.get("fc%d.weights" % i) for i in range(3)]
        gm.close()
        return r

    L.ps_tune_set(b"spin_timeout_ms", 150)
    try:
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        why = C.create_string_buffer(256)
        assert L.ps_store_join_mode(kv.h, why, 256) == 1 and why.value == b""
        for in_gemm in (0, 1):
            t0 = time.perf_counter()
            rc = L.ps_dbg_stuck_wait(kv.h, in_gemm)
            dt = time.perf_counter() - t0
            assert rc == N.PS_E_STATE, (in_gemm, rc, L.ps_last_error())
            assert b"timed out" in L.ps_last_error()
            assert 0.1 < dt < 5.0, dt
        assert L.ps_store_wait_timeouts(kv.h) == 2
        assert L.ps_store_join_mode(kv.h, why, 256) == 0 and b"timed out" in why.value
        got = train(kv)
        kv.close()
    finally:
        L.ps_tune_set(b"spin_timeout_ms", 2000)
    kv2 = ps_amd.KVStore(0, SEED)
    kv2.create_embedding([V] * F, D)
    want = train(kv2)
    kv2.close()
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)


def test_overlap_mode_outlives_the_device_side_joins():
    """The overlap mode of the sharded step (key lists on side chain 0, flat reduction + replicated update on side chain 1)
    is agreed ONCE, at a model's first begin; whether a step may join its streams by device flags is decided per step, and
    can go away (a wait that timed out, a second model on the device, ps_tune_set).  The backward then writes the flat
    gradient [fc | wide G | wide C | bias] on the TRAINING stream while the reduction still runs on side chain 1: the step
    must order the two (ps_comm.hip).  4 x 30 pipelined sharded steps -- default, dev_wait = 0, tail_dev = 0,
    both -- equal 120 fused steps bit for bit."""
    import ps_amd
    from ps_amd import native as N
    from ps_amd.sharded import NativeWorker
    L = N.lib()
    F, D, X, fc, V, B, WS = 5, 16, 3, [32, 16, 1], 500, 512, 61
    rng = np.random.default_rng(22)
    data = batches(rng, 9, B, F, X, V, WS)
    phases = [{}, {"dev_wait": 0}, {"tail_dev": 0}, {"dev_wait": 0, "tail_dev": 0}]
    res = []
    try:
        for native in (False, True):
            kv = ps_amd.KVStore(0, SEED)
            kv.create_embedding([V] * F, D)
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
            bs = [ps_amd.DeviceBatch(kv, E, Xd, Y, W) for E, Xd, Y, W in data]
            wk = NativeWorker([gm], 1, 0) if native else None
            for knobs in phases:
                if native:
                    for k, v in knobs.items():
                        L.ps_tune_set(k.encode(), v)
                    wk.run(bs, 30)
                    kv.sync()
                    for k in knobs:
                        L.ps_tune_set(k.encode(), 1)
                else:
                    for i in range(30):
                        gm.train_async(bs[i % len(bs)])
            kv.sync()
            res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)],
                        kv.get_wide(np.arange(WS)), kv.get("wide.bias"), kv.global_step()))
            if wk:
                wk.close()
            for b in bs:
                b.close()
            gm.close(); kv.close()
    finally:
        for k in ("dev_wait", "tail_dev"):
            L.ps_tune_set(k.encode(), 1)
    a, b = res
    assert a[4] == b[4] == 120
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2]); np.testing.assert_array_equal(a[3], b[3])


@pytest.mark.parametrize("knobs", [{}, {"blk_cap": 64}, {"plan_fused": 0, "blk_cap": 64}, {"rccl_force": 2, "blk_cap": 64},
                                   # round 5: the plan head with the forward / behind the backward, the slot kernel inside the gather's launch
                                   # or behind its own spinner -- every combination
                                   {"plan_mid": 0}, {"slots_in_gather": 0}, {"plan_mid": 0, "slots_in_gather": 0},
                                   {"slots_in_gather": 0, "blk_cap": 64}, {"rccl_force": 2, "slots_in_gather": 0}])
def test_sharded_steps_with_the_one_launch_plan(knobs):
    """A key space large enough for the one-launch plan (k_plan_fused: 20 000 rows -> 4 look-back workgroups; the toy shapes
    above take the three-launch form), 60 pipelined sharded steps == 60 fused steps bit for bit -- also with wire blocks of 64
    rows, so that EVERY step's lists overflow and travel again at full size (packed by the plan's launch, or by
    k_pack_blocks), and with that second exchange going through RCCL (1-rank communicators, self send/recv)."""
    import ps_amd
    from ps_amd import native as N
    from ps_amd.sharded import NativeWorker
    import ctypes as C
    L = N.lib()
    F, D, X, fc, V, B, WS = 4, 16, 3, [32, 16, 1], 5000, 512, 61
    rng = np.random.default_rng(23)
    data = []
    for _ in range(7):
        E = rng.integers(0, V, size=(B, F)).astype(np.int64)
        E[: B // 4] = np.minimum(rng.zipf(1.3, (B // 4, F)) - 1, V - 1)        # some hot keys, many distinct ones
        data.append((E, rng.standard_normal((B, X)).astype(f32), (rng.random(B) < 0.3).astype(f32), E % WS))
    res, st10 = [], None
    try:
        for native in (False, True):
            kv = ps_amd.KVStore(0, SEED)
            kv.create_embedding([V] * F, D)
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
            bs = [ps_amd.DeviceBatch(kv, E, Xd, Y, W) for E, Xd, Y, W in data]
            if native:
                for k, v in knobs.items():
                    L.ps_tune_set(k.encode(), v)
                wk = NativeWorker([gm], 1, 0)
                wk.run(bs, 60)
                kv.sync()
                st = (C.c_int64 * 10)()
                N.check(L.ps_shard_exchange_stats(gm.h, st, 10))
                st10 = [int(x) for x in st]
                wk.close()
            else:
                for i in range(60):
                    gm.train_async(bs[i % len(bs)])
            kv.sync()
            res.append(([kv.get_rows(f, np.arange(V)) for f in range(F)], [kv.get("fc%d.weights" % i) for i in range(3)],
                        kv.get_wide(np.arange(WS)), kv.global_step()))
            for b in bs:
                b.close()
            gm.close(); kv.close()
    finally:
        for k in knobs:
            L.ps_tune_set(k.encode(), {"plan_fused": 1, "plan_mid": 1, "slots_in_gather": 1}.get(k, 0))
    a, b = res
    assert a[3] == b[3] == 60
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2])
    if "blk_cap" in knobs:
        assert st10[8] == 60 and st10[7] < st10[9], st10        # every step took the full-size exchange
    else:
        assert st10[8] == 0, st10


@pytest.mark.parametrize("kind,F,D,X,fc,V,B", [("widedeep", 5, 16, 3, [32, 16, 1], 500, 512), ("widedeep", 8, 16, 4, [128, 64, 1], 20000, 4096),
                                               ("dnn", 4, 8, 2, [48, 1], 300, 256)])
def test_steps_enqueued_back_to_back_equal_steps_with_a_wait_in_between(kind, F, D, X, fc, V, B):
    """Consecutive fused steps overlap at their edges (the next gather beside the dense update's tail, the side chains of one step
    beside the main chain of the next): 80 steps enqueued without a wait (train_async on device-resident batches) must leave
    the tables of 80 steps with a wait behind each -- at toy sizes, where a step is a few kernels of a few microseconds and
    whatever is not ordered explicitly does meet."""
    import ps_amd
    rng = np.random.default_rng(B + F)
    WS = 61
    data = batches(rng, 7, B, F, X, V, WS)
    res = []
    for waits in (True, False, False):
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D)
        if kind == "widedeep":
            gm = ps_amd.WideDeepNN.buildModel(F, D, X, fc, WS, store=kv, max_batch=B)
        else:
            gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B)
        bs = [ps_amd.DeviceBatch(kv, E, Xd, Y, W if kind == "widedeep" else None) for E, Xd, Y, W in data]
        for i in range(80):
            gm.train_async(bs[i % len(bs)])
            if waits:
                gm.sync()
        kv.sync()
        out = [kv.get_rows(f, np.arange(V)) for f in range(F)] + [kv.get("fc%d.weights" % i) for i in range(len(fc))] + [kv.get("fc%d.bias" % i) for i in range(len(fc))]
        if kind == "widedeep":
            out += [kv.get_wide(np.arange(WS)), kv.get("wide.bias")]
        res.append(out)
        for b in bs:
            b.close()
        gm.close(); kv.close()
    for other in res[1:]:
        for x, y in zip(res[0], other):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("seg", [1, 0])
def test_multi_hot_steps_enqueued_back_to_back(seg):
    """The same for multi-hot batches (key kernel and sort chain beside the gather, the segmented sort's scan in front of it;
    FTRL rows): 40 steps without a wait == 40 steps with a wait behind each."""
    import ps_amd
    from ps_amd import native as N
    F, D, X, fc, V, B = 4, 16, 2, [32, 1], 3000, 600
    rng = np.random.default_rng(77)
    data = []
    for _ in range(5):
        lens = rng.poisson(9, size=B * F)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = np.minimum(rng.zipf(1.2, int(offsets[-1])) - 1, V - 1).astype(np.int64)
        data.append((ids, offsets, rng.standard_normal((B, X)).astype(f32), (rng.random(B) < 0.3).astype(f32)))
    nnz_max = max(int(d[1][-1]) for d in data)
    N.lib().ps_tune_set(b"mh_seg_sort", seg)
    res = []
    try:
        for waits in (True, False, False):
            kv = ps_amd.KVStore(0, SEED)
            kv.create_embedding([V] * F, D)
            kv.set_updater("emF", ps_amd.FtrlUpdater())
            gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, max_nnz=nnz_max)
            bs = [ps_amd.DeviceBatch(kv, ids, Xd, Y, None, offsets) for ids, offsets, Xd, Y in data]
            for i in range(40):
                gm.train_async(bs[i % len(bs)])
                if waits:
                    gm.sync()
            kv.sync()
            res.append([kv.get_rows(f, np.arange(V)) for f in range(F)] + [kv.get("fc%d.weights" % i) for i in range(2)])
            for b in bs:
                b.close()
            gm.close(); kv.close()
    finally:
        N.lib().ps_tune_set(b"mh_seg_sort", 1)
    for other in res[1:]:
        for x, y in zip(res[0], other):
            np.testing.assert_array_equal(x, y)


def test_multi_hot_presort_and_one_launch_segments_change_no_bit():
    """Round 5: the multi-hot step's scan / key kernel / first sort pass run on side chain 0 WITHOUT a join with the training
    stream (beside the previous step's backward: ps_tune_set("mh_presort"), modes 0..3 of kernels_sort.hip), and the run boundaries
    come from ONE look-back launch (k_seg_fused: "seg_fused").  60 steps enqueued back to back under every combination leave the same tables, bit for bit --
    batches of very different sizes in a row, so that a sort that started early on the wrong buffers would show."""
    import ps_amd
    from ps_amd import native as N
    F, D, X, fc, V, B = 5, 16, 2, [32, 1], 4000, 700
    rng = np.random.default_rng(78)
    data = []
    for k in range(6):
        lens = rng.poisson(3 + 5 * (k % 3), size=B * F)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ids = np.minimum(rng.zipf(1.2, int(offsets[-1])) - 1, V - 1).astype(np.int64)
        data.append((ids, offsets, rng.standard_normal((B, X)).astype(f32), (rng.random(B) < 0.3).astype(f32)))
    nnz_max = max(int(d[1][-1]) for d in data)
    res = []
    L = N.lib()
    try:
        for knobs in ({}, {"mh_presort": 0}, {"mh_presort": 1}, {"mh_presort": 2}, {"seg_fused": 0}, {"mh_presort": 0, "seg_fused": 0}, {"mh_presort": 3, "mh_prio": 1},
                      # (the update launch's list role: only the runs with super partials as in round 5 / nearly every chunked run on 7 workgroups / no such role)
                      {"emb_list_min": 128}, {"emb_list_min": 2, "emb_list_grid": 7}, {"super_in_update": 0}):
            for k, v in knobs.items():
                N.check(L.ps_tune_set(k.encode(), v))
            kv = ps_amd.KVStore(0, SEED)
            kv.create_embedding([V] * F, D)
            kv.set_updater("emF", ps_amd.FtrlUpdater())
            gm = ps_amd.DNN.buildModel(F, D, X, fc, store=kv, max_batch=B, max_nnz=nnz_max)
            bs = [ps_amd.DeviceBatch(kv, ids, Xd, Y, None, offsets) for ids, offsets, Xd, Y in data]
            for i in range(60):
                gm.train_async(bs[i % len(bs)])
            kv.sync()
            res.append([kv.get_rows(f, np.arange(V)) for f in range(F)] + [kv.get_rows(f, np.arange(V), 1) for f in range(F)] + [kv.get("fc%d.weights" % i) for i in range(2)])
            for b in bs:
                b.close()
            gm.close(); kv.close()
            for k in knobs:
                L.ps_tune_set(k.encode(), {"mh_presort": 3, "mh_prio": 0, "emb_list_min": 16, "emb_list_grid": 256}.get(k, 1))
    finally:
        L.ps_tune_set(b"mh_presort", 3); L.ps_tune_set(b"seg_fused", 1); L.ps_tune_set(b"mh_prio", 0)
        L.ps_tune_set(b"emb_list_min", 16); L.ps_tune_set(b"emb_list_grid", 256); L.ps_tune_set(b"super_in_update", 1)
    for other in res[1:]:
        for x, y in zip(res[0], other):
            np.testing.assert_array_equal(x, y)
