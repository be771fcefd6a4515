"""configs[2] AT FULL SIZE with 8 ranks, on the one GPU a test box has (VERDICT r3 next #1e; was tools/rehearse_n8.py).

Eight rank PROCESSES share the device: 26 fields x 100 k ids x 16 floats sharded id mod 8, FC[512,256,1], batch 4096 per
rank, truncated Zipf(1.05) ids -- BASELINE.json's third config, ps_shard_step over gloo with host staging in place of
RCCL (the RCCL calls themselves run in tests/test_gpu_rccl_wire.py).  Everything else is what an 8-GPU node runs: the
wire blocks of 2 * nnz / 8 rows, 8 peers in the owner-side push, the overlap mode, the one-model pipeline, own keys in
place (PS_COMM_OWN_IN_PLACE, the self part of every receive buffer poisoned).

Checked:
  * after step 1, EVERY embedding row of ranks 0 and 5 in three fields, bit for bit, against the parameter-server
    semantics of net/PServer.java:164-214 -- per key the mean, in worker order, over the workers that pushed it, one Adam
    step (oracle's adam_update) -- applied to the eight workers' per-key gradients as the single-GPU split step
    (ps_model_forward / ps_model_backward, verified against the oracle in test_gpu_configs.py) computes them from the same
    initial parameters; rows nobody pushed still hold their initial values;
  * after step 1, every FC tensor of rank 0 against Adam on the rank-order sum of the eight workers' dense gradients / 8;
  * after 4 pipelined steps: no error, no device-side wait timed out, joins by device flags on every rank, no list
    outgrew its wire block, the replicated tensors (FC, wide) bit-identical on the 8 ranks;
  * the exchange's sizes: id blocks <= 0.8 MB per rank and step (round 3: 2.98 MB).

Round 5: configs[4] the same way -- "multi-hot variable-length bags (avg 30 ids/field), fused FTRL
sparse update, 8 GPUs async push (-DisAsync)": 8 rank processes, Poisson(30) bags (~3.2 M ids per rank and step), FTRL on
the embedding rows, is_async = 1 (net/PServer.java:176-184: kvStore.sum + update(updater, key) PER PUSH, no averaging,
arrival = rank order), own keys in place + poisoned self blocks.  After step 1 every row (w, z, n) of ranks 0 and 5 in
three fields is bit-equal to the oracle's ftrl_update applied per push in rank order to the workers' per-key gradients."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
f32 = np.float32
TEST_TIMEOUT_S = 420            # tests/conftest.py: eight processes, each with torch + a full-size shard, on a possibly cold box
CHECK_FIELDS = (0, 7, 25)
CHECK_RANKS = (0, 5)


C4_MAX_NNZ = 4096 * 26 * 31          # Poisson(30) bags: 3.195 M ids expected per batch, sd ~1 800 (every rank sizes its blocks from this)


def c4_batch(cfg, rng):
    """bench.py multi_hot_step's batch: bag lengths Poisson(30) clipped to [1, 100] per (sample, field), ids
    Zipf(1.05) over V, wide ids uniform"""
    from ps_amd import synth
    B, F, V = cfg["B"], cfg["F"], cfg["V"]
    lens = np.clip(rng.poisson(30, size=B * F), 1, 100)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = synth.draw_ids(rng, 1.05, V, int(offsets[-1]), "zipf_truncated")
    W = rng.integers(0, cfg["wide"], size=(B, F)).astype(np.int64)
    return ids, rng.standard_normal((B, cfg["X"])).astype(f32), (rng.random(B) < 0.25).astype(f32), W, offsets


def rank_process(rank, world, port, steps, q, snapshot=True, case="c2", tune=None):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
        import ctypes as C, hashlib, time
        import torch, torch.distributed as dist
        import ps_amd
        from ps_amd import native as N
        from ps_amd.sharded import NativeWorker
        from bench import C2, synth_batch
        t0 = time.time()
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        cfg = dict(C2)
        c4 = case == "c4"
        kv = ps_amd.KVStore(0, cfg["seed"])
        kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"], shard=rank, nshards=world)
        if c4:
            kv.set_updater("emF", ps_amd.FtrlUpdater())
            gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"], max_nnz=C4_MAX_NNZ)
        else:
            gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
        L = N.lib()
        for k_, v_ in (tune or {}).items():          # (mapped_peer: rows and gradients through the peers' mapped memory, ps_native.h)
            N.check(L.ps_tune_set(k_.encode(), int(v_)))

        class GlooOps:
            """ps_comm_ops_t over gloo, host staged; own keys in place: the self part of a receive buffer is poisoned"""
            def __init__(self):
                self.ops = N.ps_comm_ops_t(); self.ops.ctx, self.ops.nranks, self.ops.rank = None, world, rank
                self._k = (N.ALL_GATHER_FN(self.ag), N.ALL_TO_ALL_V_FN(self.a2a), N.ALL_REDUCE_FN(self.ar))
                self.ops.all_gather, self.ops.all_to_all_v, self.ops.all_reduce_sum_f32 = self._k
                self.ops.flags = N.PS_COMM_OWN_IN_PLACE
                self.err, self.checking = None, False
            def _down(self, p, n):
                a = np.empty(n, np.uint8)
                if n: N.check(L.ps_dev_download(kv.h, a.ctypes.data, p, n))
                return a
            def _up(self, p, a):
                if a.size: a = np.ascontiguousarray(a); N.check(L.ps_dev_upload(kv.h, p, a.ctypes.data, a.nbytes))
            def _g(self, fn, stream):
                try:
                    N.check(L.ps_stream_sync(kv.h, stream)); fn(); return 0
                except BaseException as e:      # noqa: BLE001
                    self.err = e; return 500
            def ag(self, ctx, send, recv, nb, stream):
                def f():
                    m = torch.from_numpy(self._down(send, nb)); parts = [torch.empty_like(m) for _ in range(world)]
                    dist.all_gather(parts, m); self._up(recv, torch.cat(parts).numpy())
                return self._g(f, stream)
            def a2a(self, ctx, send, sc, recv, rc, eb, stream):
                def f():
                    scl = [int(sc[i]) * eb for i in range(world)]; rcl = [int(rc[i]) * eb for i in range(world)]
                    out = torch.empty(sum(rcl), dtype=torch.uint8)
                    dist.all_to_all_single(out, torch.from_numpy(self._down(send, sum(scl))), output_split_sizes=rcl, input_split_sizes=scl)
                    o = out.numpy()
                    if not self.checking: o[sum(rcl[:rank]):sum(rcl[:rank + 1])] = 0xFF
                    self._up(recv, o)
                return self._g(f, stream)
            def ar(self, ctx, buf, n, stream):
                def f():
                    m = torch.from_numpy(self._down(buf, n * 4).view(f32)); parts = [torch.empty_like(m) for _ in range(world)]
                    dist.all_gather(parts, m)
                    tot = parts[0].numpy().copy()
                    for p in parts[1:]: tot = (tot + p.numpy()).astype(f32)          # rank order
                    self._up(buf, tot)
                return self._g(f, stream)

        comm = GlooOps()
        wk = NativeWorker([gm], world, rank, ops=comm.ops, is_async=c4)
        comm.checking = True; wk.selfcheck(); comm.checking = False
        rng = np.random.default_rng(cfg["seed"] + 1000 * rank)
        bs = [ps_amd.DeviceBatch(kv, *(c4_batch(cfg, rng) if c4 else synth_batch(cfg, rng))) for _ in range(3 if c4 else 4)]
        snap = None
        wk.run(bs, 1)                       # step 1 on its own: its result is checked key by key
        kv.sync()
        if comm.err is not None: raise comm.err
        if snapshot and rank in CHECK_RANKS:
            ids = np.arange(rank, cfg["V"], world)
            snap = {"rows": {f: kv.get_rows(f, ids) for f in CHECK_FIELDS}}
            if c4:
                snap["z"] = {f: kv.get_rows(f, ids, 1) for f in CHECK_FIELDS}; snap["n"] = {f: kv.get_rows(f, ids, 2) for f in CHECK_FIELDS}
            if rank == 0:
                snap["fcW"] = [kv.get("fc%d.weights" % l) for l in range(3)]; snap["fcb"] = [kv.get("fc%d.bias" % l) for l in range(3)]
                wid = np.arange(cfg["wide"])       # (w, z, n): Ftrl's w lags z and n by one update -- after step 1 the state is what moved
                snap["wide"] = [kv.get_wide(wid, k) for k in range(3)]
        wk.run(bs[1:] + bs[:1], steps - 1)  # ... and the pipeline: begin of step t + 1 inside finish of step t
        kv.sync()
        if comm.err is not None: raise comm.err
        loss = wk.step(bs[0], want_loss=True)
        st = (C.c_int64 * 10)(); N.check(L.ps_shard_exchange_stats(gm.h, st, 10))
        why = C.create_string_buffer(256)
        mode = L.ps_store_join_mode(kv.h, why, 256)
        mp5 = (C.c_int64 * 5)(); N.check(L.ps_shard_mapped_info(gm.h, mp5))
        h = hashlib.sha256()
        for l in range(3): h.update(kv.get("fc%d.weights" % l).tobytes()); h.update(kv.get("fc%d.bias" % l).tobytes())
        h.update(kv.get_wide(np.arange(cfg["wide"])).tobytes())
        dist.barrier()
        q.put((rank, "ok", dict(loss=float(loss), digest=h.hexdigest(), stats=[int(x) for x in st], join_mode=mode, why=why.value.decode(),
                                timeouts=int(L.ps_store_wait_timeouts(kv.h)), seconds=time.time() - t0, snap=snap, mapped=[int(x) for x in mp5])))
        gm.close(); kv.close(); dist.destroy_process_group()
    except BaseException:       # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))


def run_ranks(world, steps, snapshot=True, timeout=600, case="c2", tune=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=rank_process, args=(r, world, port, steps, q, snapshot, case, tune), daemon=True) for r in range(world)]
    for p in ps: p.start()
    try:
        res = [q.get(timeout=timeout) for _ in ps]
    finally:
        for p in ps:
            p.join(30)
            if p.is_alive(): p.kill()           # exactly the processes started here
    res.sort(key=lambda r: r[0])
    return res


def ps_semantics_after_one_step(orc, world):
    """net/PServer.java:164-214 on the eight workers' first batches, from the initial parameters: per key the mean, in worker
    order, over the workers that pushed it + one Adam step; dense tensors: the rank-order sum / workers + one Adam step.
    The workers' gradients come from the single-GPU split step on an UNSHARDED store (same init: a pure function of
    (seed, field, id, column)).  Returns rows[f] = (ids pushed by anyone, their expected rows), W0[f] (every initial row of
    the checked fields), and the expected FC tensors."""
    import ps_amd
    from bench import C2, synth_batch
    cfg = dict(C2)
    F, D, V = cfg["F"], cfg["D"], cfg["V"]
    kv = ps_amd.KVStore(0, cfg["seed"])
    kv.create_embedding([V] * F, D)
    gm = ps_amd.WideDeepNN.buildModel(F, D, cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"])
    W0 = {f: kv.get_rows(f, np.arange(V)) for f in CHECK_FIELDS}
    fc0 = [(kv.get("fc%d.weights" % l), kv.get("fc%d.bias" % l)) for l in range(3)]
    pushes = {f: [] for f in CHECK_FIELDS}
    dW = [None] * 3; db = [None] * 3
    gbar, wmask = [], np.zeros(cfg["wide"], np.int64)
    for w in range(world):
        rng = np.random.default_rng(cfg["seed"] + 1000 * w)
        E, X, Y, Wd = synth_batch(cfg, rng)
        gm.forward({"E": E, "X": X, "Y": Y, "W": Wd})
        gm.backward()
        # the worker's wide push (layer/LRLayer.java:106-117): every key it has touched gets rowMeans(delta) -- summed the way k_loss_reduce
        # sums (1024 strided partial sums, a halving tree), / B
        P = np.asarray(gm.p(cfg["B"]), f32); Yf = np.asarray(Y, f32).reshape(-1)
        t = (P * (f32(1) - P)).astype(f32)
        d = (((P - Yf).astype(f32) / t).astype(f32) * t).astype(f32)       # loss/CrossEntropy.java:25, activations/Sigmoid.java:18 (kernels_head.inc)
        part = np.zeros(1024, f32)
        for i in range(0, len(d), 1024):
            seg = d[i:i + 1024]; part[:len(seg)] = (part[:len(seg)] + seg).astype(f32)
        off = 512
        while off:
            part[:off] = (part[:off] + part[off:2 * off]).astype(f32); off >>= 1
        gbar.append(f32(part[0] / f32(len(d))))
        wmask[np.unique(Wd)] |= 1 << w
        for f in CHECK_FIELDS:
            pushes[f].append(gm.emb_grads(f))
        for l in range(3):
            gw, gb = gm.fc_grad(l), gm.fc_grad(l, bias=True)
            dW[l] = gw if dW[l] is None else (dW[l] + gw).astype(f32)
            db[l] = gb if db[l] is None else (db[l] + gb).astype(f32)
    rows = {}
    for f in CHECK_FIELDS:
        ids = np.unique(np.concatenate([p[0] for p in pushes[f]]))
        S = np.zeros((len(ids), D), f32); cnt = np.zeros(len(ids), f32)
        for pid, g in pushes[f]:                                   # worker order
            ix = np.searchsorted(ids, pid)
            S[ix] = (g + S[ix]).astype(f32); cnt[ix] += 1
        mean = (S / cnt[:, None]).astype(f32)
        z = np.zeros(mean.size, f32)
        w1, _, _ = orc.adam_update(W0[f][ids].reshape(-1), mean.reshape(-1), z, z.copy())
        rows[f] = (ids, w1.reshape(-1, D))
    fc1 = []
    for l in range(3):
        gw = (dW[l] / f32(world)).astype(f32); gb = (db[l] / f32(world)).astype(f32)
        fc1.append((orc.adam_update(fc0[l][0], gw, np.zeros_like(gw), np.zeros_like(gw))[0],
                    orc.adam_update(fc0[l][1], gb, np.zeros_like(gb), np.zeros_like(gb))[0]))
    # the wide table (w = z = n = 0 before the step): per key the mean, in worker order, over the workers that pushed it -- what the
    # all-reduced worker slots are turned into (kernels_emb.h WideUpdArgs.slots) -- then one Ftrl step (update/FtrlUpdater.java:51-76)
    wide1 = [np.zeros(cfg["wide"], f32) for _ in range(3)]
    for m in np.unique(wmask):
        if m == 0: continue
        ws = [w for w in range(world) if (int(m) >> w) & 1]
        G = gbar[ws[0]]
        for w in ws[1:]: G = f32(gbar[w] + G)
        g = f32(G / f32(len(ws)))
        wzn = orc.ftrl_update(np.zeros(1, f32), np.array([g], f32), np.zeros(1, f32), np.zeros(1, f32))
        for k in range(3): wide1[k][wmask == m] = wzn[k][0]
    wbias1 = None
    gm.close(); kv.close()
    return rows, W0, fc1, wide1, wbias1


MAPPED = {"mapped_peer": 1, "spin_timeout_ms": 30000}       # (eight processes share one GPU: a peer's launch may be a time slice away)


@pytest.mark.parametrize("mapped", [False, True])
def test_config2_full_size_eight_ranks_on_one_gpu(orc, mapped):
    """mapped: the rows / gradient exchanges as stores into the seven peer PROCESSES' mapped memory (round 6) -- the same bits"""
    world, steps = 8, 4
    res = run_ranks(world, steps, timeout=330, tune=MAPPED if mapped else None)
    bad = [r for r in res if r[1] != "ok"]
    assert not bad, "\n".join("rank %d:\n%s" % (r[0], r[2]) for r in bad)
    info = [r[2] for r in res]
    # the pipeline ran clean on every rank
    for r, i in enumerate(info):
        assert i["mapped"][0] == (1 if mapped else 0) and (not mapped or (i["mapped"][2] == steps + 1 and i["mapped"][3] == steps + 1)), i["mapped"]
        assert i["timeouts"] == 0 and i["join_mode"] == 1 and i["why"] == "", (r, i["why"], i["timeouts"])
        st = i["stats"]
        assert st[0] == steps + 1 and st[8] == 0, st                       # steps counted; no list outgrew its wire block
        per_step_blocks = st[1] / st[0]
        assert per_step_blocks <= 0.8e6, "id blocks: %.0f B per rank and step (round 3: 2 983 680)" % per_step_blocks
        assert 40000 < st[5] / st[0] < 52000 and 40000 < st[6] / st[0] < 52000, st      # ~46 k unique keys requested / served
        assert np.isfinite(i["loss"]) and 0.2 < i["loss"] < 20, i["loss"]
    assert all(i["digest"] == info[0]["digest"] for i in info), "replicated tensors differ across ranks"
    # step 1 against the parameter-server semantics, bit for bit
    rows, W0, fc1, wide1, wbias1 = ps_semantics_after_one_step(orc, world)
    pushed = 0
    for r in CHECK_RANKS:
        got = info[r]["snap"]["rows"]
        own = np.arange(r, 100000, world)
        for f in CHECK_FIELDS:
            want = W0[f][own].copy()                                       # rows nobody pushed: initial values
            ids, w1 = rows[f]
            mine = ids % world == r
            want[(ids[mine] - r) // world] = w1[mine]
            np.testing.assert_array_equal(got[f], want, err_msg="rank %d, field %d" % (r, f))
            pushed += int(mine.sum())
    assert pushed > 3000
    for l in range(3):
        np.testing.assert_array_equal(info[0]["snap"]["fcW"][l], fc1[l][0], err_msg="fc%d.weights after step 1" % l)
        np.testing.assert_array_equal(info[0]["snap"]["fcb"][l], fc1[l][1], err_msg="fc%d.bias after step 1" % l)
    # the wide table and its bias after step 1 (replicated: rank 0's copy), every one of the 100 000 keys
    assert (wide1[1] != 0).sum() > 50000 and (wide1[2] != 0).sum() > 50000
    for k, what in enumerate(("w", "z", "n")):
        np.testing.assert_array_equal(info[0]["snap"]["wide"][k], wide1[k], err_msg="wide table after step 1: %s" % what)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rehearse_n8%s.log" % ("_mapped" if mapped else "")), "w") as fo:
            for r, i in enumerate(info):
                st = i["stats"]; n = st[0]
                fo.write("rank %d: loss %.5f  joins %s  timeouts %d  per step: %d keys requested, %d served, id blocks %d B (wire block %d words, full %d), rows %d B, "
                         "gradients %d B, all-reduce %d B, full-size id exchanges %d  (%.1f s)\n" % (
                             r, i["loss"], "flags" if i["join_mode"] == 1 else "events", i["timeouts"], st[5] // n, st[6] // n, st[1] // n, st[7], st[9],
                             st[2] // n, st[3] // n, st[4] // n, st[8], i["seconds"]))
            fo.write("8 ranks x %d steps at configs[2] size: replicated tensors bit-identical; %d pushed rows of ranks %s and rank 0's FC tensors equal the "
                     "PS semantics after step 1 bit for bit, and so do the (w, z, n) of all %d wide keys (rank-order mean over the workers that touched a key, Ftrl)\n" % (steps + 1, pushed, list(CHECK_RANKS), len(wide1[0])))
    except OSError:
        pass


def ps_async_semantics_after_one_step(orc, world):
    """net/PServer.java:176-184 (-DisPsAsync=1) on the eight workers' first configs[4] batches, from the initial parameters:
    every push is kvStore.sum + update(updater, key) on arrival -- no averaging -- and arrival order is rank order; the
    updater of the embedding rows is Ftrl (update/FtrlUpdater.java:51-76, w lags z and n by one update, a gradient whose
    first component is 0 is skipped).  The workers' per-key gradients come from the single-GPU split step on an unsharded
    store (verified against the oracle in test_gpu_configs.py::test_config4_*).  Returns per checked field the pushed ids
    with their expected (w, z, n), the initial rows, and the expected FC tensors (dense: rank-order sum / workers, Adam)."""
    import ps_amd
    from bench import C2
    cfg = dict(C2)
    F, D, V = cfg["F"], cfg["D"], cfg["V"]
    kv = ps_amd.KVStore(0, cfg["seed"])
    kv.create_embedding([V] * F, D)
    kv.set_updater("emF", ps_amd.FtrlUpdater())
    gm = ps_amd.WideDeepNN.buildModel(F, D, cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=cfg["B"], max_nnz=C4_MAX_NNZ)
    W0 = {f: kv.get_rows(f, np.arange(V)) for f in CHECK_FIELDS}
    fc0 = [(kv.get("fc%d.weights" % l), kv.get("fc%d.bias" % l)) for l in range(3)]
    pushes = {f: [] for f in CHECK_FIELDS}
    dW = [None] * 3; db = [None] * 3
    nnz = []
    for w in range(world):
        rng = np.random.default_rng(cfg["seed"] + 1000 * w)
        ids, X, Y, Wd, offsets = c4_batch(cfg, rng)
        nnz.append(ids.size)
        gm.forward({"E": ids, "X": X, "Y": Y, "W": Wd, "offsets": offsets})
        gm.backward()
        for f in CHECK_FIELDS:
            pushes[f].append(gm.emb_grads(f))
        for l in range(3):
            gw, gb = gm.fc_grad(l), gm.fc_grad(l, bias=True)
            dW[l] = gw if dW[l] is None else (dW[l] + gw).astype(f32)
            db[l] = gb if db[l] is None else (db[l] + gb).astype(f32)
    rows = {}
    for f in CHECK_FIELDS:
        ids = np.unique(np.concatenate([p[0] for p in pushes[f]]))
        w = W0[f][ids].copy(); z = np.zeros_like(w); n = np.zeros_like(w)
        npush = np.zeros(len(ids), np.int64)
        for pid, g in pushes[f]:                                   # arrival = worker order; one Ftrl step per push
            ix = np.searchsorted(ids, pid)
            # the oracle's ftrl_update takes ONE key's vectors (its "dw[0] == 0 -> skip" is per key): all keys of this push
            # in one call as one long vector, then the skipped keys put back
            w1, z1, n1, _ = orc.ftrl_update(w[ix].reshape(-1), g.reshape(-1), z[ix].reshape(-1), n[ix].reshape(-1))
            w1 = w1.reshape(-1, D); z1 = z1.reshape(-1, D); n1 = n1.reshape(-1, D)
            skip = g[:, 0] == 0
            w1[skip] = w[ix][skip]; z1[skip] = z[ix][skip]; n1[skip] = n[ix][skip]
            w[ix] = w1; z[ix] = z1; n[ix] = n1
            npush[ix] += 1
        rows[f] = (ids, w, z, n, npush)
    fc1 = []
    for l in range(3):
        gw = (dW[l] / f32(world)).astype(f32); gb = (db[l] / f32(world)).astype(f32)
        fc1.append((orc.adam_update(fc0[l][0], gw, np.zeros_like(gw), np.zeros_like(gw))[0],
                    orc.adam_update(fc0[l][1], gb, np.zeros_like(gb), np.zeros_like(gb))[0]))
    gm.close(); kv.close()
    return rows, W0, fc1, nnz


@pytest.mark.parametrize("mapped", [False, True])
def test_config4_full_size_eight_ranks_async_ftrl_on_one_gpu(orc, mapped):
    world, steps = 8, 3
    res = run_ranks(world, steps, timeout=380, case="c4", tune=MAPPED if mapped else None)
    bad = [r for r in res if r[1] != "ok"]
    assert not bad, "\n".join("rank %d:\n%s" % (r[0], r[2]) for r in bad)
    info = [r[2] for r in res]
    for r, i in enumerate(info):
        assert i["mapped"][0] == (1 if mapped else 0) and (not mapped or (i["mapped"][2] == steps + 1 and i["mapped"][3] == steps + 1)), i["mapped"]
        assert i["timeouts"] == 0 and i["join_mode"] == 1 and i["why"] == "", (r, i["why"], i["timeouts"])
        st = i["stats"]
        assert st[0] == steps + 1 and st[8] == 0, st                       # steps counted; no list outgrew its wire block
        assert 500000 < st[5] / st[0] < 720000 and 500000 < st[6] / st[0] < 720000, st      # ~617 k unique rows requested / served
        assert np.isfinite(i["loss"]) and 0.2 < i["loss"] < 20, i["loss"]
    assert all(i["digest"] == info[0]["digest"] for i in info), "replicated tensors differ across ranks"
    rows, W0, fc1, nnz = ps_async_semantics_after_one_step(orc, world)
    assert all(3.0e6 < x <= C4_MAX_NNZ for x in nnz), nnz
    pushed = many = 0
    for r in CHECK_RANKS:
        snap = info[r]["snap"]
        own = np.arange(r, 100000, world)
        for f in CHECK_FIELDS:
            ids, w1, z1, n1, npush = rows[f]
            mine = ids % world == r
            at = (ids[mine] - r) // world
            want_w = W0[f][own].copy(); want_z = np.zeros_like(want_w); want_n = np.zeros_like(want_w)     # rows nobody pushed: untouched
            want_w[at] = w1[mine]; want_z[at] = z1[mine]; want_n[at] = n1[mine]
            np.testing.assert_array_equal(snap["rows"][f], want_w, err_msg="rank %d, field %d: w" % (r, f))
            np.testing.assert_array_equal(snap["z"][f], want_z, err_msg="rank %d, field %d: z" % (r, f))
            np.testing.assert_array_equal(snap["n"][f], want_n, err_msg="rank %d, field %d: n" % (r, f))
            pushed += int(mine.sum()); many += int((npush[mine] == world).sum())
    assert pushed > 20000 and many > 1000, (pushed, many)      # rows pushed by all eight workers exist: the per-push order matters
    for l in range(3):
        np.testing.assert_array_equal(info[0]["snap"]["fcW"][l], fc1[l][0], err_msg="fc%d.weights after step 1" % l)
        np.testing.assert_array_equal(info[0]["snap"]["fcb"][l], fc1[l][1], err_msg="fc%d.bias after step 1" % l)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rehearse_c4_n8%s.log" % ("_mapped" if mapped else "")), "w") as fo:
            for r, i in enumerate(info):
                st = i["stats"]; n = st[0]
                fo.write("rank %d: loss %.5f  joins %s  timeouts %d  per step: %d keys requested, %d served, id blocks %d B (wire block %d words, full %d), rows %d B, "
                         "gradients %d B, all-reduce %d B, steps with the full-size id exchange (a list overflowed its wire block) %d  (%.1f s)\n" % (
                             r, i["loss"], "flags" if i["join_mode"] == 1 else "events", i["timeouts"], st[5] // n, st[6] // n, st[1] // n, st[7], st[9],
                             st[2] // n, st[3] // n, st[4] // n, st[8], i["seconds"]))
            fo.write("8 ranks x %d steps at configs[4] size (%d..%d ids per rank and step, Ftrl rows, async push): replicated tensors bit-identical; %d pushed rows "
                     "(w, z, n) of ranks %s in fields %s -- %d of them pushed by all 8 workers -- and rank 0's FC tensors equal net/PServer.java:176-184's "
                     "per-push semantics after step 1 bit for bit\n" % (steps + 1, min(nnz), max(nnz), pushed, list(CHECK_RANKS), list(CHECK_FIELDS), many))
    except OSError:
        pass
