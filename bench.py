#!/usr/bin/env python
"""bench.py -- Wide&Deep training examples/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (EmbeddingField lookup -> FcLayer
forward/backward -> fused Adam/Ftrl sparse update) over one synthetic
Criteo-shaped batch, inputs already resident in HBM.  N=1 runs BASELINE
configs[1]: 26 sparse fields x 100k vocab x 16-dim, 13 dense, FC[512,256,1],
batch 4096, WideDeepNN with wideSize 100000.  For N>1 the driver launches one
rank per GPU with torch.distributed.run; rows are hash-sharded across ranks
(configs[2]) -- see ps_amd/sharded.py.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0   # MI355X_MICROARCH.md: measured device-to-device copy ceiling (read + write)
F32_MFMA_PEAK_TFS = 157.3  # MI355X_MICROARCH.md: dense FP32 MFMA peak (exact f32; no TF32 on gfx950)

C2 = dict(F=26, V=100000, D=16, X=13, fc=[512, 256, 1], B=4096, wide=100000, zipf=1.05, seed=0x5EED, idgen="zipf_truncated")


def synth_batch(cfg, rng, B=None):
    """ids ~ Zipf(1.05) over V per field (the TRUNCATED law over ranks 1..V, drawn by inverse CDF:
    ps_amd/synth.py), dense ~ N(0,1), labels ~ Bernoulli(0.25), wide ids = id mod wideSize (CTR.java:65,
    MatrixUtil.hash).  cfg["idgen"]: "zipf_truncated" (default), "zipf_clamped" (rounds 1-2: unbounded Zipf
    clamped to V - 1, 55 % of the draws on one row per field), "uniform" (SURVEY 8d's variant)."""
    from ps_amd import synth
    B = B or cfg["B"]
    gen = cfg.get("idgen", "zipf_truncated") if cfg["zipf"] > 1.0 else "uniform"
    E = synth.draw_ids(rng, cfg["zipf"], cfg["V"], (B, cfg["F"]), gen)
    X = rng.standard_normal((B, cfg["X"])).astype(np.float32)
    Y = (rng.random(B) < 0.25).astype(np.float32)
    return E, X, Y, E % cfg["wide"]


def fc_flops_per_step(cfg):
    # fwd 2*B*sum(in*out), bwd 4*B*sum(in*out)
    dims = [cfg["F"] * cfg["D"] + cfg["X"]] + cfg["fc"]
    s = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    return 6.0 * cfg["B"] * s, dims


# kernel launches behind one profiled group (the others are one launch): per-launch time decides the dominant KERNEL
# launches per profiled group.  emb_sort: ONE launch for single-hot batches (field sort: pairs, segments and the long-run
# list); multi-hot batches go through the general chain (9 radix launches + the 2 of emb_segments).
GROUP_LAUNCHES = {"emb_sort": 1, "emb_segments": 2}


def tn_splits(cfg):
    """Split-K factors of the dW GEMMs (kernels_gemm.hip gemm_tn_choose_split: ~224 workgroups of 64 x 64 tiles,
    at least 128 batch rows per split); the out = 1 layer goes through k_last_bwd instead."""
    dims = [cfg["F"] * cfg["D"] + cfg["X"]] + cfg["fc"]
    out = []
    for a, b in zip(dims[:-1], dims[1:]):
        if b == 1:
            out.append(1)               # folded to one slab by k_dense_prereduce
            continue
        tiles = -(-(a + 1) // 64) * -(-b // 64) if b > 32 else -(-(a + 1) // 128) * -(-b // 32)
        out.append(max(1, min(-(-224 // tiles), max(1, cfg["B"] // 128), 64)))
    return out


def group_algorithmic(cfg, name, nnz, uniq):
    """Algorithmic work of one launch group (bytes for HBM-bound, flop for MFMA-bound), SURVEY 8d's per-unit figures.
    Every group of the step is a candidate for the dominant kernel."""
    B, D, F = cfg["B"], cfg["D"], cfg["F"]
    dims = [F * D + cfg["X"]] + cfg["fc"]
    dense = sum((a + 1) * b for a, b in zip(dims[:-1], dims[1:]))
    if name == "emb_fwd":
        return "hbm", nnz * (4.0 * D + 8.0)                       # row + int64 id (read roofline)
    if name == "emb_bwd_update":
        return "hbm", nnz * 4.0 * D + uniq * 6 * 4.0 * D + nnz * 8.0     # delta rows + {W,M,V} read and written per key + sort pairs
    if name == "emb_sort":
        if cfg.get("multi_hot"):
            return "hbm", 3 * 24.0 * nnz                          # 3 radix passes: pairs read twice (histogram, scatter), written once
        return "hbm", 20.0 * nnz                                  # field sort: keys read; sorted keys, entries, run ids, run starts written
    if name == "emb_segments":
        return "hbm", 20.0 * nnz                                  # keys read twice, run ids written
    if name == "dense_update":
        slabs = sum(s * (a + 1) * b for s, a, b in zip(tn_splits(cfg), dims[:-1], dims[1:]))
        return "hbm", 4.0 * (slabs + dense * 8)                   # the split slabs + W/M/V read, W/Wt/M/V and the flat gradient written
    if name == "head_last_bwd":
        return "hbm", 4.0 * B * (3 * dims[-2] + 2 * F + 8)        # the out = 1 layer's input twice, delta_prev written, wide ids
    if name in ("head", "fc_bwd_last"):
        return "hbm", 4.0 * B * (2 * dims[-2] + 2 * F + 8)
    if name == "wide_update":
        return "hbm", 13.0 * cfg["wide"]
    if name == "loss_reduce":
        return "hbm", 8.0 * B
    if name == "fc_fwd01":                                        # the first two forward GEMMs in one launch (k_fc_fwd_pair)
        return "mfma", 2.0 * B * (dims[0] * dims[1] + dims[1] * dims[2])
    for l in range(len(cfg["fc"])):
        if name == "fc_fwd%d" % l:
            return "mfma", 2.0 * B * dims[l] * dims[l + 1]
        if name == "fc_bwd_dw%d" % l:
            return "mfma", 2.0 * B * dims[l] * dims[l + 1]
        if name == "fc_bwd_data%d" % l:
            return "mfma", 2.0 * B * (dims[l] if l > 0 else F * D) * dims[l + 1]
    return None, 0.0


def kernel_of_group(name):
    """The device function a kernel group of the step launches (names as in rocprofv3's kernel stats)."""
    if name == "fc_fwd01" or name == "fc_fwd_pair":
        return "k_fc_fwd_pair"
    if name.startswith("fc_fwd") or name.startswith("fc_bwd_data"):
        return "k_gemm_nt"
    if name.startswith("fc_bwd_dw"):
        return "k_gemm_tn"
    return {"emb_bwd_update": "k_emb_reduce_update", "emb_fwd": "k_emb_fwd", "emb_sort": "k_field_sort_segments",
            "head_last_bwd": "k_last_bwd", "dense_update": "k_dense_update"}.get(name, name)


def host_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def cpu_baseline(cfg, budget_s=15.0):
    """The restated reference CPU path (oracle, string keys + hash maps, thread = 1 as CTR.java:72
    forces) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    rng = np.random.default_rng(cfg["seed"])
    st = orc.Store(cfg["seed"])
    om = orc.Model(st, orc.WIDEDEEP, cfg["F"], cfg["D"], cfg["X"], cfg["fc"], wide_size=cfg["wide"])
    n, t_total = 0, 0.0
    E, X, Y, W = synth_batch(cfg, rng)
    om.train(E.astype(np.float32), X, Y, W.astype(np.float32))          # warm-up (creates keys)
    while t_total < budget_s and n < 6:
        E, X, Y, W = synth_batch(cfg, rng)
        t0 = time.perf_counter()
        om.train(E.astype(np.float32), X, Y, W.astype(np.float32))
        t_total += time.perf_counter() - t0
        n += 1
    return {"value": cfg["B"] * n / t_total, "unit": "examples/s", "cores": 1, "kind": "port",
            "sample": "%d Wide&Deep steps of batch %d (same synthetic config) after 1 warm-up, oracle/ps_oracle.c "
                      "restatement in compat mode (string keys, hash maps, thread=1 as CTR.java:72)" % (n, cfg["B"])}


def cpu_baseline_threads(cfg, threads, steps=2):
    """The reference's thread-DP (train/Trainer.java:77-95: T replicas, one minibatch each per round) as an UPPER BOUND:
    T independent replicas of the restated step, one host thread each (the C oracle releases the GIL), with none of the
    serialisation the reference adds on top (every kvStore.sum / get goes through one monitor, store/KVStore.java:136,192)."""
    import threading
    from oracle import oracle as orc

    def work(i, out):
        rng = np.random.default_rng(cfg["seed"] + 17 * i)
        st = orc.Store(cfg["seed"])
        om = orc.Model(st, orc.WIDEDEEP, cfg["F"], cfg["D"], cfg["X"], cfg["fc"], wide_size=cfg["wide"])
        bs = [synth_batch(cfg, rng) for _ in range(steps + 1)]
        E, X, Y, W = bs[0]
        om.train(E.astype(np.float32), X, Y, W.astype(np.float32))      # warm-up
        gate.wait()
        t0 = time.perf_counter()
        for E, X, Y, W in bs[1:]:
            om.train(E.astype(np.float32), X, Y, W.astype(np.float32))
        out[i] = (t0, time.perf_counter())

    gate = threading.Barrier(threads)
    out = [None] * threads
    th = [threading.Thread(target=work, args=(i, out)) for i in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(o[1] for o in out) - min(o[0] for o in out)
    return {"value": cfg["B"] * steps * threads / wall, "unit": "examples/s", "cores": threads, "kind": "port",
            "sample": "%d independent replicas x %d steps of batch %d, one host thread each: an upper bound of the reference's "
                      "thread-DP (Trainer.java:77-95), which serialises kvStore.sum/get on one monitor" % (threads, steps, cfg["B"])}


def run_single(args):
    import ps_amd
    cfg = dict(C2)
    cfg["zipf"] = args.zipf
    cfg["idgen"] = args.idgen
    from ps_amd import synth
    rng = np.random.default_rng(cfg["seed"])
    kv = ps_amd.KVStore(0, cfg["seed"])
    kv.create_embedding([cfg["V"]] * cfg["F"], cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(cfg["F"], cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv,
                                      max_batch=cfg["B"], use_graph=int(args.graph))
    # 64 resident batches (262 144 samples): the labels are independent of the features, so the model can only memorise;
    # with a handful of batches it does within ~1500 steps, the loss falls under the reference's stop threshold
    # (model/DNN.java:58-63: loss <= 0.01 -> no backward) and the step would silently get cheaper
    nb = min(4096, max(64, (2 * args.steps + args.warmup + args.priming + (650 if args.steps < 1000 else 50)) // 8))   # a batch is seen at most ~8 times
    batches, stats = [], []
    for _ in range(nb):
        E, X, Y, W = synth_batch(cfg, rng)
        batches.append(ps_amd.DeviceBatch(kv, E, X, Y, W))
        if len(stats) < 16:
            stats.append(synth.id_stats(E))
    uniq = int(round(np.mean([s_["unique_keys"] for s_ in stats])))      # unique (field, id) keys of a batch, mean of 16
    hottest = int(max(s_["hottest_run"] for s_ in stats))               # longest run of one key in one field
    nnz = cfg["B"] * cfg["F"]
    for i in range(3):                            # (module load, first launches)
        gm.train_async(batches[i % nb])
    gm.sync()
    # short pass with every kernel group bracketed by HIP events (one stream): finds the dominant kernel
    PROFILE_STEPS = 20
    gm.set_profile(True)
    for i in range(PROFILE_STEPS):
        gm.train_async(batches[i % nb])
    gm.sync()
    prof = gm.profile_report()
    gm.set_profile(False)
    # ---- a short region right behind an idle queue (what round 4's line timed as its headline): 5 steps, a wait, 20 steps.
    # The GPU's clocks follow its load: tools/ramp_probe.py / ramp_pre.py (profiles/r05_clock_ramp.txt) -- 142 us per step in
    # the first 20 steps behind an idle queue, 132 from step ~120 on; behind 300 steps of the same work 0.1345 ms, behind
    # 28 ms of HBM-bound gathers 0.140, behind nothing 0.1435: the same kernels on a ramping MFMA clock.
    from_idle = None
    if args.priming > 0:
        time.sleep(0.3)
        for i in range(5):
            gm.train_async(batches[i % nb])
        gm.sync()
        t0i = time.perf_counter()
        for i in range(20):
            gm.train_async(batches[(5 + i) % nb])
        gm.sync()
        from_idle = {"steps": 20, "ms_per_step": 1e3 * (time.perf_counter() - t0i) / 20,
                     "note": "5 steps, a 0.3 s idle queue, then 20 timed steps: the step on ramping clocks (round 4's headline sat behind 48 steps instead and read 0.1392 ms)"}
    # ---- priming (untimed, disclosed in config; the sharded line has primed the same way since round 2): the training job this line
    # stands for runs for millions of steps -- the timed region starts on the clocks it would run on
    for i in range(args.priming):
        gm.train_async(batches[i % nb])
    if args.priming:
        gm.sync()       # (a wait between the priming and the warm-up: a short region right behind a LONG asynchronous run pays the HIP
                        #  runtime's housekeeping on the host -- 0.18 ms per step for 20 steps behind 305; tools/shard_short_run.py, round 4)
    for i in range(max(args.warmup, 1)):          # the W untimed warm-up steps
        gm.train_async(batches[i % nb])
    gm.sync()
    # ---- the timed region: K steps, nothing else on the stream ----
    t0 = time.perf_counter()
    for i in range(args.steps):
        gm.train_async(batches[i % nb])
    gm.sync()
    dt = time.perf_counter() - t0
    # ---- roofline of the dominant kernel: same steps again with HIP events around every launch of that kernel ----
    # dominant KERNEL = the kernel (device function) whose launches take the most time per step -- what tops rocprofv3's
    # per-kernel stats table of the same command (profiles/r02_c2_bench_kernel_stats.csv).  A kernel launched for several
    # groups of the step (k_gemm_nt: both forward GEMMs and both data-gradient GEMMs; k_gemm_tn: the dW GEMMs) counts
    # with all of them: algorithmic work per launch = the groups' work / their launches, duration = their average.
    by_kernel = {}
    for k, v in prof.items():
        if group_algorithmic(cfg, k, 1, 1)[0]:
            by_kernel.setdefault(kernel_of_group(k), []).append(k)
    dom_kernel = max(by_kernel, key=lambda kn: sum(prof[g][1] / max(prof[g][0], 1) for g in by_kernel[kn]))
    dom_groups = sorted(by_kernel[dom_kernel])
    gm.set_profile(True, only=",".join(dom_groups))
    for i in range(args.steps):
        gm.train_async(batches[i % nb])
    gm.sync()
    rep = gm.profile_report()
    gm.set_profile(False)
    # ---- the same step once the GPU's clocks have ramped up (reported BESIDE the headline, never as `value`): a short
    # region runs on a ramping clock -- tools/ramp_probe.py: 143 us per step in steps 0-19 after an idle queue, 139 in 40-59, 133
    # from step ~120 on, the first forward GEMM 20.4 -> 18.5 us -- so K = 20 prices the ramp, K >= 1000 the step
    steady = None
    if args.steps < 1000:
        for i in range(300):
            gm.train_async(batches[i % nb])
        gm.sync()
        t0s = time.perf_counter()
        for i in range(300):
            gm.train_async(batches[(300 + i) % nb])
        gm.sync()
        dts = (time.perf_counter() - t0s) / 300
        gm.set_profile(True, only=",".join(dom_groups))
        for i in range(300):
            gm.train_async(batches[i % nb])
        gm.sync()
        rs = gm.profile_report()
        gm.set_profile(False)
        cnt_s = sum(rs[g][0] * GROUP_LAUNCHES.get(g, 1) for g in dom_groups)
        avg_s_s = sum(rs[g][1] for g in dom_groups) / cnt_s / 1e3
        work_s = sum(group_algorithmic(cfg, g, nnz, uniq)[1] * rs[g][0] for g in dom_groups) / cnt_s
        peak_s = F32_MFMA_PEAK_TFS * 1e12 if group_algorithmic(cfg, dom_groups[0], nnz, uniq)[0] == "mfma" else HBM_PEAK_GBS * 1e9
        steady = {"after_untimed_steps": 300 + 2 * args.steps + args.warmup + 3 + PROFILE_STEPS + args.priming + 25, "steps": 300, "ms_per_step": 1e3 * dts,
                  "roofline_avg_launch_us": avg_s_s * 1e6, "roofline_frac": work_s / avg_s_s / peak_s,
                  "note": "the same step %d steps later, 300 steps timed instead of %d (a longer sample of the headline's steady state; not part of `value`)" % (300 + args.steps, args.steps)}
    loss = gm.train(batches[0])
    if not loss > 0.01:
        raise RuntimeError("loss %.4g is under the reference's stop threshold (no backward below 0.01): the timed steps are not "
                           "full training steps -- use more or fresh batches" % loss)
    cnt = sum(rep[g][0] * GROUP_LAUNCHES.get(g, 1) for g in dom_groups)            # launches of the kernel in this pass
    ms = sum(rep[g][1] for g in dom_groups)
    kind = group_algorithmic(cfg, dom_groups[0], nnz, uniq)[0]
    work = sum(group_algorithmic(cfg, g, nnz, uniq)[1] * rep[g][0] for g in dom_groups) / cnt     # per launch
    avg_s = ms / cnt / 1e3
    if kind == "mfma":
        roof = {"kernel": dom_kernel, "bound": "mfma", "achieved": work / avg_s / 1e12, "peak": F32_MFMA_PEAK_TFS,
                "unit": "TFLOP/s", "traffic": None}
    else:
        roof = {"kernel": dom_kernel, "bound": "hbm", "achieved": work / avg_s / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "traffic": None}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["groups"] = dom_groups
    roof["algorithmic_per_launch"] = work
    # HBM bytes per launch of that kernel from the committed PMC passes (tools/profile_round.sh +
    # tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs, gfx950 corrections applied)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        roof["traffic"] = sum(pmc["groups"][g]["hbm_bytes_per_launch"] * rep[g][0] for g in dom_groups) / sum(rep[g][0] for g in dom_groups)
        roof["traffic_unit"] = "bytes/launch"
        roof["traffic_source"] = "profiles/pmc_traffic.json (" + pmc["source"] + ")"
    except (OSError, KeyError, ValueError):
        pass
    roof["avg_launch_us"] = avg_s * 1e6
    roof["launches_per_step"] = len(dom_groups)
    groups = {k: {"launches": v[0], "avg_us": 1e3 * v[1] / max(v[0], 1)} for k, v in prof.items()}
    # every group against its own roofline (HIP events around the group, all groups on one stream: includes ~4 us of
    # launch latency per group, so these are lower bounds of the kernels' own fractions)
    roof_groups = {}
    for k, g in groups.items():
        kind_k, work_k = group_algorithmic(cfg, k, nnz, uniq)
        if not kind_k:
            continue
        t = g["avg_us"] * 1e-6
        frac_k = work_k / t / 1e12 / F32_MFMA_PEAK_TFS if kind_k == "mfma" else work_k / t / 1e9 / HBM_PEAK_GBS
        roof_groups[k] = {"bound": kind_k, "avg_us": round(g["avg_us"], 2), "frac": round(frac_k, 4)}
    fc_flops, _ = fc_flops_per_step(cfg)
    out = {
        "metric": "Wide&Deep training examples/sec", "value": cfg["B"] * args.steps / dt, "unit": "examples/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Wide&Deep synthetic, 26 sparse fields x 100k vocab x 16-dim emb, "
                               "13 dense, FC[512,256,1], batch 4096, Zipf(1.05) ids, Adam + Ftrl(wide), 1 MI355X",
                   "global_batch": cfg["B"], "parallelism": "single", "resident_inputs": True, "hip_graph": bool(args.graph),
                   # "ids ~ Zipf(1.05) over V": the truncated law by inverse CDF (ps_amd/synth.py)
                   "id_generator": ("uniform" if cfg["zipf"] <= 1.0 else cfg["idgen"]) + ("(alpha=%g, V=%d)" % (cfg["zipf"], cfg["V"])),
                   "lookups_per_batch": nnz, "unique_keys_per_batch": uniq, "hottest_run": hottest,
                   "priming_steps_untimed": args.priming,
                   "untimed_steps_before_the_timed_region": {"first_launches": 3, "kernel_group_profile_pass": PROFILE_STEPS, "from_idle_probe": 25 if args.priming > 0 else 0,
                                                             "priming": args.priming, "warmup": max(args.warmup, 1)}},
        "roofline": roof,
        "roofline_groups": roof_groups,
        # all FC flops of the step / step time / f32 MFMA peak: the matrix cores' utilisation over the WHOLE step
        "mfma_utilisation": fc_flops / (dt / args.steps) / 1e12 / F32_MFMA_PEAK_TFS,
        "kernel_groups_us": {k: round(v["avg_us"], 2) for k, v in groups.items()},
        # serialised kernel-group time / step time: > 1 means the three streams overlap that much work
        "kernel_sum_over_step": sum(v["avg_us"] for v in groups.values()) / (1e6 * dt / args.steps),
        "launches_per_step": sum(GROUP_LAUNCHES.get(k, 1) for k in groups),
        "final_loss": loss,
        "host": host_info(),
    }
    if steady:
        out["after_clock_ramp"] = steady
    if from_idle:
        out["from_idle_queue"] = from_idle
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(cfg)
        nthr = min(os.cpu_count() or 1, 64)
        if nthr > 1:
            out["cpu_baseline_threads"] = cpu_baseline_threads(cfg, nthr)
    if args.clamped and cfg["zipf"] > 1.0 and cfg["idgen"] != "zipf_clamped":
        # rounds 1-2's id generator (unbounded Zipf clamped to V - 1) on the same model, so the change of workload is visible
        for b in batches:
            b.close()
        batches = []
        cfg_c = dict(cfg, idgen="zipf_clamped")
        rng_c = np.random.default_rng(cfg["seed"])
        st_c = []
        for _ in range(64):
            E, X, Y, W = synth_batch(cfg_c, rng_c)
            batches.append(ps_amd.DeviceBatch(kv, E, X, Y, W))
            if len(st_c) < 16:
                st_c.append(synth.id_stats(E))
        nsteps = min(args.steps, 1000)
        for i in range(64):
            gm.train_async(batches[i])
        gm.sync()
        t0 = time.perf_counter()
        for i in range(nsteps):
            gm.train_async(batches[i % 64])
        gm.sync()
        dtc = time.perf_counter() - t0
        out["zipf_clamped"] = {"id_generator": "zipf_clamped(alpha=%g, V=%d): min(Zipf - 1, V - 1), rounds 1-2" % (cfg["zipf"], cfg["V"]),
                               "steps": nsteps, "ms_per_step": 1e3 * dtc / nsteps, "examples_per_s": cfg["B"] * nsteps / dtc,
                               "unique_keys_per_batch": int(round(np.mean([s_["unique_keys"] for s_ in st_c]))),
                               "hottest_run": int(max(s_["hottest_run"] for s_ in st_c))}
    if args.gather:
        out["gather_hbm"] = gather_roofline(kv, args)
    # the headline model and its store are CLOSED before the other legs: with two live models on the device every stream join of
    # the multi-hot leg would take its event form (ps_store.hip dev_waits_ok) and the leg would under-measure its own step
    # (0.411 ms in the driver's line against 0.390-0.395 stand-alone)
    for b in batches:
        b.close()
    batches = []
    gm.close(); kv.close()
    # (the sharded leg -- a child process with its own RCCL communicators -- first: once, behind the two 246 GB tables of the fused-Adam leg, it
    #  did not finish within its limit; nothing of this process is on the device while it runs either way)
    if args.sharded_leg:
        out["sharded_n1"] = sharded_n1_leg(args)
    if args.multi_hot:
        out["multi_hot"] = multi_hot_step(cfg)
    if args.fused_adam:
        out["fused_adam_hbm"] = fused_adam_roofline(args)
    if args.ingest_fed:
        out["ingest_fed"] = ingest_fed_leg(cfg, out["value"])
    if args.sharded_leg:
        try:
            out["sharded_n1"]["projected_scaling_n8"] = project_n8(out["ms_per_step"], out["sharded_n1"]["ms_per_step"]["rccl_with_own_keys_in_place"], cfg)
            # the same arithmetic on the mapped-peer step (rows and gradients as stores into the peers' mapped memory, no RCCL launch on the chain)
            out["sharded_n1"]["projected_scaling_n8_mapped_peer"] = project_n8(out["ms_per_step"], out["sharded_n1"]["ms_per_step"]["mapped_peer"], cfg, mapped=True)
        except (KeyError, TypeError):
            pass
    return out


def project_n8(fused_ms, rccl_n1_ms, cfg=None, mapped=False):
    """A PROJECTION, not a measurement (no multi-GPU node has run this code; DESIGN.md 6.2 has the reasoning): the 8-rank step =
    the N = 1 step with every collective through RCCL (launch + handshake of the four collectives included, the self parts moved at
    device bandwidth) + what real links add on the critical chain.  scaling = 8 x fused step / that."""
    link_gbs = 153.0                                   # one xGMI link (MI355X_MICROARCH.md); an all-to-all-v uses the 7 links in parallel
    rows_mb = grads_mb = 0.37                          # per peer and step at N = 8 (profiles/r04_rehearse_n8.log: 2.60 MB over 7 peers)
    a2a_us = 1e3 * (rows_mb + grads_mb) / link_gbs     # both sit on the critical chain
    cfg = cfg or C2
    dims = [cfg["F"] * cfg["D"] + cfg["X"]] + list(cfg["fc"])
    dense = sum((dims[i] + 1) * dims[i + 1] for i in range(len(dims) - 1))
    # [fc | wide bias g | 8 worker slots of (gbar_w, touched_w packed 24 bits to a float)]: 1.54 MB (rounds 2-4: [fc | wide G | wide C | bias] = 2.21 MB)
    flat_mb = 4e-6 * (dense + 1 + 8 * (1 + (cfg["wide"] + 23) // 24))
    ring_us = 1e3 * 2.0 * (7.0 / 8.0) * flat_mb / link_gbs          # ring all-reduce, bandwidth term only
    slack_us = 18.0                                    # flat gradient ready -> the next step's first GEMM needs the replicated update (stamps), minus the update's 8 us
    exposed_us = max(0.0, ring_us - slack_us)
    step8 = rccl_n1_ms + 1e-3 * (a2a_us + exposed_us)
    return {"is_a_projection": True, "n1_step_through_rccl_ms": round(rccl_n1_ms, 5), "plus_link_time_rows_and_gradients_us": round(a2a_us, 2),
            "all_reduce_mb": round(flat_mb, 3), "all_reduce_ring_us": round(ring_us, 1), "of_which_exposed_us": round(exposed_us, 1), "projected_step_ms": round(step8, 5),
            "fused_n1_ms": round(fused_ms, 5), "scaling_1_to_8": round(8.0 * fused_ms / step8, 2),
            "exchange": "rows and gradients as stores into the peers' mapped memory (one launch each + flags); id blocks and all-reduce through RCCL" if mapped else "every collective through RCCL",
            "note": ("bandwidth terms only for the links: the flag round trip between devices (one xGMI write + a polling load) comes on top" if mapped else
                     "bandwidth terms only for the links: RCCL's small-message latency between devices is unknown here and comes on top")}


def multi_hot_step(cfg, steps=60):
    """BASELINE configs[4]'s shape on this GPU (reported beside the headline, not part of `value`): the same model with
    Poisson(30) ids per (sample, field) -- ~3.2 M ids per step -- Zipf(1.05), sum pooling, FTRL on the embedding rows."""
    import ps_amd
    from ps_amd import synth
    rng = np.random.default_rng(cfg["seed"] + 5)
    B, F, V = cfg["B"], cfg["F"], cfg["V"]
    kv = ps_amd.KVStore(0, cfg["seed"])
    kv.create_embedding([V] * F, cfg["D"])
    kv.set_updater("emF", ps_amd.FtrlUpdater())
    bs, nnz_max, nnz_sum, uniq = [], 0, 0, []
    for _ in range(16):
        lens = np.clip(rng.poisson(30, size=B * F), 1, 100)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nnz = int(offsets[-1]); nnz_max = max(nnz_max, nnz); nnz_sum += nnz
        ids = synth.draw_ids(rng, 1.05, V, nnz, cfg.get("idgen", "zipf_truncated"))
        if len(uniq) < 3:       # unique (field, id) rows of a batch: what the fused backward + Ftrl reads and writes once each
            field = np.repeat(np.tile(np.arange(F, dtype=np.int64), B), lens)
            uniq.append(int(np.unique(field * V + ids).size))
        W = rng.integers(0, cfg["wide"], size=(B, F)).astype(np.int64)
        bs.append(ps_amd.DeviceBatch(kv, ids, rng.standard_normal((B, cfg["X"])).astype(np.float32),
                                     (rng.random(B) < 0.25).astype(np.float32), W, offsets))
    gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], cfg["X"], cfg["fc"], cfg["wide"], store=kv, max_batch=B, max_nnz=nnz_max)
    for i in range(16):
        gm.train_async(bs[i])
    gm.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        gm.train_async(bs[i % 16])
    gm.sync()
    dt = (time.perf_counter() - t0) / steps
    loss = gm.train(bs[0])
    import ctypes as C_
    from ps_amd import native as N_
    why = C_.create_string_buffer(256)
    joins = "device flags" if N_.lib().ps_store_join_mode(kv.h, why, 256) == 1 else "events (%s)" % why.value.decode()
    for b in bs:
        b.close()
    gm.close(); kv.close()
    # its own roofline: the step's algorithmic HBM bytes -- gather nnz (4 D + 8) + 8 (B F + 1), fused backward + Ftrl
    # nnz 4 D + U 6 * 4 D + nnz 8 -- over the WHOLE step's time against 8 TB/s (the FC chain and the sort run inside that time too)
    D, nnz_avg, U = cfg["D"], nnz_sum / 16, float(np.mean(uniq))
    gather_b = nnz_avg * (4 * D + 8) + 8 * (B * F + 1)
    bwd_b = nnz_avg * 4 * D + U * 6 * 4 * D + nnz_avg * 8
    return {"workload": "configs[4] shape on 1 GPU: bags of Poisson(30) ids per (sample, field), Zipf(1.05), FTRL rows, batch 4096",
            "id_generator": cfg.get("idgen", "zipf_truncated"),
            "ids_per_step": nnz_sum // 16, "unique_rows_per_step": int(U), "ms_per_step": 1e3 * dt, "examples_per_s": B / dt, "ids_per_s": nnz_sum / 16 / dt,
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": {"gather": gather_b, "backward_plus_ftrl": bwd_b},
                         "achieved": (gather_b + bwd_b) / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (gather_b + bwd_b) / dt / 1e9 / HBM_PEAK_GBS,
                         "note": "whole step (gather, sort, FC chain, per-key reduce + Ftrl) over the algorithmic bytes of its two HBM-bound kernels"},
            "stream_joins": joins, "final_loss": loss}


def leg_sharded_n1(args):
    """configs[2]'s step (ps_shard_step: plan, id-block exchange, owner gather, rows, train, gradients, owner push, flat
    reduction) on ONE GPU with a 1-rank table, timed twice: collectives as device copies, and with every collective through
    RCCL (three 1-rank communicators on their three streams, self send/recv, all-reduce) -- the wire this box can run."""
    from ps_amd import sharded
    cfg = dict(C2)
    cfg["zipf"] = args.zipf
    cfg["idgen"] = args.idgen
    steps = min(args.steps, 1000)
    ct, regs = {}, {}
    res, info = sharded.sharded_n1_modes(cfg, synth_batch, 0, steps, modes=(0, 2, 3), with_info=True, coll_times=ct, regions=regs)
    return {"workload": "configs[2]'s sharded step on 1 GPU (1-rank table), batch %d; %d steps after 300 priming steps (268 + a wait + 32)" % (cfg["B"], steps),
            "ms_per_step": {k: round(v, 5) for k, v in res.items()},
            # (under 200 steps: the median of five regions of that many steps, all five here)
            "regions_ms": regs,
            "wire_cost_ms_per_step": round(res["rccl_with_own_keys_in_place"] - res["device_copies"], 5),
            "examples_per_s": {k: cfg["B"] / (1e-3 * v) for k, v in res.items()},
            # device time of every collective of the step by kind (HIP events around each call, a pass of its own)
            "collective_device_us": ct, "rccl": info}


def sharded_n1_leg(args):
    """leg_sharded_n1 in a child process under a time limit: loading librccl and creating communicators is the one thing in
    this file that talks to something outside the process (bootstrap sockets on loopback); it must not be able to take the
    headline line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", "sharded_n1", "--steps", str(min(args.steps, 1000)),
           "--zipf", str(args.zipf), "--idgen", args.idgen]
    env = {k: v for k, v in os.environ.items() if k != "PS_BENCH_STDOUT_FD"}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr.strip()[-400:])}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired as e:
        tail = (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))[-600:]
        return {"error": "the sharded_n1 leg did not finish within 150 s", "stderr_tail": tail}


def gather_roofline(kv, args):
    """BASELINE configs[3] shape: ONE 1e9-row x 64-dim f32 table (256 GB of the 288 GB HBM), uniformly random
    ids: 2^22 single-hot lookups / launch, and 2^17 bags of 32 ids (configs[4]'s multi-hot shape)."""
    import ctypes as C
    from ps_amd import native as N
    res = []
    for rows, D, n, bag in ((args.gather_rows, 64, 1 << 22, 1), (args.gather_rows, 64, 1 << 17, 32)):
        ms, br, bw = C.c_double(), C.c_double(), C.c_double()
        rc = N.lib().ps_bench_gather(kv.h, rows, D, n, bag, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw))
        if rc != 0 and rows > 64 * 1000 * 1000:          # the 256 GB table did not fit beside what else is resident
            rows = 64 * 1000 * 1000
            rc = N.lib().ps_bench_gather(kv.h, rows, D, n, bag, 20, 0x5EED, C.byref(ms), C.byref(br), C.byref(bw))
        N.check(rc)
        res.append({"rows": rows, "D": D, "lookups": n * bag, "bag": bag, "table_GB": rows * D * 4 / 1e9,
                    "avg_launch_us": ms.value * 1e3, "read_GBs": br.value / ms.value / 1e6,
                    "read_plus_write_GBs": (br.value + bw.value) / ms.value / 1e6,
                    "frac_of_8TBs_read": br.value / ms.value / 1e6 / HBM_PEAK_GBS,
                    "frac_of_8TBs_read_plus_write": (br.value + bw.value) / ms.value / 1e6 / HBM_PEAK_GBS,
                    # what is physically left: against the guide's MEASURED copy ceiling (MI355X_MICROARCH.md: 6.29 TB/s read + write) --
                    # the single-hot gather writes as many bytes as it reads, so its read-only fraction cannot pass one half of that
                    "frac_of_6.29TBs_read_plus_write": (br.value + bw.value) / ms.value / 1e6 / HBM_COPY_CEILING_GBS})
    return res


def ingest_fed_leg(cfg, resident_examples_per_s, epochs=5):
    """configs[1] trained STRAIGHT FROM libsvm TEXT (data/DataSet.java:77-100's reader threads, data/LibsvmParser.java:13-25,
    CTR.java:47-68) through ps_ingest_*: CTR-shaped lines (label + 26 `idx:1` + 13 `idx:val`) held in memory, parsed by
    min(nproc, 32) host threads into a ring of pinned batches, groups of batches per H2D copy, the same fused step on the batches as they
    arrive.  Reported beside the resident number: the headline's inputs are in HBM when its timed region starts; this is the rate
    when they are not."""
    import ps_amd
    F, X, B, V = cfg["F"], cfg["X"], cfg["B"], cfg["V"]
    rng = np.random.default_rng(cfg["seed"] + 77)
    nlines, nbatch = 4 * B, 512       # (an epoch of 96 batches measured the epoch boundary: threads restarted, every slot waiting for the previous epoch's kernels)
    E, Xd, Y, _ = synth_batch(cfg, rng, B=nlines)
    lines = np.array([(str(int(Y[i])) + " " + " ".join("%d:1" % v for v in E[i]) + " " + " ".join("%d:%.6f" % (F + 1 + j, Xd[i, j]) for j in range(X))).encode()
                      for i in range(nlines)], dtype=object)
    # 512 batches of lines drawn from those 16 384 (the text, not the arrays, is what the leg starts from)
    text = b"\n".join(lines[rng.integers(0, nlines, size=nbatch * B)]) + b"\n"
    threads = int(os.environ.get("PS_INGEST_THREADS", max(1, min(os.cpu_count() or 1, 32))))      # (32 feed the step with room to spare; 96 measured no better)
    kv = ps_amd.KVStore(0, cfg["seed"])
    kv.create_embedding([V] * F, cfg["D"])
    gm = ps_amd.WideDeepNN.buildModel(F, cfg["D"], X, cfg["fc"], cfg["wide"], store=kv, max_batch=B)
    ds = ps_amd.DataSet(kv, text, F, X, B, wide_size=cfg["wide"], threads=threads)
    assert ds.lines() == nbatch * B
    # the pipeline alone (parse + H2D, nothing trains): what the host side can deliver
    for _ in ds:
        pass
    ds.reset()
    t0 = time.perf_counter()
    k = 0
    for _ in ds:
        k += 1
    pipe_dt = time.perf_counter() - t0
    ds.reset()
    st0 = ds.stats()
    # training from the pipeline
    ds.train(gm)                                    # one epoch of warm-up (ps_ingest_train: CTR.java:84-100's loop, in C)
    gm.sync(); ds.reset()
    # (every epoch timed on its own, the MEDIAN reported with all of them beside it: this leg is the one the host's other activity
    #  reaches -- on one box two runs minutes apart read 0.181 and 0.153 ms per step; the mean of two epochs was whichever it met)
    steps, ep_ms = 0, []
    for _ in range(epochs):
        t0 = time.perf_counter()
        n_ep = ds.train(gm)
        gm.sync()
        ep_ms.append(1e3 * (time.perf_counter() - t0) / max(n_ep, 1))
        steps += n_ep
        ds.reset()
    dt = 1e-3 * float(np.median(ep_ms)) * steps
    st1 = ds.stats()
    loss = gm.train(ps_amd.DeviceBatch(kv, *synth_batch(cfg, rng)))
    ds.close(); gm.close(); kv.close()
    thread_s_per_line = (st1["parse_seconds"] - st0["parse_seconds"]) / max(st1["lines"] - st0["lines"], 1)
    block = B * (F * 8 * (2 if cfg["wide"] else 1) + X * 4 + 4)
    fed = B * steps / dt
    out = {"workload": "configs[1] from libsvm text in memory: %d lines (%.1f MB), %d parser threads, ring of pinned batches, groups of batches per H2D copy" % (nbatch * B, len(text) / 1e6, threads),
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "epochs_ms_per_step": [round(x, 5) for x in ep_ms], "examples_per_s": fed,
           "resident_examples_per_s": resident_examples_per_s, "fed_over_resident": fed / resident_examples_per_s,
           "pipeline_alone_lines_per_s": k * B / pipe_dt,
           "parser": {"threads": threads, "thread_us_per_line": 1e6 * thread_s_per_line, "lines_per_s_all_threads": threads / thread_s_per_line,
                      "text_MB_per_s_all_threads": threads / thread_s_per_line * len(text) / (nbatch * B) / 1e6},
           "h2d_bytes_per_step": block, "h2d_GBs_at_this_rate": block * steps / dt / 1e9, "final_loss": loss}
    if fed >= 0.9 * resident_examples_per_s:
        out["bounded_by"] = "nothing on the host: within 10 % of the resident step"
    elif out["pipeline_alone_lines_per_s"] < 1.1 * fed:
        out["bounded_by"] = "the pipeline (parse + H2D copy): it delivers no more on its own"
    else:
        out["bounded_by"] = ("the step itself, on the GPU: every H2D call the pipeline makes beside the step's 14 launches costs it time whatever the bytes "
                             "(one call per batch 0.19 ms per step, groups of up to half the ring %.3f, no copies 0.144: profiles/r06_ingest_probes.txt); "
                             "the pipeline alone delivers %.1f M lines/s, the parsers %.0f M" % (out["ms_per_step"], out["pipeline_alone_lines_per_s"] / 1e6, out["parser"]["lines_per_s_all_threads"] / 1e6))
    return out


def fused_adam_roofline(args):
    """BASELINE configs[3], "fused Adam, 1-GPU HBM-roofline run": DNN over ONE HBM-resident table of R rows x 64 (R = 320 M:
    W + Adam M, V = 246 GB of the 288 GB), 2^22 uniformly random ids per step -- single-hot (EmbeddingField's own shape) and in
    bags of 32.  What is graded is the fused backward (layer/EmbeddingField.java:86-104's per-key reduce) + AdamUpdater
    (update/AdamUpdater.java:57-70) launch group of the training step, bracketed by HIP events on its own stream:
    algorithmic bytes = nnz 4 D (delta rows) + U 6 * 4 D (W, M, V read and written once per unique key) + nnz 8."""
    import ps_amd
    R, D, X = args.adam_rows, 64, 13
    res = []
    for bag in (1, 32):
        nnz = 1 << 22
        B = nnz // bag
        rng = np.random.default_rng(7 + bag)
        kv = ps_amd.KVStore(0, 0x5EED)
        try:
            kv.create_embedding([R], D)
        except Exception:
            if R <= 64 * 1000 * 1000:
                raise
            kv.close()
            R = 64 * 1000 * 1000                       # the 246 GB table did not fit beside what else is resident
            kv = ps_amd.KVStore(0, 0x5EED)
            kv.create_embedding([R], D)
        gm = ps_amd.DNN.buildModel(1, D, X, [256, 64, 1], store=kv, max_batch=B, max_nnz=nnz)
        batches, uniq = [], []
        for _ in range(3):
            ids = rng.integers(0, R, size=nnz).astype(np.int64)
            uniq.append(int(np.unique(ids).size))
            offsets = None if bag == 1 else (np.arange(B + 1) * bag).astype(np.int64)
            E = ids.reshape(B, 1) if bag == 1 else ids
            batches.append(ps_amd.DeviceBatch(kv, E, rng.standard_normal((B, X)).astype(np.float32),
                                              (rng.random(B) < 0.25).astype(np.float32), None, offsets))
        for i in range(4):
            gm.train_async(batches[i % 3])
        gm.sync()
        gm.set_profile(True, only="emb_bwd_update")
        n = 12
        for i in range(n):
            gm.train_async(batches[i % 3])
        gm.sync()
        rep = gm.profile_report()
        gm.set_profile(False)
        t0 = time.perf_counter()
        for i in range(n):
            gm.train_async(batches[i % 3])
        gm.sync()
        step_ms = 1e3 * (time.perf_counter() - t0) / n
        loss = gm.train(batches[0])
        cnt, ms = rep["emb_bwd_update"]
        us = 1e3 * ms / max(cnt, 1)
        U = float(np.mean(uniq))
        algo = nnz * 4.0 * D + U * 6 * 4.0 * D + nnz * 8.0
        gbs = algo / us / 1e3
        res.append({"rows": R, "D": D, "table_plus_adam_state_GB": R * D * 4 * 3 / 1e9, "lookups": nnz, "bag": bag, "unique_keys": int(U),
                    "kernel": "k_emb_reduce_update" + ("" if bag == 1 else " (+ k_emb_partials' tile walk in front of it)"),
                    "launch_group_us": us, "algorithmic_bytes": algo, "achieved_GBs": gbs,
                    "frac_of_8TBs": gbs / HBM_PEAK_GBS, "frac_of_6.29TBs_copy_ceiling": gbs / HBM_COPY_CEILING_GBS,
                    "step_ms": step_ms, "final_loss": loss})
        for b in batches:
            b.close()
        gm.close(); kv.close()
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        for r in res:
            g = pmc.get("fused_adam_hbm", {}).get("bag%d" % r["bag"])
            if g:
                r["traffic"] = g["hbm_bytes_per_launch"]
                r["traffic_over_algorithmic"] = g["hbm_bytes_per_launch"] / r["algorithmic_bytes"]
                r["traffic_source"] = "profiles/pmc_traffic.json (" + pmc["source"] + ")"
    except (OSError, KeyError, ValueError):
        pass
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--zipf", type=float, default=1.05, help="id distribution exponent; <= 1 means uniform")
    ap.add_argument("--idgen", default="zipf_truncated", choices=["zipf_truncated", "zipf_clamped"],
                    help="zipf_truncated = Zipf(alpha) over V by inverse CDF (SURVEY 8d); zipf_clamped = rounds 1-2's min(Zipf - 1, V - 1)")
    ap.add_argument("--clamped", type=int, default=1, help="also time the rounds 1-2 generator (zipf_clamped) as an extra entry")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="run the sharded (multi-GPU) path even at N=1")
    ap.add_argument("--is-async", type=int, default=0, help="async push (-DisPsAsync=1): no averaging, arrival order")
    ap.add_argument("--overlap", type=int, default=0, help="sharded path: plan step t+1 (key lists) while step t trains")
    ap.add_argument("--native", type=int, default=1, help="sharded path: 1 = ps_shard_step (the library drives RCCL), 0 = torch.distributed wire")
    ap.add_argument("--priming", type=int, default=300, help="extra untimed steps in front of the warm-up and the timed region (disclosed as config.priming_steps_untimed; 0: none, and no from-idle probe)")
    ap.add_argument("--prefetch-thread", type=int, default=0, help="sharded path: run that prefetch in its own host thread")
    ap.add_argument("--phases", type=int, default=0, help="sharded path: also report a per-phase stopwatch (serialised)")
    ap.add_argument("--rccl-force", type=int, default=0, help="--sharded on one GPU: 1 | 2 = every collective through RCCL anyway (ps_native.h ps_comm_rccl_create)")
    ap.add_argument("--wire-cost", type=int, default=1, help="--sharded on one GPU: also time the step with device copies / RCCL / RCCL + own keys in place")
    ap.add_argument("--sharded-leg", type=int, default=1, help="N = 1: also report configs[2]'s sharded step on this GPU, with device copies and through RCCL (child process)")
    ap.add_argument("--leg", default="", help=argparse.SUPPRESS)
    ap.add_argument("--gather", type=int, default=1)
    ap.add_argument("--multi-hot", type=int, default=1, help="also report the configs[4] shape (multi-hot bags, FTRL) on this GPU")
    ap.add_argument("--ingest-fed", type=int, default=1, help="also train configs[1] straight from libsvm text through ps_ingest_* (examples/s beside the resident number)")
    ap.add_argument("--fused-adam", type=int, default=1, help="also report configs[3]'s fused backward + Adam on a 320 M-row HBM-resident table")
    ap.add_argument("--adam-rows", type=int, default=320 * 1000 * 1000)      # W + M + V = 246 GB
    ap.add_argument("--gather-rows", type=int, default=1000 * 1000 * 1000)   # BASELINE configs[3]: 1e9 rows x 64 f32 = 256 GB
    args = ap.parse_args()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("PS_TUNE") and not (args.gpus > 1 or world_env > 1 or args.sharded):     # (sharded: applied after torch is loaded)
        # measurement: ps_tune_set knobs for A/B runs, "knob=value,knob=value"
        from ps_amd import native as N_
        for kv_ in os.environ["PS_TUNE"].split(","):
            if "=" in kv_:
                N_.lib().ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))
    # stdout carries exactly ONE line, the JSON.  Libraries chat on fd 1 too (RCCL prints a five-line version banner
    # through C stdio, flushed at exit -- after the JSON): fd 1 is pointed at stderr for the run and the line is
    # written to the real stdout at the end.
    sys.stdout.flush()
    if os.environ.get("PS_BENCH_STDOUT_FD"):            # a later stage of a multi-GPU run (ps_amd/sharded.py: the process re-executed itself)
        real_stdout = int(os.environ["PS_BENCH_STDOUT_FD"])
    else:
        real_stdout = os.dup(1)
        os.set_inheritable(real_stdout, True)           # survives the re-execution of a stage change
        os.environ["PS_BENCH_STDOUT_FD"] = str(real_stdout)
    os.dup2(2, 1)

    def emit(out):
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.leg == "sharded_n1":
        emit(leg_sharded_n1(args))
        return
    if args.leg == "ingest_fed":              # (resident rate: the round's driver number, for the ratio only)
        emit({"ingest_fed": ingest_fed_leg(dict(C2, zipf=args.zipf, idgen=args.idgen), 30.4e6)})
        return
    if args.leg == "fused_adam":              # the configs[3] fused-Adam leg alone (tools/profile_round.sh profiles it this way)
        emit({"fused_adam_hbm": fused_adam_roofline(args)})
        return
    if args.gpus > 1 or world > 1 or args.sharded:
        from ps_amd import sharded
        out = sharded.run_bench(args, C2, synth_batch)
        if out is not None:
            emit(out)
        return
    emit(run_single(args))


if __name__ == "__main__":
    main()
