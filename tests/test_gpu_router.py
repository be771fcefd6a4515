"""The reference's own router at N > 1: net/Mod.java:13-15 routes a key by String.hashCode(key) mod n
(PS_ROUTE_JAVA_STRING, with the floorMod fix), so a shard's ids are NOT an arithmetic progression.  N ranks as N
threads on one GPU (see test_gpu_multirank.py), every store created with java_string routing: each rank holds exactly
the ids Mod.shard assigns to it, the plan groups keys by that owner, and after STEPS steps every row, the replicated
tensors and globalStep equal the key-addressed simulation of the PS semantics (which knows nothing about routing)."""
import numpy as np
import pytest

from test_gpu_multirank import CallbackComm, Shared, ThreadComm, run_ranks
from test_sharded_gloo import CFG, SEED, STEPS, expected, make_batches

pytestmark = pytest.mark.gpu
ISOLATE_IN_SUBPROCESS = True
f32 = np.float32


def owned_ids(orc, f, rank, world, V):
    return np.array([i for i in range(V) if orc.mod_shard(orc.emb_key(f, float(i)), world, True) == rank], np.int64)


def java_rank_main(rank, world, shared, native, out, errs):
    try:
        import ps_amd
        from oracle import oracle as orc
        from ps_amd import native as N
        from ps_amd.sharded import HipBackend, NativeWorker, ShardedWorker
        F, D, V = CFG["F"], CFG["D"], CFG["V"]
        kv = ps_amd.KVStore(0, SEED)
        kv.create_embedding([V] * F, D, shard=rank, nshards=world, route_mode=N.PS_ROUTE_JAVA_STRING)
        gm = ps_amd.WideDeepNN.buildModel(F, D, CFG["X"], CFG["fc"], CFG["wide"], store=kv, max_batch=CFG["B"])
        mine = [owned_ids(orc, f, rank, world, V) for f in range(F)]
        # before any step: exactly the owner's ids are held, with the init values of THEIR id (not of their local row)
        xav = orc.xavier_scale(1, D)
        for f in (0, F - 1):
            np.testing.assert_array_equal(kv.get_rows(f, mine[f]), orc.init_rows(SEED, f, mine[f], D, xav))
            other = np.setdiff1d(np.arange(V), mine[f])
            if len(other):
                with pytest.raises(N.PsError):
                    kv.get_rows(f, other[:1])
        if native:
            comm = CallbackComm(rank, shared, kv)
            wk = NativeWorker(gm, world, rank, ops=comm.ops)
        else:
            comm = ThreadComm(rank, shared, kv)
            wk = ShardedWorker(HipBackend([gm]), comm)
        for b in make_batches(rank, STEPS):
            wk.step(ps_amd.Batch(b["E"], b["X"], b["Y"], b["W"]))
        kv.sync()
        if native and comm.err is not None:
            raise comm.err
        rows = {}
        for f in range(F):
            w = kv.get_rows(f, mine[f])
            for i, idv in enumerate(mine[f]):
                rows[(f, int(idv))] = w[i]
        out[rank] = (rows, [kv.get("fc%d.weights" % l) for l in range(3)], kv.get_wide(np.arange(CFG["wide"])), kv.global_step())
        if not native:
            comm.free()
        gm.close(); kv.close()
    except BaseException:       # noqa: BLE001
        import traceback
        errs.append((rank, traceback.format_exc()))
        shared.barrier.abort()


@pytest.mark.parametrize("world,native", [(2, False), (3, True)])
def test_java_string_router_n_ranks(orc, world, native):
    shared = Shared(world)
    out, errs = [None] * world, []
    run_ranks(java_rank_main, [(r, world, shared, native, out, errs) for r in range(world)])
    assert not errs, "\n".join("rank %d:\n%s" % e for e in errs)
    emb, fcW, fcb, ww, wb = expected(world, False)
    xav = orc.xavier_scale(1, CFG["D"])
    tol = 2e-5 * STEPS
    seen, touched = set(), 0
    for r in range(world):
        rows, W, wide, gstep = out[r]
        assert gstep == STEPS
        for (f, i), got in rows.items():
            assert (f, i) not in seen
            seen.add((f, i))
            assert orc.mod_shard(orc.emb_key(f, float(i)), world, True) == r
            if (f, i) in emb:
                assert np.abs(got - emb[(f, i)][0]).max() <= tol, "rank %d emF%d.%d" % (r, f, i)
                touched += 1
            else:
                np.testing.assert_array_equal(got, orc.init_rows(SEED, f, [i], CFG["D"], xav)[0])
        for l in range(3):
            assert np.abs(W[l] - fcW[l]).max() <= tol
            np.testing.assert_array_equal(W[l], out[0][1][l])
        assert np.abs(wide - ww).max() <= tol
    assert len(seen) == CFG["F"] * CFG["V"] and touched > 0          # the shards partition the key space
