"""net/PServer.java's control flow in ps_amd/ps_server.py, without a GPU: the servicer over a stand-in store (a dict)
and real gRPC on the loopback interface.  What the stand-in cannot check (the arithmetic of psUpdate) is checked on the
device in tests/test_gpu_ps_server.py; what is checked here is the protocol: who waits for whom at the barrier, what is
queued and when it is handed to the store, the answers for null weights / unknown updaters, upsert's replace flag."""
import threading

import numpy as np
import pytest

S = pytest.importorskip("ps_amd.ps_server")
f32 = np.float32


class DictStore:
    """The five calls the servicer makes, over a dict; push_update records what psUpdate was given."""

    def __init__(self):
        self.kv = {"emF0.1.0": np.arange(4, dtype=f32), "fc0.weights": np.zeros(6, f32), "fc0.bias": np.zeros(2, f32)}
        self.rounds, self.step, self.updaters = [], 0, {}

    def get(self, key):
        v = self.kv.get(key)
        return None if v is None else v.copy()

    def put(self, key, val):
        self.kv[key] = np.asarray(val, f32).copy()

    def set_updater(self, group, upd):
        self.updaters[group] = upd

    def key_length(self, key):
        if key not in self.kv:
            raise S.N.PsError(S.N.PS_MISSING, "unknown key %s" % key)
        return self.kv[key].size

    def push_update(self, messages, is_async=False):
        self.rounds.append(([k for k, _ in messages], is_async))

    def advance_global_step(self, by=1):
        self.step += by


@pytest.fixture()
def served(monkeypatch):
    monkeypatch.setattr(S.Updater, "fromName", staticmethod(lambda name: (_ for _ in ()).throw(S.N.PsError(S.N.PS_NO_UPDATER, "no updater"))
                                                             if name.startswith("bogus") else name))
    made = []

    def start(worker_num, is_async=False):
        st = DictStore()
        server, port, sv = S.serve(st, 0, worker_num, is_async)
        made.append(server)
        return st, "127.0.0.1:%d" % port

    yield start
    for s in made:
        s.stop(0)


def test_get_and_upsert_answers(served):
    st, target = served(1)
    c = S.PsClient(target)
    v, ec = c.get("emF0.1.0")
    assert ec == 200 and v.tolist() == [0, 1, 2, 3]
    assert c.get("emF0.9.0") == (None, 204)
    got = c.getList(["emF0.1.0", "absent"])
    assert got["absent"] is None and got["emF0.1.0"].tolist() == [0, 1, 2, 3]
    res, ec = c.upsertList({"emF0.1.0": np.ones(4, f32), "new.key": np.full(3, 7, f32)})
    assert ec == 200
    assert res["emF0.1.0"][1] is True and res["emF0.1.0"][0].tolist() == [0, 1, 2, 3]      # kept: exists and no replace
    assert res["new.key"][1] is False and st.kv["new.key"].tolist() == [7, 7, 7]           # inserted
    res, _ = c.upsertList({"emF0.1.0": np.ones(4, f32)}, replace=True)
    assert res["emF0.1.0"][1] is False and st.kv["emF0.1.0"].tolist() == [1, 1, 1, 1]
    c.close()


def test_bsp_barrier_releases_everybody_after_one_update(served):
    st, target = served(3)
    cs = [S.PsClient(target, "w%d" % i) for i in range(3)]
    assert cs[0].push("emF0.1.0", np.ones(4, f32), "adam@x") == 0
    assert cs[1].push("fc0.weights", np.ones(6, f32), "adam@x") == 0
    assert cs[2].push("emF0.1.0", np.ones(4, f32), "adam@x") == 0
    assert cs[0].push("k", np.ones(1, f32), "bogus@x") == 500 and st.rounds == []          # unknown updater; nothing applied yet
    out = []
    ts = [threading.Thread(target=lambda c=c: out.append(c.barrier()), daemon=True) for c in cs[:2]]
    for t in ts:
        t.start()
    for t in ts:
        t.join(0.4)
    assert all(t.is_alive() for t in ts) and st.rounds == [] and st.step == 0               # two of three: everybody waits
    assert cs[2].barrier() == 200
    for t in ts:
        t.join(10)
    assert out == [200, 200]
    assert st.rounds == [(["emF0.1.0", "fc0.weights", "emF0.1.0"], False)] and st.step == 1  # ONE psUpdate, arrival order
    assert st.updaters.keys() == {"emF", "fc0.weights"}
    # the next round starts clean
    assert cs[1].push("fc0.bias", np.ones(2, f32), "adam@x") == 0
    ts = [threading.Thread(target=lambda c=c: out.append(c.barrier()), daemon=True) for c in cs[1:]]
    for t in ts:
        t.start()
    assert cs[0].barrier() == 200
    for t in ts:
        t.join(10)
    assert st.rounds[1] == (["fc0.bias"], False) and st.step == 2
    for c in cs:
        c.close()


def test_async_mode_never_blocks(served):
    st, target = served(4, is_async=True)
    c = S.PsClient(target)
    assert c.push("emF0.1.0", np.ones(4, f32), "adam@x", is_async=True) == 0
    assert st.rounds == [(["emF0.1.0"], True)]                                              # applied at once
    assert c.barrier() == 200 and c.barrier() == 200 and st.step == 2                       # every barrier: globalStep++
    c.close()


def test_bsp_push_is_validated_when_it_arrives(served):
    """a bad key or length is answered with ec 500 at push time; the round's other pushes and its barrier are
    unaffected (the bad message never reaches psUpdate)."""
    st, target = served(1)
    c = S.PsClient(target)
    assert c.push("fc0.weights", np.ones(6, f32), "adam@x") == 0
    assert c.push("fc0.weights", np.ones(5, f32), "adam@x") == 500          # wrong length
    assert c.push("nosuch.key", np.ones(1, f32), "adam@x") == 500           # the store does not hold it
    assert c.barrier() == 200
    assert st.rounds == [(["fc0.weights"], False)]
    c.close()
