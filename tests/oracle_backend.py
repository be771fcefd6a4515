"""CPU stand-in for ps_amd.sharded.HipBackend, built on the oracle (tests only).

It gives the world_size-2 gloo tests a backend with the same device-side
contract (plan / serve_pull / forward_backward / grads / apply_push /
flat_grad / apply_flat) so that the ORCHESTRATION of ps_amd/sharded.py --
routing, split sizes, buffer order, BSP / async semantics -- is exercised
without a GPU.  Never imported by the product.
"""
import numpy as np

from oracle import oracle as orc

f32 = np.float32


def local_count(V, shard, n):
    return (V - shard + n - 1) // n if V > shard else 0


class _Plan:
    pass


class _Bound:
    """Backend attribute access with the per-step fields (batch, uniq, slot, _grads, _flat ...)
    redirected to one plan context."""
    _PLAN = ("batch", "uniq", "slot", "ufield", "uid", "_grads", "_flat")

    def __init__(self, be, plan):
        object.__setattr__(self, "_be", be)
        object.__setattr__(self, "_pl", plan)

    def __getattr__(self, k):
        return getattr(self._pl if k in _Bound._PLAN else self._be, k)

    def __setattr__(self, k, v):
        setattr(self._pl if k in _Bound._PLAN else self._be, k, v)


class OracleBackend:
    nctx = 3

    def __init__(self, torch, rank, world, cfg, seed):
        self.plans = [_Plan() for _ in range(self.nctx)]
        self.t, self.rank, self.world, self.cfg, self.seed = torch, rank, world, cfg, seed
        F, D = cfg["F"], cfg["D"]
        self.F, self.D = F, D
        self.V = [cfg["V"]] * F
        # local row bases of every shard (same packing as ps_store_create_embedding)
        self.lrb = np.zeros((world, F + 1), np.int64)
        for o in range(world):
            for f in range(F):
                self.lrb[o, f + 1] = self.lrb[o, f] + local_count(self.V[f], o, world)
        xav = orc.xavier_scale(1, D)
        rows = []
        for f in range(F):
            ids = np.arange(rank, self.V[f], world)
            rows.append(orc.init_rows(seed, f, ids, D, xav))
        self.W = np.concatenate(rows).astype(f32)
        self.M = np.zeros_like(self.W); self.Vv = np.zeros_like(self.W)
        # replicated tensors
        dims = [F * D + cfg["X"]] + list(cfg["fc"])
        self.dims = dims
        self.fcW = [orc.init_dense(seed, orc.TABLE_FC(l), dims[l] * dims[l + 1], orc.xavier_scale(dims[l], dims[l + 1])) for l in range(len(cfg["fc"]))]
        self.fcb = [orc.init_dense(seed, orc.TABLE_FC(l) + 1, dims[l + 1], orc.xavier_scale(dims[l], 1)) for l in range(len(cfg["fc"]))]
        self.fcS = [[np.zeros_like(w), np.zeros_like(w), np.zeros_like(b), np.zeros_like(b)] for w, b in zip(self.fcW, self.fcb)]
        ws = cfg["wide"]
        self.ww = np.zeros(ws, f32); self.wz = np.zeros(ws, f32); self.wn = np.zeros(ws, f32)
        self.wb = np.zeros(1, f32); self.wbz = np.zeros(1, f32); self.wbn = np.zeros(1, f32)
        self.store = orc.Store(seed)
        self.model = orc.Model(self.store, orc.WIDEDEEP, F, D, cfg["X"], cfg["fc"], wide_size=ws)
        self.model.set_grad_mode(orc.GRAD_COMPAT, orc.GRAD_COMPAT, 0)     # single-hot batches: the reference order

    # ---- worker: PSRouterClient.getList fan-out
    def plan_launch(self, batch, world, ctx=0, stream=None):
        self.plans[ctx].pending = (batch, world)

    def plan_finish(self, ctx=0):
        batch, world = self.plans[ctx].pending
        return self.plan(batch, world, ctx)

    def plan(self, batch, world, ctx=0, stream=None):
        self = _Bound(self, self.plans[ctx])
        E = batch["E"]
        B, F = E.shape
        owner = E % world
        local = self.lrb[owner, np.arange(F)[None, :]] + E // world
        comp = owner * (1 << 40) + local
        uniq, inv = np.unique(comp.ravel(), return_inverse=True)      # sorted: owner-major, ascending local row
        self.batch, self.uniq, self.slot = batch, uniq, inv.reshape(B, F)
        # (field, id) of every unique key, for the string keys of the restated model
        first = np.zeros(len(uniq), np.int64)
        first[inv[::-1]] = np.arange(B * F)[::-1]
        self.ufield, self.uid = first % F, E.ravel()[first]
        counts = [int(((uniq >> 40) == o).sum()) for o in range(world)]
        return counts, self.t.tensor((uniq & ((1 << 40) - 1)).astype(np.int32))

    # ---- owner: PServer.getList
    def serve_pull(self, recv_rows, n):
        return self.t.from_numpy(self.W[recv_rows.numpy().astype(np.int64)].copy())

    # ---- worker: Model.train on the pulled rows
    def forward_backward(self, ctx, cache, want_loss=True):
        self = _Bound(self, self.plans[ctx])
        cache = cache.numpy()
        for u in range(len(self.uniq)):
            self.store.put(orc.emb_key(int(self.ufield[u]), float(self.uid[u])), cache[u], self.D, 1)
        for l in range(len(self.fcW)):
            self.store.put("fc%d.weights" % l, self.fcW[l], self.dims[l + 1], self.dims[l])
            self.store.put("fc%d.bias" % l, self.fcb[l], self.dims[l + 1], 1)
        b = self.batch
        for k in np.unique(b["W"]):
            self.store.put(orc.wide_key(float(k)), self.ww[k:k + 1], 1, 1)
        self.store.put("wide.bias", self.wb, 1, 1)
        loss = self.model.train(b["E"].astype(f32), b["X"], b["Y"], b["W"].astype(f32), do_update=False)
        self._grads = np.stack([self.model.grad(orc.emb_key(int(self.ufield[u]), float(self.uid[u]))) for u in range(len(self.uniq))]).astype(f32)
        dense = []
        for l in range(len(self.fcW)):
            dense += [self.model.grad("fc%d.weights" % l), self.model.grad("fc%d.bias" % l)]
        ws = self.cfg["wide"]
        G = np.zeros(ws, f32); Cc = np.zeros(ws, f32)
        gbar = self.model.grad("wide.bias")[0]
        for k in self.model.grad_keys():
            if k.startswith("wide.weights."):
                i = int(float(k[len("wide.weights."):]))
                G[i] = gbar; Cc[i] = 1
        self._flat = self.t.from_numpy(np.concatenate(dense + [G, Cc, np.array([gbar], f32)]).astype(f32))
        return loss

    def grads(self, ctx):
        return self.t.from_numpy(self.plans[ctx]._grads)

    # ---- owner: PServer.push + psUpdate
    def apply_push(self, recv_rows, recv_grads, n, peer_counts, is_async):
        rows = recv_rows.numpy().astype(np.int64); g = recv_grads.numpy()
        order = np.argsort(rows, kind="stable")                   # arrival (= worker) order within a key
        i = 0
        while i < n:
            j = i
            while j < n and rows[order[j]] == rows[order[i]]:
                j += 1
            r = rows[order[i]]
            if is_async:
                for k in range(i, j):
                    self.W[r], self.M[r], self.Vv[r] = orc.adam_update(self.W[r], g[order[k]], self.M[r], self.Vv[r])
            else:
                S = g[order[i]].copy()
                for k in range(i + 1, j):
                    S = (g[order[k]] + S).astype(f32)
                S = (S / f32(j - i)).astype(f32)
                self.W[r], self.M[r], self.Vv[r] = orc.adam_update(self.W[r], S, self.M[r], self.Vv[r])
            i = j

    def flat_grad(self, ctx):
        return self.plans[ctx]._flat

    def apply_flat(self, ctx, world):
        flat = self.plans[ctx]._flat.numpy()
        off = 0
        for l in range(len(self.fcW)):
            nw, nb = self.fcW[l].size, self.fcb[l].size
            gw = (flat[off:off + nw] / f32(world)).astype(f32); off += nw
            gb = (flat[off:off + nb] / f32(world)).astype(f32); off += nb
            S = self.fcS[l]
            self.fcW[l], S[0], S[1] = orc.adam_update(self.fcW[l], gw, S[0], S[1])
            self.fcb[l], S[2], S[3] = orc.adam_update(self.fcb[l], gb, S[2], S[3])
        ws = self.cfg["wide"]
        G, Cc, bg = flat[off:off + ws], flat[off + ws:off + 2 * ws], flat[off + 2 * ws]
        for k in np.nonzero(Cc > 0)[0]:
            g = f32(G[k] / Cc[k])
            w, z, n, _ = orc.ftrl_update(self.ww[k:k + 1], [g], self.wz[k:k + 1], self.wn[k:k + 1])
            self.ww[k], self.wz[k], self.wn[k] = w[0], z[0], n[0]
        w, z, n, _ = orc.ftrl_update(self.wb, [f32(bg / f32(world))], self.wbz, self.wbn)
        self.wb, self.wbz, self.wbn = w, z, n
