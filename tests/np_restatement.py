"""Independent numpy/float32 restatement of the reference step, written
object-for-object after the Java (dicts of arrays, in-place ops, matrices
kept BY REFERENCE) so that the aliasing quirks fall out of the structure
instead of being re-derived algebraically.  It exists only to cross-check
oracle/ps_oracle.c bit-for-bit (tests/test_oracle_twin.py); small sizes only.

Citations: /root/reference/src/main/java/<file>:<line>.
"""
import math

import numpy as np

f32 = np.float32


def java_hashcode(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h - (1 << 32) if h >= (1 << 31) else h


def float_key(v):
    """Float.toString for the integer-valued ids the reference produces."""
    v = float(f32(v))
    assert v == int(v) and abs(v) < 1e7
    return "%d.0" % int(v)


def mmul(A, B):
    """jblas mmul restated as sequential-k f32 accumulate (order unknowable)."""
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    Cm = np.zeros((M, N), f32)
    for k in range(K):
        Cm += np.outer(A[:, k], B[k, :]).astype(f32)
    return Cm


def sigmoid_clip(x):
    # activations/Sigmoid.java:11
    return f32(np.float64(f32(0.001)) + np.float64(f32(f32(.999) - f32(0.001))) / (1.0 + math.exp(-float(x))))


class Adam:
    # update/AdamUpdater.java
    def __init__(self, alfa=0.005, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alfa, self.beta1, self.beta2, self.eps = f32(alfa), f32(beta1), f32(beta2), f32(eps)
        self.M, self.V = {}, {}

    def update(self, key, w, dw):
        if key not in self.M:
            t = np.zeros_like(dw)          # :77-84 the SAME zero matrix for M and V
            self.M[key] = t
            self.V[key] = t
        c1 = f32(f32(1) - self.beta1)
        c2 = f32(f32(1) - self.beta2)
        mo = self.M[key]; mo *= self.beta1                  # muli in place
        self.M[key] = (dw * c1) + mo                        # :61
        vo = self.V[key]; vo *= self.beta2
        self.V[key] = ((dw * dw) * c2) + vo                 # :62
        Mm = self.M[key] / c1
        Vv = self.V[key] / c2
        den = np.sqrt(Vv.astype(np.float64)).astype(f32) + self.eps
        w += (Mm / den) * f32(f32(-1) * self.alfa)          # :69 in place on the store's matrix
        return w


class Ftrl:
    # update/FtrlUpdater.java
    def __init__(self, alfa=0.005, beta=1.0, l1=0.001, l2=0.001):
        self.alfa, self.beta, self.l1, self.l2 = f32(alfa), f32(beta), f32(l1), f32(l2)
        self.Z, self.N = {}, {}

    def update(self, key, w, dw):
        if dw.ravel()[0] == 0:
            return w
        if key not in self.N:
            self.N[key] = np.zeros(w.size, f32)
        if key not in self.Z:
            self.Z[key] = np.zeros(w.size, f32)
        zi, ni = self.Z[key], self.N[key]
        wf = w.reshape(-1)
        for i in range(wf.size):
            if abs(zi[i]) <= self.l1:
                wf[i] = 0
            else:
                sign = f32(1) if zi[i] >= 0 else f32(-1)
                den = f32(f32(self.l2 + f32(self.beta + f32(math.sqrt(float(ni[i]))))) / self.alfa)
                wf[i] = f32(-f32(zi[i] - f32(sign * self.l1)) / den)
        g = dw.reshape(-1)
        g2 = (g.astype(np.float64) ** 2).astype(f32)
        s = np.sqrt((ni + g2).astype(np.float64)).astype(f32) - np.sqrt((ni / self.alfa).astype(np.float64)).astype(f32)
        zi += g - s * wf
        ni += g2
        return w


class KVStore:
    # store/KVStore.java (standalone)
    def __init__(self):
        self.store, self.sum, self.sumCnt = {}, {}, {}

    def get(self, key, init):
        if key not in self.store:
            self.store[key] = init()
        return self.store[key]

    def sum_(self, key, val):                                # :192-200
        if key not in self.sum:
            self.sum[key] = val                              # by reference
            self.sumCnt[key] = 1
        else:
            s = self.sum[key]
            s += val                                         # addi; val may be s itself
            self.sumCnt[key] += 1

    def update(self, updaters, grads_out=None):              # :240-261
        for key in list(self.sum.keys()):
            upd = updaters.get(key)
            if upd is None:
                for uk in updaters:
                    if key.startswith(uk):
                        upd = updaters[uk]
            if upd is None:
                upd = updaters["default"]
            g = self.sum[key]
            g /= f32(self.sumCnt[key])                       # divi in place
            if grads_out is not None:
                grads_out[key] = g.copy()
            upd.update(key, self.store[key], g)

    def clear(self):
        self.sum.clear(); self.sumCnt.clear()


class Model:
    """model/DNN.java:92-128 / model/WideDeepNN.java:105-161, thread = 1."""

    def __init__(self, kv, wide, F, D, X, fc_dims, init_emb, init_fc_w, init_fc_b):
        self.kv, self.wide, self.F, self.D, self.X = kv, wide, F, D, X
        self.fc = []
        inn = F * D + X
        for i, out in enumerate(fc_dims):
            last = i == len(fc_dims) - 1
            act = ("none" if wide else "sigmoid") if last else "relu"
            self.fc.append(dict(name="fc%d" % i, inn=inn, out=out, act=act))
            inn = out
        self.init_emb, self.init_fc_w, self.init_fc_b = init_emb, init_fc_w, init_fc_b
        self.updaters = {"default": Adam()}
        if wide:
            ftrl = Ftrl()
            self.updaters["wide.weights"] = ftrl
            self.updaters["wide.bias"] = ftrl
            self.wide_weights = {}                           # LRLayer.weights, never cleared
            self.kv.get("wide.bias", lambda: np.zeros((1, 1), f32))
        self.field_w = [dict() for _ in range(F)]
        self.field_g = [dict() for _ in range(F)]
        self.field_n = [dict() for _ in range(F)]

    def pull_weights(self):
        for f in range(self.F):
            self.field_w[f].clear(); self.field_g[f].clear(); self.field_n[f].clear()
        for i, L in enumerate(self.fc):
            L["W"] = self.kv.get(L["name"] + ".weights", lambda i=i, L=L: self.init_fc_w(i, L["out"], L["inn"]))
            L["b"] = self.kv.get(L["name"] + ".bias", lambda i=i, L=L: self.init_fc_b(i, L["out"]))

    def forward(self, E, Xd, Wd):
        """E [F x B], Xd [X x B], Wd [F x B] -- reference orientation."""
        F, D = self.F, self.D
        B = E.shape[1]
        self.E = E
        embs = []
        self.fieldZ = []
        for f in range(F):
            WX = np.zeros((D, B), f32)
            for i in range(B):
                key = "emF%d.%s" % (f, float_key(E[f, i]))
                if key not in self.field_w[f]:
                    self.field_w[f][key] = self.kv.get(key, lambda f=f, i=i: self.init_emb(f, int(E[f, i])))
                WX[:, i] = self.field_w[f][key].reshape(-1)
            np.maximum(WX, 0, out=WX)                         # relu in place, Z == A
            self.fieldZ.append(WX)
            embs.append(WX)
        self.embA = np.concatenate(embs, axis=0)
        self.concatA = np.concatenate([self.embA, Xd], axis=0)
        A = self.concatA
        for L in self.fc:
            Z = mmul(L["W"], A)
            Z += L["b"].reshape(-1, 1)
            if L["act"] == "relu":
                np.maximum(Z, 0, out=Z)
            elif L["act"] == "sigmoid":
                Z[...] = np.vectorize(sigmoid_clip, otypes=[f32])(Z)
            L["A"] = Z
            A = Z
        P = A
        if self.wide:
            bias = self.kv.store["wide.bias"]
            WX = np.zeros((1, B), f32)
            for i in range(B):
                s = f32(0)
                for j in range(F):
                    key = "wide.weights." + float_key(Wd[j, i])
                    wi = self.kv.get(key, lambda: np.zeros((1, 1), f32))
                    self.wide_weights[key] = wi
                    s = f32(s + wi[0, 0])
                WX[0, i] = s
            WX += bias[0, 0]
            self.wideZ = WX
            Z = (P + WX).astype(f32)
            Z[...] = np.vectorize(sigmoid_clip, otypes=[f32])(Z)
            self.addA = Z
            P = Z
        self.P = P
        return P

    def emb_backward(self, delta):
        D = self.D
        for f in range(self.F):
            off = f * D
            Z = self.fieldZ[f]
            for k in range(self.E.shape[1]):
                key = "emF%d.%s" % (f, float_key(self.E[f, k]))
                g = delta[off:off + D, k:k + 1].copy()
                g *= (Z[:, k:k + 1] > 0).astype(f32)
                if key not in self.field_g[f]:
                    self.field_g[f][key] = g
                    self.field_n[f][key] = 1
                else:
                    G = self.field_g[f][key]
                    G += g
                    self.field_n[f][key] += 1
            for key in self.field_n[f]:
                G = self.field_g[f][key]
                G /= f32(self.field_n[f][key])
                self.kv.sum_(key, G)

    def train(self, E, Xd, Wd, Y, grads_out=None):
        self.pull_weights()
        P = self.forward(E, Xd, Wd)
        B = Y.shape[1]
        s = f32(0)
        for i in range(B):
            p, l = P[0, i], Y[0, i]
            s = f32(s + f32(-float(l) * math.log(float(p)) - (float(f32(1) - l) * math.log(float(f32(1) - p)))))
        loss = f32(s / f32(B))
        delta = ((P - Y) / (P * (f32(1) - P))).astype(f32)
        if loss <= f32(0.01) or math.isnan(loss):
            return loss
        if self.wide:
            delta *= self.addA * (f32(1) - self.addA)
            gs = f32(0)
            for c in range(B):
                gs = f32(gs + delta[0, c])
            gbar = np.array([[gs / f32(B)]], f32)
            self.kv.sum_("wide.bias", gbar)
            for key in self.wide_weights:
                self.kv.sum_(key, gbar)
        for li in range(len(self.fc) - 1, -1, -1):
            L = self.fc[li]
            preA = self.concatA if li == 0 else self.fc[li - 1]["A"]
            if L["act"] == "relu":
                delta *= (L["A"] > 0).astype(f32)
            elif L["act"] == "sigmoid":
                delta *= L["A"] * (f32(1) - L["A"])
            db = np.zeros((L["out"], 1), f32)
            for c in range(B):
                db[:, 0] += delta[:, c]
            db /= f32(B)
            self.kv.sum_(L["name"] + ".bias", db)
            dW = mmul(delta, np.ascontiguousarray(preA.T))
            dW /= f32(B)
            self.kv.sum_(L["name"] + ".weights", dW)
            ndelta = np.zeros((L["inn"], B), f32)
            Wt = L["W"].T
            for k in range(L["out"]):
                ndelta += np.outer(Wt[:, k], delta[k, :]).astype(f32)
            L["delta"] = ndelta
            delta = ndelta
        self.emb_backward(delta)      # ConcatLayer.backward -> embedding.backward()
        self.emb_backward(delta)      # model loop reaches layers[0]
        self.kv.update(self.updaters, grads_out)
        self.kv.clear()
        return loss
