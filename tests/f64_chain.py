"""The training step of the reference in FLOAT64, as a chain: forward, loss, backward, per-key embedding gradient,
updaters, over as many steps as a test runs -- the yardstick the end-to-end tolerances are stated against
(VERDICT r2 next #6).

The float32 implementations under test (the HIP path; the C oracle, whose sgemm order is a guess at jblas') both
differ from the exact result by their own propagated roundoff.  With x64 the float64 chain started from the SAME
initial parameters and fed the SAME batches, the tests assert for every tensor x of a step

        |x_gpu - x64|  <=  RTOL * (|x64| + max|x64|)  +  C * max|x_oracle - x64|  (+ the roundoff floor of x's own last operation)

i.e. the HIP path may be as far from the exact chain as north_star's 1e-5 relative (elementwise, and relative to the
tensor's magnitude: the same reading the forward checks use against the oracle), plus a small multiple of how far the
reference-order float32 chain itself is -- no hand-picked absolute floors, no loosened end-to-end rtol.

Semantics are the reference's as written, vectorised: clipped sigmoid (activations/Sigmoid.java:11),
CrossEntropy (loss/CrossEntropy.java:10-28), FcLayer.backward (layer/FcLayer.java:93-110), the double
EmbeddingLayer.backward (effective gradient S (n + 1) / (2 n^2), App. A.6), AdamUpdater / FtrlUpdater as written
(update/AdamUpdater.java:57-70, update/FtrlUpdater.java:51-76; their float constants are PARAMETERS and enter with
their float32 values), LRLayer's "every key seen so far gets the batch-mean delta" (layer/LRLayer.java:100-120).
Test infrastructure only."""
import numpy as np

f32, f64 = np.float32, np.float64

ALFA = f64(f32(0.005)); B1 = f64(f32(0.9)); B2 = f64(f32(0.999)); EPS = f64(f32(1e-8))
C1 = f64(f32(1) - f32(0.9)); C2 = f64(f32(1) - f32(0.999))                # evaluated in float32 by the reference
F_ALFA = f64(f32(0.005)); F_BETA = f64(f32(1.0)); F_L1 = f64(f32(0.001)); F_L2 = f64(f32(0.001))
S_LO = f64(f32(0.001)); S_SPAN = f64(f32(f32(.999) - f32(0.001)))


def sigmoid_clip(z):
    return S_LO + S_SPAN / (1.0 + np.exp(-z))


def adam(w, g, m, v):
    m = g * C1 + m * B1
    v = (g * g) * C2 + v * B2
    w = w + ((m / C1) / (np.sqrt(v / C2) + EPS)) * (-1.0 * ALFA)
    return w, m, v


def ftrl(w, g, z, n):
    """vector form; rows (leading axis) whose g[..., 0] == 0 are skipped whole (FtrlUpdater.java:52)"""
    w, g, z, n = (np.array(a, f64, copy=True) for a in (w, g, z, n))
    live = (g.reshape(g.shape[0], -1)[:, 0] != 0) if g.ndim > 1 else np.array(g.reshape(-1)[0] != 0)
    sign = np.where(z >= 0, 1.0, -1.0)
    wn = np.where(np.abs(z) <= F_L1, 0.0, -(z - sign * F_L1) / ((F_L2 + (F_BETA + np.sqrt(n))) / F_ALFA))
    s = np.sqrt(n + g * g) - np.sqrt(n / F_ALFA)
    zn = z + (g - s * wn)
    nn = n + g * g
    if g.ndim > 1:
        k = live.reshape((-1,) + (1,) * (g.ndim - 1))
        return np.where(k, wn, w), np.where(k, zn, z), np.where(k, nn, n)
    return (wn, zn, nn) if bool(live) else (w, z, n)


class Chain:
    def __init__(self, wide, F, D, X, fc, wide_size=0, emb_updater="adam"):
        self.wide, self.F, self.D, self.X, self.fc, self.ws = wide, F, D, X, list(fc), wide_size
        self.dims = [F * D + X] + list(fc)
        self.rows = [dict() for _ in range(F)]          # id -> [w, s1, s2]
        self.W, self.b, self.S = [], [], []             # W[l]: [in][out]
        self.emb_updater = emb_updater
        if wide:
            self.ww = np.zeros(wide_size, f64); self.wz = np.zeros(wide_size, f64); self.wn = np.zeros(wide_size, f64)
            self.wb = np.zeros(1, f64); self.wbz = np.zeros(1, f64); self.wbn = np.zeros(1, f64)
            self.seen = np.zeros(wide_size, bool)       # LRLayer.weights membership: never cleared

    def load_fc(self, weights, biases):
        """weights[l]: the store's "fc<l>.weights" ([in][out] row-major, flat), biases[l]: [out]"""
        self.W = [np.asarray(w, f64).reshape(self.dims[l], self.dims[l + 1]).copy() for l, w in enumerate(weights)]
        self.b = [np.asarray(b, f64).copy() for b in biases]
        self.S = [[np.zeros_like(w), np.zeros_like(w), np.zeros_like(b), np.zeros_like(b)] for w, b in zip(self.W, self.b)]

    def step(self, E, Xd, Y, Wd, init_rows, update=True):
        """One training step.  init_rows(f, ids) -> the initial float32 rows of ids this chain has not seen yet."""
        F, D, nfc = self.F, self.D, len(self.fc)
        B = E.shape[0]
        for f in range(F):
            new = [int(i) for i in np.unique(E[:, f]) if int(i) not in self.rows[f]]
            if new:
                r = np.asarray(init_rows(f, np.array(new, np.int64)), f64)
                for k, i in enumerate(new):
                    self.rows[f][i] = [r[k].copy(), np.zeros(D, f64), np.zeros(D, f64)]
        Z0 = np.concatenate([np.stack([self.rows[f][int(i)][0] for i in E[:, f]]) for f in range(F)], axis=1)      # [B][F*D]
        A0 = np.maximum(Z0, 0)
        A = [np.concatenate([A0, np.asarray(Xd, f64)], axis=1)]
        for l in range(nfc):
            z = A[l] @ self.W[l] + self.b[l]
            if l < nfc - 1:
                z = np.maximum(z, 0)
            elif not self.wide:
                z = sigmoid_clip(z)
            A.append(z)
        out = {"A": A}
        Yc = np.asarray(Y, f64).reshape(B, 1)
        if self.wide:
            zw = self.ww[Wd].sum(axis=1, keepdims=True) + self.wb[0]
            P = sigmoid_clip(A[nfc] + zw)
        else:
            P = A[nfc]
        out["P"] = P[:, 0]
        out["loss"] = float(np.mean(-Yc * np.log(P) - (1 - Yc) * np.log(1 - P)))
        d = ((P - Yc) / (P * (1 - P))) * (P * (1 - P))              # CrossEntropy' then Sigmoid' (y (1 - y) of the CLIPPED y)
        out["gbar"] = float(d.mean())
        deltas, dW, db = [None] * nfc, [None] * nfc, [None] * nfc
        # mag_*: sum of |terms| of each quantity's own last contraction (its float32 roundoff floor is ~ 8 eps * that)
        mag_d, mag_dW, mag_db = [None] * nfc, [None] * nfc, [None] * nfc
        for l in range(nfc - 1, -1, -1):
            dW[l] = A[l].T @ d / B
            db[l] = d.mean(axis=0)
            mag_dW[l] = np.abs(A[l]).T @ np.abs(d) / B
            mag_db[l] = np.abs(d).mean(axis=0)
            dn = d @ self.W[l].T
            mg = np.abs(d) @ np.abs(self.W[l]).T
            dn = dn[:, :F * D] * (A[0][:, :F * D] > 0) if l == 0 else dn * (A[l] > 0)
            deltas[l] = dn                                          # delta INTO layer l (relu' of its input applied)
            mag_d[l] = mg[:, :F * D] if l == 0 else mg
            d = dn
        out["delta"], out["dW"], out["db"] = deltas, dW, db
        out["mag_delta"], out["mag_dW"], out["mag_db"] = mag_d, mag_dW, mag_db
        # EmbeddingField.backward twice + KVStore.sum aliasing: g_eff = S (n + 1) / (2 n^2)
        dx = deltas[0]
        geff = []
        for f in range(F):
            ids, inv, cnt = np.unique(E[:, f], return_inverse=True, return_counts=True)
            S = np.zeros((len(ids), D), f64)
            np.add.at(S, inv, dx[:, f * D:(f + 1) * D])
            n = cnt.astype(f64).reshape(-1, 1)
            geff.append((ids, S * (n + 1) / (2 * n * n)))
        out["geff"] = geff
        if not update:
            return out
        for f in range(F):
            ids, g = geff[f]
            for k, i in enumerate(ids):
                r = self.rows[f][int(i)]
                if self.emb_updater == "adam":
                    r[0], r[1], r[2] = adam(r[0], g[k], r[1], r[2])
                else:
                    r[0], r[1], r[2] = ftrl(r[0], g[k], r[1], r[2])
        for l in range(nfc):
            self.W[l], self.S[l][0], self.S[l][1] = adam(self.W[l], dW[l], self.S[l][0], self.S[l][1])
            self.b[l], self.S[l][2], self.S[l][3] = adam(self.b[l], db[l], self.S[l][2], self.S[l][3])
        if self.wide:
            self.seen[np.unique(Wd)] = True
            k = np.nonzero(self.seen)[0]
            g = np.full(len(k), out["gbar"], f64)
            if out["gbar"] != 0:
                w_, z_, n_ = ftrl(self.ww[k].reshape(-1, 1), g.reshape(-1, 1), self.wz[k].reshape(-1, 1), self.wn[k].reshape(-1, 1))
                self.ww[k], self.wz[k], self.wn[k] = w_[:, 0], z_[:, 0], n_[:, 0]
                self.wb, self.wbz, self.wbn = ftrl(self.wb, np.array([out["gbar"]]), self.wbz, self.wbn)
        return out


def bound(x_gpu, x_orc, x64, what, rtol=1e-5, c=4.0, floor=0.0):
    """|gpu - x64| <= rtol (|x64| + max|x64|) + c max|orc - x64| + floor, elementwise (floor: the roundoff floor of x's own
    last operation, 8 eps sum|terms|, where a caller has it).  Returns (max gpu error, max oracle error) for reporting."""
    g, o, t = (np.asarray(a, f64) for a in (x_gpu, x_orc, x64))
    eg, eo = np.abs(g - t), np.abs(o - t)
    lim = rtol * (np.abs(t) + (np.abs(t).max() if t.size else 0.0)) + c * (eo.max() if eo.size else 0.0) + floor
    excess = eg - lim
    assert excess.size == 0 or excess.max() <= 0, "%s: |gpu - f64| exceeds 1e-5 (|f64| + max|f64|) + %g max|oracle - f64| by %.3e (max gpu err %.3e, max oracle err %.3e, max|f64| %.3e)" % (
        what, c, excess.max(), eg.max(), eo.max(), np.abs(t).max())
    return (float(eg.max()) if eg.size else 0.0), (float(eo.max()) if eo.size else 0.0)
