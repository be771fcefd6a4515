"""The training step of the reference in FLOAT64, as a chain: forward, loss, backward, per-key embedding gradient,
updaters, over as many steps as a test runs -- the yardstick the end-to-end tolerances are stated against
(VERDICT r2 next #6).

The float32 implementations under test (the HIP path; the C oracle, whose sgemm order is a guess at jblas') both
differ from the exact result by their own propagated roundoff.  With x64 the float64 chain started from the SAME
initial parameters and fed the SAME batches, the tests assert for every tensor x of a step

        |x_gpu - x64|  <=  RTOL * |x64|  +  C * max|x_oracle - x64|  +  floor(x)

i.e. the HIP path may be as far from the exact chain as north_star's 1e-5 relative, ELEMENTWISE, plus a small multiple of
how far the reference-order float32 chain itself is, plus floor(x): how far ANY correctly rounded float32 evaluation of the
same chain may lie from the exact one at that element -- the roundoff of every contraction (8 eps sum|terms|), carried
forward through |W|, through the head's Lipschitz constants, back through the backward GEMMs, through relu' (an activation
whose sign the bound cannot decide contributes the whole unmasked value) and through the updaters (their sensitivity to
the gradient's and the state's bounds, evaluated at the corners), step after step.  Round 5: these
propagated floors replace the tensor-wide RTOL * max|x64| term -- no quantity passes any more because it is within 1e-5
of its tensor's LARGEST element; no hand-picked absolute floors, no loosened end-to-end rtol.

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


EPS32 = f64(np.finfo(f32).eps)


def _corners(fn, args, errs, clip0=()):
    """max over the 2^k corners of |fn(args +- errs) - fn(args)|, per output; clip0: indices of args kept >= 0 (V, n)"""
    base = fn(*args)
    worst = [np.zeros_like(np.asarray(b, f64)) for b in base]
    k = len(args)
    for m in range(1, 2 ** k):
        pert = []
        for i in range(k):
            sgn = 1.0 if (m >> i) & 1 else -1.0
            a = np.asarray(args[i], f64) + sgn * np.asarray(errs[i], f64)
            if i in clip0:
                a = np.maximum(a, 0.0)
            pert.append(a)
        out = fn(*pert)
        for j in range(len(base)):
            worst[j] = np.maximum(worst[j], np.abs(np.asarray(out[j], f64) - np.asarray(base[j], f64)))
    return base, worst


def adam_floor(w, g, m, v, ew, eg, em, ev):
    """how far a float32 Adam step fed (w, g, m, v) +- (ew, eg, em, ev) may end from the exact one: the corners of (g, m, v)
    (w enters additively) + 8 eps of every result"""
    _, (dw, dm, dv) = _corners(lambda g_, m_, v_: adam(np.zeros_like(np.asarray(w, f64)), g_, m_, v_), (g, m, v), (eg, em, ev), clip0=(2,))
    w1, m1, v1 = adam(w, g, m, v)
    return ew + dw + 8 * EPS32 * (np.abs(w1) + np.abs(w1 - w)), dm + 8 * EPS32 * np.abs(m1), dv + 8 * EPS32 * np.abs(v1)


def _ftrl_live(g, z, n):
    """FtrlUpdater.java:64-74 for a row that is NOT skipped, elementwise"""
    sign = np.where(z >= 0, 1.0, -1.0)
    wn = np.where(np.abs(z) <= F_L1, 0.0, -(z - sign * F_L1) / ((F_L2 + (F_BETA + np.sqrt(n))) / F_ALFA))
    s = np.sqrt(n + g * g) - np.sqrt(n / F_ALFA)
    return wn, z + (g - s * wn), n + g * g


def ftrl_floor(w, g, z, n, ew, eg, ez, en):
    """... and a float32 Ftrl step as written (w from the OLD z, n).  FtrlUpdater.java:52 skips a row whose g[0] is exactly 0: a
    row whose g[0] lies within its bound of 0 may have been skipped on one side and updated on the other -- it gets the distance
    between the two outcomes on top."""
    w, g, z, n, ew, eg, ez, en = (np.asarray(a, f64) for a in (w, g, z, n, ew, eg, ez, en))
    (wl, zl, nl), (dw, dz, dn) = _corners(_ftrl_live, (g, z, n), (eg, ez, en), clip0=(2,))
    fw = dw + 8 * EPS32 * np.abs(wl); fz = dz + 8 * EPS32 * (np.abs(zl) + np.abs(zl - z)); fn_ = dn + 8 * EPS32 * np.abs(nl)
    if g.ndim > 1:
        g0 = g.reshape(g.shape[0], -1)[:, 0]; e0 = eg.reshape(g.shape[0], -1)[:, 0]
        shp = (-1,) + (1,) * (g.ndim - 1)
    else:
        g0 = g.reshape(-1)[:1]; e0 = eg.reshape(-1)[:1]
        shp = (1,) * max(g.ndim, 1)
    skipped = (g0 == 0).reshape(shp); amb = (np.abs(g0) <= e0).reshape(shp) & ((e0 > 0).reshape(shp) | skipped)
    # a row skipped on both sides keeps its bounds; an ambiguous one: the larger outcome distance on top of the live bounds
    fw_s, fz_s, fn_s = ew + 0 * fw, ez + 0 * fz, en + 0 * fn_
    certain_skip = skipped & ~((e0 > 0).reshape(shp))
    fw = np.where(certain_skip, fw_s, np.where(amb, fw + np.abs(wl - w) + ew, fw))
    fz = np.where(certain_skip, fz_s, np.where(amb, fz + np.abs(zl - z) + ez, fz))
    fn_ = np.where(certain_skip, fn_s, np.where(amb, fn_ + np.abs(nl - n) + en, fn_))
    return fw, fz, fn_


KSIG = 8.0          # a floor is KSIG standard deviations of the propagated float32 noise
RND = 4.0           # variance of one contraction's roundoff: RND eps^2 sum (a w)^2 (partial sums included)


class Chain:
    """Floors.  Beside every quantity x the chain carries v_x, the VARIANCE of what a correctly rounded float32 evaluation (any
    summation order) adds to it, propagated in quadrature: a contraction z = A W + b turns (v_A, v_W, v_b) into
    v_A W^2 + A^2 v_W + v_b + RND eps^2 (A^2 W^2 + b^2); relu passes it on, the clipped sigmoid scales it by its slope, relu' of an
    input whose sign KSIG sigma cannot decide contributes the whole unmasked value, the updaters are evaluated at the corners of
    (g, state) +- KSIG sigma.  floor(x) = KSIG sqrt(v_x) (e_* below).  anchor(): a test hands in the parameters the float32 side
    really holds at the start of a step; their measured distance to the chain's replaces the carried variances (squared), so
    the floors never compound over steps (one Adam step on a gradient whose sign the noise cannot decide moves a weight by
    2 alfa on one side only: carried forward as a bound it would drown every later floor)."""

    def __init__(self, wide, F, D, X, fc, wide_size=0, emb_updater="adam"):
        self.wide, self.F, self.D, self.X, self.fc, self.ws = wide, F, D, X, list(fc), wide_size
        self.dims = [F * D + X] + list(fc)
        self.rows = [dict() for _ in range(F)]          # id -> [w, s1, s2, v_w, v_s1, v_s2]
        self.W, self.b, self.S = [], [], []             # W[l]: [in][out]
        self.vW, self.vb, self.vS = [], [], []
        self.emb_updater = emb_updater
        self.row_anchor = None
        self.nsteps = 0                                 # updates applied so far (anchor()'s a-priori cap)
        if wide:
            self.ww = np.zeros(wide_size, f64); self.wz = np.zeros(wide_size, f64); self.wn = np.zeros(wide_size, f64)
            self.wb = np.zeros(1, f64); self.wbz = np.zeros(1, f64); self.wbn = np.zeros(1, f64)
            self.seen = np.zeros(wide_size, bool)       # LRLayer.weights membership: never cleared
            self.vww = np.zeros(wide_size, f64); self.vwz = np.zeros(wide_size, f64); self.vwn = np.zeros(wide_size, f64)
            self.vwb = np.zeros(1, f64); self.vwbz = np.zeros(1, f64); self.vwbn = np.zeros(1, f64)

    def load_fc(self, weights, biases):
        """weights[l]: the store's "fc<l>.weights" ([in][out] row-major, flat), biases[l]: [out]"""
        self.W = [np.asarray(w, f64).reshape(self.dims[l], self.dims[l + 1]).copy() for l, w in enumerate(weights)]
        self.b = [np.asarray(b, f64).copy() for b in biases]
        self.S = [[np.zeros_like(w), np.zeros_like(w), np.zeros_like(b), np.zeros_like(b)] for w, b in zip(self.W, self.b)]
        self.vW = [np.zeros_like(w) for w in self.W]; self.vb = [np.zeros_like(b) for b in self.b]
        self.vS = [[np.zeros_like(w), np.zeros_like(w), np.zeros_like(b), np.zeros_like(b)] for w, b in zip(self.W, self.b)]

    def anchor(self, weights=None, biases=None, rows=None, wide_w=None, wide_b=None):
        """The parameters the float32 side holds NOW (before the next step): weights[l] / biases[l] as load_fc takes them, rows(f, ids)
        -> its rows of field f, wide_w [wide_size], wide_b.  Their distance to the chain's becomes the parameters' noise."""
        if weights is not None:
            self.vW = [(np.asarray(w, f64).reshape(self.W[l].shape) - self.W[l]) ** 2 for l, w in enumerate(weights)]
            for l in range(len(self.vW)):
                self._cap_anchor(np.sqrt(self.vW[l]), self.W[l], "fc%d.weights" % l)
        if biases is not None:
            self.vb = [(np.asarray(b, f64).reshape(-1) - self.b[l]) ** 2 for l, b in enumerate(biases)]
            for l in range(len(self.vb)):
                self._cap_anchor(np.sqrt(self.vb[l]), self.b[l], "fc%d.bias" % l)
        self.row_anchor = rows
        if self.wide and wide_w is not None:
            self.vww = (np.asarray(wide_w, f64).reshape(-1) - self.ww) ** 2
        if self.wide and wide_b is not None:
            self.vwb = (np.asarray(wide_b, f64).reshape(-1)[:1] - self.wb) ** 2

    def _cap_anchor(self, dev, w, what, alfa=0.005):
        """the anchored distance becomes the next step's noise, so a float32 side that drifts systematically would widen
        its own tolerance.  An a-priori model of what n Adam steps in float32 may add caps it: every element within 2 alfa n (a
        gradient whose sign the noise cannot decide moves the weight by alfa on either side, once per step) + 64 eps |w| sqrt(n),
        and all but a few per cent of the elements within the rounding term alone, 64 eps (|w| + alfa) sqrt(n)."""
        n = max(self.nsteps, 1)
        if dev.size == 0 or self.nsteps == 0:
            assert dev.size == 0 or dev.max() <= 64 * EPS32 * np.abs(w).max() + 1e-30, "%s: anchored %.3e away before any step" % (what, dev.max())
            return
        hard = 2.0 * alfa * n + 64 * EPS32 * np.abs(w) * np.sqrt(n)
        assert (dev <= hard).all(), "%s: anchored deviation %.3e exceeds the a-priori float32 cap %.3e after %d steps" % (
            what, (dev - hard).max() + hard.flat[np.argmax(dev - hard)], hard.flat[np.argmax(dev - hard)], n)
        soft = 64 * EPS32 * (np.abs(w) + alfa) * np.sqrt(n)
        frac = float((dev > soft).mean())
        assert frac <= 0.05, "%s: %.1f %% of the elements lie further from the float64 chain than float32 rounding explains after %d steps" % (what, 100 * frac, n)

    # floors (KSIG sigma) of the parameters as they are now
    def floor_W(self, l): return KSIG * np.sqrt(self.vW[l])
    def floor_b(self, l): return KSIG * np.sqrt(self.vb[l])
    def floor_rows(self, f, ids, which=0): return KSIG * np.sqrt(np.stack([self.rows[f][int(i)][3 + which] for i in ids]))
    def floor_wide(self, k=None): return KSIG * np.sqrt(self.vww if k is None else self.vww[k])
    def floor_wide_bias(self): return KSIG * np.sqrt(self.vwb)

    def step(self, E, Xd, Y, Wd, init_rows, update=True):
        """One training step.  init_rows(f, ids) -> the initial float32 rows of ids this chain has not seen yet.
        Every returned quantity x comes with e_x = its float32 floor (the class's docstring)."""
        F, D, nfc = self.F, self.D, len(self.fc)
        B = E.shape[0]
        E2 = EPS32 * EPS32 * RND
        for f in range(F):
            ids_f = np.unique(E[:, f])
            new = [int(i) for i in ids_f if int(i) not in self.rows[f]]
            if new:
                r = np.asarray(init_rows(f, np.array(new, np.int64)), f64)
                for k, i in enumerate(new):
                    self.rows[f][i] = [r[k].copy(), np.zeros(D, f64), np.zeros(D, f64), np.zeros(D, f64), np.zeros(D, f64), np.zeros(D, f64)]
            if self.row_anchor is not None:
                have = np.asarray(self.row_anchor(f, ids_f), f64)
                for k, i in enumerate(ids_f):
                    r = self.rows[f][int(i)]
                    r[3] = (have[k] - r[0]) ** 2
        Z0 = np.concatenate([np.stack([self.rows[f][int(i)][0] for i in E[:, f]]) for f in range(F)], axis=1)      # [B][F*D]
        vZ0 = np.concatenate([np.stack([self.rows[f][int(i)][3] for i in E[:, f]]) for f in range(F)], axis=1)
        A0 = np.maximum(Z0, 0)
        A = [np.concatenate([A0, np.asarray(Xd, f64)], axis=1)]
        vA = [np.concatenate([vZ0, np.zeros((B, self.X), f64)], axis=1)]         # relu is 1-Lipschitz; the dense features are exact
        Z = [np.concatenate([Z0, np.ones((B, self.X), f64)], axis=1)]             # pre-activations (their sign decides relu')
        for l in range(nfc):
            z = A[l] @ self.W[l] + self.b[l]
            vz = vA[l] @ self.W[l] ** 2 + A[l] ** 2 @ self.vW[l] + self.vb[l] + E2 * (A[l] ** 2 @ self.W[l] ** 2 + self.b[l] ** 2)
            Z.append(z)
            if l < nfc - 1:
                z = np.maximum(z, 0)
            elif not self.wide:
                z = sigmoid_clip(z)
                vz = 0.0625 * vz + E2
            A.append(z); vA.append(vz)
        out = {"A": A, "e_A": [KSIG * np.sqrt(v) for v in vA]}
        Yc = np.asarray(Y, f64).reshape(B, 1)
        if self.wide:
            zw = self.ww[Wd].sum(axis=1, keepdims=True) + self.wb[0]
            vzw = self.vww[Wd].sum(axis=1, keepdims=True) + self.vwb[0] + E2 * ((self.ww[Wd] ** 2).sum(axis=1, keepdims=True) + self.wb[0] ** 2)
            P = sigmoid_clip(A[nfc] + zw)
            vP = 0.0625 * (vA[nfc] + vzw) + E2
        else:
            P = A[nfc]; vP = vA[nfc]
        out["P"] = P[:, 0]; out["e_P"] = KSIG * np.sqrt(vP[:, 0])
        terms = -Yc * np.log(P) - (1 - Yc) * np.log(1 - P)
        out["loss"] = float(np.mean(terms))
        out["e_loss"] = float(KSIG * np.sqrt(np.sum(vP / np.minimum(P, 1 - P) ** 2 + E2 * terms ** 2)) / B)
        d = ((P - Yc) / (P * (1 - P))) * (P * (1 - P))              # CrossEntropy' then Sigmoid' (y (1 - y) of the CLIPPED y)
        vd = vP + E2 * (d ** 2 + Yc ** 2)                            # (+ the float32 side's own (p - l) / (p (1 - p)) * p (1 - p))
        out["gbar"] = float(d.mean()); v_gbar = float(np.sum(vd + E2 * d ** 2)) / (B * B)
        out["e_gbar"] = KSIG * np.sqrt(v_gbar)
        deltas, dW, db = [None] * nfc, [None] * nfc, [None] * nfc
        v_dW, v_db = [None] * nfc, [None] * nfc
        e_delta = [None] * nfc
        # mag_*: sum of |terms| of each quantity's own last contraction (its float32 roundoff floor is ~ 8 eps * that)
        mag_d, mag_dW, mag_db = [None] * nfc, [None] * nfc, [None] * nfc
        for l in range(nfc - 1, -1, -1):
            dW[l] = A[l].T @ d / B
            db[l] = d.mean(axis=0)
            mag_dW[l] = np.abs(A[l]).T @ np.abs(d) / B
            mag_db[l] = np.abs(d).mean(axis=0)
            v_dW[l] = (vA[l].T @ d ** 2 + (A[l] ** 2).T @ vd + E2 * ((A[l] ** 2).T @ d ** 2)) / (B * B)
            v_db[l] = (vd + E2 * d ** 2).sum(axis=0) / (B * B)
            dn = d @ self.W[l].T
            mg = np.abs(d) @ np.abs(self.W[l]).T
            vdn = vd @ (self.W[l] ** 2).T + d ** 2 @ self.vW[l].T + E2 * (d ** 2 @ (self.W[l] ** 2).T)
            # relu' of layer l's INPUT: its pre-activation's sign.  Where KSIG sigma cannot decide it, the two sides may differ by
            # the whole unmasked value
            zin, vzin = (Z[0][:, :F * D], vA[0][:, :F * D]) if l == 0 else (Z[l], vA[l])
            if l == 0:
                dn, mg, vdn = dn[:, :F * D], mg[:, :F * D], vdn[:, :F * D]
            on = zin > 0
            unsure = np.abs(zin) <= KSIG * np.sqrt(vzin)
            vdn = np.where(unsure, (dn / KSIG) ** 2 + vdn, np.where(on, vdn, 0.0))
            dn = dn * on
            deltas[l] = dn                                          # delta INTO layer l (relu' of its input applied)
            e_delta[l] = KSIG * np.sqrt(vdn)
            mag_d[l] = mg
            d, vd = dn, vdn
        out["delta"], out["dW"], out["db"] = deltas, dW, db
        out["e_delta"], out["e_dW"], out["e_db"] = e_delta, [KSIG * np.sqrt(v) for v in v_dW], [KSIG * np.sqrt(v) for v in v_db]
        out["mag_delta"], out["mag_dW"], out["mag_db"] = mag_d, mag_dW, mag_db
        # EmbeddingField.backward twice + KVStore.sum aliasing: g_eff = S (n + 1) / (2 n^2)
        dx, vdx = deltas[0], vd
        geff, v_geff = [], []
        for f in range(F):
            ids, inv, cnt = np.unique(E[:, f], return_inverse=True, return_counts=True)
            S = np.zeros((len(ids), D), f64); vS = np.zeros((len(ids), D), f64); qS = np.zeros((len(ids), D), f64)
            np.add.at(S, inv, dx[:, f * D:(f + 1) * D])
            np.add.at(vS, inv, vdx[:, f * D:(f + 1) * D])
            np.add.at(qS, inv, dx[:, f * D:(f + 1) * D] ** 2)
            n = cnt.astype(f64).reshape(-1, 1)
            fac = (n + 1) / (2 * n * n)
            geff.append((ids, S * fac))
            # (the reference's order: 2 n sequential adds and two divisions -- a partial sum is rounded ~n times)
            v_geff.append((vS + E2 * (2 * n + 4) * qS) * fac ** 2)
        out["geff"] = geff; out["e_geff"] = [KSIG * np.sqrt(v) for v in v_geff]
        if not update:
            return out
        self.nsteps += 1
        K2 = KSIG * KSIG
        for f in range(F):
            ids, g = geff[f]
            eg = KSIG * np.sqrt(v_geff[f])
            rr = [self.rows[f][int(i)] for i in ids]
            w = np.stack([r[0] for r in rr]); s1 = np.stack([r[1] for r in rr]); s2 = np.stack([r[2] for r in rr])
            ew = KSIG * np.sqrt(np.stack([r[3] for r in rr])); e1 = KSIG * np.sqrt(np.stack([r[4] for r in rr])); e2 = KSIG * np.sqrt(np.stack([r[5] for r in rr]))
            if self.emb_updater == "adam":
                fw, f1, f2 = adam_floor(w, g, s1, s2, ew, eg, e1, e2)
                w, s1, s2 = adam(w, g, s1, s2)
            else:
                fw, f1, f2 = ftrl_floor(w, g, s1, s2, ew, eg, e1, e2)
                w, s1, s2 = ftrl(w, g, s1, s2)
            for k, r in enumerate(rr):
                r[0], r[1], r[2], r[3], r[4], r[5] = w[k], s1[k], s2[k], fw[k] ** 2 / K2, f1[k] ** 2 / K2, f2[k] ** 2 / K2
        for l in range(nfc):
            fW, f1, f2 = adam_floor(self.W[l], dW[l], self.S[l][0], self.S[l][1], self.floor_W(l), KSIG * np.sqrt(v_dW[l]), KSIG * np.sqrt(self.vS[l][0]), KSIG * np.sqrt(self.vS[l][1]))
            fb, f3, f4 = adam_floor(self.b[l], db[l], self.S[l][2], self.S[l][3], self.floor_b(l), KSIG * np.sqrt(v_db[l]), KSIG * np.sqrt(self.vS[l][2]), KSIG * np.sqrt(self.vS[l][3]))
            self.vW[l], self.vS[l][0], self.vS[l][1] = fW ** 2 / K2, f1 ** 2 / K2, f2 ** 2 / K2
            self.vb[l], self.vS[l][2], self.vS[l][3] = fb ** 2 / K2, f3 ** 2 / K2, f4 ** 2 / K2
            self.W[l], self.S[l][0], self.S[l][1] = adam(self.W[l], dW[l], self.S[l][0], self.S[l][1])
            self.b[l], self.S[l][2], self.S[l][3] = adam(self.b[l], db[l], self.S[l][2], self.S[l][3])
        if self.wide:
            self.seen[np.unique(Wd)] = True
            k = np.nonzero(self.seen)[0]
            g = np.full(len(k), out["gbar"], f64)
            eg = np.full(len(k), out["e_gbar"], f64)
            col = lambda a: np.asarray(a, f64).reshape(-1, 1)      # noqa: E731
            if out["gbar"] != 0 or out["e_gbar"] > 0:
                fw, fz, fn_ = ftrl_floor(col(self.ww[k]), col(g), col(self.wz[k]), col(self.wn[k]),
                                         col(KSIG * np.sqrt(self.vww[k])), col(eg), col(KSIG * np.sqrt(self.vwz[k])), col(KSIG * np.sqrt(self.vwn[k])))
                self.vww[k], self.vwz[k], self.vwn[k] = fw[:, 0] ** 2 / K2, fz[:, 0] ** 2 / K2, fn_[:, 0] ** 2 / K2
                fb = ftrl_floor(col(self.wb), col([out["gbar"]]), col(self.wbz), col(self.wbn),
                                col(KSIG * np.sqrt(self.vwb)), col([out["e_gbar"]]), col(KSIG * np.sqrt(self.vwbz)), col(KSIG * np.sqrt(self.vwbn)))
                self.vwb, self.vwbz, self.vwbn = fb[0].reshape(1) ** 2 / K2, fb[1].reshape(1) ** 2 / K2, fb[2].reshape(1) ** 2 / K2
            if out["gbar"] != 0:
                w_, z_, n_ = ftrl(self.ww[k].reshape(-1, 1), g.reshape(-1, 1), self.wz[k].reshape(-1, 1), self.wn[k].reshape(-1, 1))
                self.ww[k], self.wz[k], self.wn[k] = w_[:, 0], z_[:, 0], n_[:, 0]
                self.wb, self.wbz, self.wbn = ftrl(self.wb, np.array([out["gbar"]]), self.wbz, self.wbn)
        return out


def bound(x_gpu, x_orc, x64, what, rtol=1e-5, c=4.0, floor=0.0):
    """|gpu - x64| <= rtol |x64| + c |orc - x64| + floor, ELEMENTWISE in every term (round 6: the oracle's own error counts at the
    element it occurred in, no longer as a tensor-wide maximum) -- floor: the chain's propagated float32 floor of x (the
    e_* entries of Chain.step, Chain.eW / eb / row_floor: how far any correctly rounded float32 evaluation may lie from the exact
    chain at that element).  No tensor-wide term in |x64| (round 5).  Returns (max gpu error, max oracle error) for reporting."""
    g, o, t = (np.asarray(a, f64) for a in (x_gpu, x_orc, x64))
    eg, eo = np.abs(g - t), np.abs(o - t)
    lim = rtol * np.abs(t) + c * eo + np.asarray(floor, f64)
    excess = eg - lim
    assert excess.size == 0 or excess.max() <= 0, "%s: |gpu - f64| exceeds 1e-5 |f64| + %g |oracle - f64| + its float32 floor (elementwise) by %.3e (max gpu err %.3e, max oracle err %.3e, max|f64| %.3e, max floor %.3e)" % (
        what, c, excess.max(), eg.max(), eo.max(), np.abs(t).max(), float(np.max(floor)) if np.size(floor) else 0.0)
    return (float(eg.max()) if eg.size else 0.0), (float(eo.max()) if eo.size else 0.0)
