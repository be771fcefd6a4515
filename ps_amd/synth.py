"""Synthetic id generators for the benchmark and the full-size tests.

SURVEY 8d specifies "ids ~ Zipf(alpha = 1.05) over V per field".  A Zipf law *over V* is the
truncated distribution  P(rank = k) = k^-alpha / H(V, alpha),  k = 1..V  -- drawn here by
inverse CDF (a uniform draw, a binary search in the cumulative weights).  Rank k maps to id
k - 1, so id 0 is the hottest key of a field.

Rounds 1-2 drew numpy's *unbounded* Zipf and clamped to V - 1; with alpha = 1.05 that puts
~55 % of all draws on the last row of every field (a tail artefact, not the specified law).
`zipf_clamped` keeps that generator for one round of side-by-side numbers.
"""
import numpy as np

_CDF = {}


def zipf_cdf(alpha, V):
    """Cumulative weights of the truncated Zipf(alpha) law over ranks 1..V (float64, last = 1)."""
    key = (float(alpha), int(V))
    c = _CDF.get(key)
    if c is None:
        w = np.arange(1, V + 1, dtype=np.float64) ** (-float(alpha))
        c = np.cumsum(w)
        c /= c[-1]
        c[-1] = 1.0
        _CDF[key] = c
    return c


def zipf_truncated(rng, alpha, V, size):
    """ids in [0, V): rank - 1 of a truncated Zipf(alpha) draw (inverse CDF)."""
    u = rng.random(size)
    return np.searchsorted(zipf_cdf(alpha, V), u, side="left").astype(np.int64).clip(0, V - 1)


def zipf_clamped(rng, alpha, V, size):
    """Rounds 1-2's generator: unbounded Zipf, clamped (the tail piles up on id V - 1)."""
    return np.minimum(rng.zipf(alpha, size=size) - 1, V - 1).astype(np.int64)


def draw_ids(rng, alpha, V, size, generator="zipf_truncated"):
    if generator == "uniform":
        return rng.integers(0, V, size=size).astype(np.int64)
    if generator == "zipf_clamped":
        return zipf_clamped(rng, alpha, V, size)
    if generator == "zipf_truncated":
        return zipf_truncated(rng, alpha, V, size)
    raise ValueError("unknown id generator %r" % generator)


def id_stats(E):
    """Per-batch shape of a single-hot id matrix E[B][F]: unique keys and the longest run of one key in a field."""
    F = E.shape[1]
    uniq, hot = 0, 0
    for f in range(F):
        _, c = np.unique(E[:, f], return_counts=True)
        uniq += len(c)
        hot = max(hot, int(c.max()))
    return {"unique_keys": int(uniq), "hottest_run": hot, "lookups": int(E.size)}
