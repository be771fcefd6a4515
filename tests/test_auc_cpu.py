"""evaluate/AUC.java restated (oracle.auc): the TestAuc.java vector, and the closed form it equals --
(# (positive, negative) pairs with the positive ranked above, ties by input order) / (P * N)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "auc_testauc.npz")


def pair_fraction(p, y):
    """Independent of the walk: rank by (p, input index); count pairs by brute force."""
    p = np.asarray(p, np.float32).astype(np.float64); y = np.asarray(y, np.float32)
    rank = np.empty(len(p), np.int64)
    rank[np.lexsort((np.arange(len(p)), np.signbit(p) ^ True, p))] = np.arange(len(p))   # Double.compareTo: -0.0 < 0.0
    pos, neg = rank[y > 0], rank[~(y > 0)]
    return float((pos[:, None] > neg[None, :]).sum()) / (len(pos) * len(neg))


def test_testauc_vector(orc):
    z = np.load(GOLD)
    assert z["p"].size == 1000 == z["y"].size
    got = orc.auc(z["p"], z["y"])
    assert got == z["expected"][0]
    assert abs(got - pair_fraction(z["p"], z["y"])) < 1e-12
    assert 0.9 < got <= 1.0                      # a trained CTR model's test AUC


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_restatement_equals_pair_fraction_with_ties(orc, seed):
    rng = np.random.default_rng(seed)
    n = 400
    p = np.round(rng.random(n), 2).astype(np.float32)        # heavy ties
    p[::7] *= -1                                             # incl. -0.0 vs 0.0 (Double.compareTo orders them)
    y = (rng.random(n) < 0.3).astype(np.float32)
    assert abs(orc.auc(p, y) - pair_fraction(p, y)) < 1e-12


def test_degenerate_label_sets(orc):
    assert orc.auc([0.2, 0.3], [1, 1]) == 0.0                 # no negatives: x never moves
    assert np.isnan(orc.auc([0.2, 0.3], [0, 0]))              # no positives: Infinity * 0
    assert orc.auc([0.5, 0.5], [1, 0]) == 0.0 and orc.auc([0.5, 0.5], [0, 1]) == 1.0    # a tie goes by input order
