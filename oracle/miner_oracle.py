"""CPU restatement (numpy) of the reference's per-sequence triplet mining.

TEST INFRASTRUCTURE ONLY.  Parity status: PINNED against the reference's own TripletMiner
(src/gnn/triplet_miner.py imported by oracle/gen_golden_miner.py; tests/golden/miner.npz).
"""
import numpy as np


def w1_numpy(h1, h2, eps=1e-8):
    """wasserstein_distance_1d_numpy, src/retrieval/wasserstein.py:20-52"""
    s1, s2 = h1.sum(), h2.sum()
    if s1 > eps:
        h1 = h1 / s1
    if s2 > eps:
        h2 = h2 / s2
    return float(np.abs(np.cumsum(h1) - np.cumsum(h2)).sum())


def candidates(positions, la, pmax=5.0, ptmin=30, nmin=10.0, nmax=50.0, ntmin=30):
    """Positive / negative candidate sets of local anchor la (triplet_miner.py:165-201): the cKDTree
    ball queries are inclusive (<= r)."""
    d = np.linalg.norm(positions - positions[la], axis=1)
    gap = np.abs(np.arange(len(positions)) - la)
    notself = np.arange(len(positions)) != la
    pos = np.nonzero(notself & (d <= pmax) & (gap >= ptmin))[0]
    neg = np.nonzero(notself & (d <= nmax) & ~(d <= nmin) & (gap >= ntmin))[0]
    return pos, neg


def mine_sequence(descriptors, positions, **kw):
    """Per anchor: (positive candidates, negative candidates, hard negative = argmin W1) or None."""
    out = []
    for la in range(len(positions)):
        pos, neg = candidates(positions, la, **kw)
        if len(pos) == 0 or len(neg) == 0:
            out.append(None)
            continue
        dist = np.array([w1_numpy(descriptors[la], descriptors[j]) for j in neg])       # :339-345
        out.append((pos, neg, int(neg[np.argmin(dist)]), dist))
    return out


def semi_hard(neg, dist):
    """mining_strategy='semi-hard' (triplet_miner.py:352-357): the candidate at position len // 2 of the W1 order."""
    return int(neg[np.argsort(dist)[len(dist) // 2]])
