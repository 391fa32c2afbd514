"""CPU restatement (numpy, float32) of the reference's stage-1 Wasserstein retrieval.

TEST INFRASTRUCTURE ONLY.  Parity status: PINNED against outputs of the reference itself
(src/retrieval/wasserstein.py imported by oracle/gen_golden_retrieval.py; tests/golden/wasserstein.npz).
"""
import numpy as np


def batch(query, database, eps=1e-8):
    """wasserstein_distance_batch_torch, wasserstein.py:134-172 (float32 throughout)."""
    q = np.asarray(query, np.float32)
    db = np.asarray(database, np.float32)
    qs = q.sum(dtype=np.float32)
    if qs > eps:
        q = q / qs                                            # :153-155 (no epsilon in the denominator)
    ds = db.sum(axis=1, keepdims=True, dtype=np.float32)
    db = np.where(ds > eps, db / (ds + np.float32(eps)), db)  # :158-163
    qc = np.cumsum(q, dtype=np.float32)
    dc = np.cumsum(db, axis=1, dtype=np.float32)
    return np.abs(dc - qc[None, :]).sum(axis=1, dtype=np.float32)      # :170


def matrix(h1, h2=None, eps=1e-8):
    """wasserstein_distance_matrix_torch, wasserstein.py:232-273."""
    h1 = np.asarray(h1, np.float32)
    h2 = h1 if h2 is None else np.asarray(h2, np.float32)

    def norm(h):
        s = h.sum(axis=1, keepdims=True, dtype=np.float32)
        return np.where(s > eps, h / (s + np.float32(eps)), h)
    c1 = np.cumsum(norm(h1), axis=1, dtype=np.float32)
    c2 = np.cumsum(norm(h2), axis=1, dtype=np.float32)
    return np.abs(c1[:, None, :] - c2[None, :, :]).sum(axis=2, dtype=np.float32)


def one_d(h1, h2, eps=1e-8):
    """wasserstein_distance_1d_numpy / _torch, wasserstein.py:20-87: both histograms divided by their plain sum."""
    h1, h2 = np.asarray(h1, np.float32), np.asarray(h2, np.float32)
    s1, s2 = h1.sum(dtype=np.float32), h2.sum(dtype=np.float32)
    if s1 > eps:
        h1 = h1 / s1
    if s2 > eps:
        h2 = h2 / s2
    return float(np.abs(np.cumsum(h1, dtype=np.float32) - np.cumsum(h2, dtype=np.float32)).sum(dtype=np.float32))


def topk(dist, k):
    """WassersteinRetriever.query, wasserstein.py:360-366: k smallest, ascending (ties: lower index)."""
    order = np.lexsort((np.arange(len(dist)), dist))[:k]
    return order, dist[order]


def stage1(desc, poses, has_pose, query, top_k=10, min_dist=50.0, eps=1e-8):
    """Stage 1 of TwoStageRetrieval for one query keyframe of the database itself
    (two_stage_retrieval.py:145-202): rows closer than min_dist to the query (both poses known, translation
    distance, :160-170) are skipped, the rest ranked by W1 (wasserstein.py:134-172), top_k returned.
    PINNED by tests/golden/two_stage.npz (oracle/gen_golden_two_stage.py)."""
    desc = np.asarray(desc, np.float32)
    pos = np.asarray(poses, np.float64)[:, :3, 3]
    d = batch(desc[query], desc, eps).astype(np.float64)
    if has_pose[query]:
        near = np.linalg.norm(pos - pos[query], axis=1) < min_dist
        d = np.where(near & np.asarray(has_pose, bool), np.inf, d)
    order = np.lexsort((np.arange(len(d)), d))
    order = order[np.isfinite(d[order])][:top_k]
    return order, d[order]
