"""Validation recall.  The restatement is not pinned by reference outputs
(oracle/recall_oracle.py header); it is checked on constructed cases, and the HIP path against it."""
import numpy as np
import pytest

import recall_oracle as ro


def _loop_poses(n, seed=0, laps=2, radius=30.0):
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 2 * np.pi * laps, n)
    xy = radius * np.stack([np.cos(t), np.sin(t)], 1) + rng.normal(0, 0.3, (n, 2))
    poses = np.tile(np.eye(4), (n, 1, 1))
    poses[:, 0, 3], poses[:, 1, 3] = xy[:, 0], xy[:, 1]
    return poses


def test_oracle_constructed_cases():
    n = 200
    poses = _loop_poses(n)
    pos = poses[:, :3, 3]
    # perfect embeddings = positions: the nearest non-neighbour in embedding space is the true revisit
    emb = np.zeros((n, 8)); emb[:, :3] = pos
    r1, nq = ro.recall_loop_closure(emb, poses, 1, 5.0, 30)
    assert nq > 50 and r1 == 1.0
    # embeddings = time index: nearest candidates are temporal (|c-q| = 31), which are > 5 m away
    emb2 = np.arange(n, dtype=np.float64)[:, None] * np.ones((1, 4))
    r, nq2 = ro.recall_loop_closure(emb2, poses, 1, 5.0, 30)
    assert nq2 == nq and r < 0.2
    # no revisits at all on a straight line
    line = np.tile(np.eye(4), (100, 1, 1)); line[:, 0, 3] = np.arange(100) * 2.0
    assert ro.recall_loop_closure(np.random.rand(100, 4), line, 5, 5.0, 30) == (0.0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(300, 1), (1200, 2)])
def test_gpu_recall_matches_oracle(n, seed):
    import torch
    from neural_spectral_codec_amd.gnn.trainer import GNNTrainer
    rng = np.random.default_rng(seed)
    poses = _loop_poses(n, seed, laps=3)
    pos = poses[:, :3, 3]
    emb = np.concatenate([pos * 0.05 + rng.normal(0, 0.08, (n, 3)), rng.normal(0, 0.05, (n, 29))], 1).astype(np.float32)
    tr = GNNTrainer.__new__(GNNTrainer)            # only the recall helpers are exercised here
    tr.device = "cuda"
    et = torch.from_numpy(emb).cuda()
    for k in (1, 5, 10):
        got = tr._compute_recall_loop_closure(et, poses, k, 5.0, 30)
        ref = ro.recall_loop_closure(emb, poses, k, 5.0, 30)
        assert got[1] == ref[1] and abs(got[0] - ref[0]) < 1e-12, (k, got, ref)
    assert 0.05 < ref[0] < 1.0 or k == 10           # the case is not trivially all-right / all-wrong
