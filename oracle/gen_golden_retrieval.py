#!/usr/bin/env python3
"""tests/golden/wasserstein.npz from the reference's own functions (build container only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
from retrieval.wasserstein import (wasserstein_distance_batch_torch, wasserstein_distance_matrix_torch,  # noqa: E402
                                   wasserstein_distance_1d_numpy, wasserstein_distance_batch_numpy,
                                   wasserstein_distance_matrix_numpy, WassersteinRetriever)

torch.set_num_threads(1)
rng = np.random.default_rng(0)
db = (rng.random((300, 800)) ** 4).astype(np.float32)
db /= db.sum(1, keepdims=True)
db[7] = 0.0                                   # empty histogram row: stays unnormalised (:158-163)
db[9] *= 3.0                                  # not normalised on input
q = (rng.random((5, 800)) ** 4).astype(np.float32)
q[1] /= q[1].sum()
q[4] = 0.0
d_batch = np.stack([wasserstein_distance_batch_torch(torch.from_numpy(x), torch.from_numpy(db)).numpy() for x in q])
d_mat = wasserstein_distance_matrix_torch(torch.from_numpy(q), torch.from_numpy(db)).numpy()
r = WassersteinRetriever(use_torch=True, device="cpu")
r.add_to_database(db[:100])
r.add_to_database(db[100:])
idx, dist = r.query(q[0], top_k=10)
small = rng.random((40, 50)).astype(np.float32)
d_small = wasserstein_distance_matrix_torch(torch.from_numpy(small)).numpy()
# the numpy-side functions (float32 inputs keep them in float32): pairs (q[i], db[j]), one batch, one matrix
pairs = np.array([[0, 0], [0, 7], [1, 9], [2, 100], [3, 299], [4, 5], [4, 7]])
d_1d = np.array([wasserstein_distance_1d_numpy(q[i], db[j]) for i, j in pairs])
d_batch_np = wasserstein_distance_batch_numpy(q[2], db)
d_mat_np = wasserstein_distance_matrix_numpy(q, db[:50])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "wasserstein.npz"), db=db, q=q, d_batch=d_batch,
                    d_mat=d_mat, top_idx=idx, top_dist=dist, small=small, d_small=d_small, pairs=pairs, d_1d=d_1d,
                    d_batch_np=d_batch_np, d_mat_np=d_mat_np)
print(d_batch[:, :3], idx, dist[:3])
