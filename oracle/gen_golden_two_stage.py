#!/usr/bin/env python3
"""tests/golden/two_stage.npz: stage 1 of the reference's two-stage loop closing (build container only).

src/retrieval/two_stage_retrieval.py cannot be imported here (it pulls in open3d through geometric_verification.py),
so the fixture is produced from the pieces of the reference that CAN be imported -- ``WassersteinRetriever`` in its
torch form (what ``create_two_stage_retrieval(use_gpu=True)`` constructs on a GPU host, pipeline.py:91-94; run on
the CPU here), ``euclidean_distance`` (data/pose_utils.py) -- driven by the candidate loop of ``_global_retrieval``
(:156-200) restated line by line below.  (The numpy form cannot serve: ``_global_retrieval`` asks for
top_k = len(database) and wasserstein.py:380 then calls np.argpartition with kth == n, which raises.)
Inputs and expected outputs only; no reference text is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
from retrieval.wasserstein import WassersteinRetriever          # noqa: E402
from data.pose_utils import euclidean_distance                  # noqa: E402

rng = np.random.default_rng(3)
n, top_k, thr = 400, 10, 50.0
# a trajectory that folds back on itself: revisits exist, and many rows lie inside the 50 m filter radius of a query
t = np.linspace(0, 6 * np.pi, n)
poses = np.tile(np.eye(4), (n, 1, 1))
poses[:, 0, 3] = 120 * np.cos(t) + rng.normal(0, 1.0, n)
poses[:, 1, 3] = 70 * np.sin(2 * t) + rng.normal(0, 1.0, n)
poses[:, 2, 3] = rng.normal(0, 0.2, n)
has_pose = np.ones(n, bool)
has_pose[[5, 77, 200]] = False                 # keyframes without a pose are never filtered (:163)
place = (np.arange(n) % 133)                   # revisited places share a base descriptor
base = (rng.random((133, 800)) ** 4).astype(np.float32)
desc = base[place] * (1 + 0.05 * rng.random((n, 800)).astype(np.float32))
desc = (desc / desc.sum(1, keepdims=True)).astype(np.float32)
queries = [3, 150, 399, 77, 260]               # 77 has no pose: nothing is filtered for it

retriever = WassersteinRetriever(use_torch=True, device="cpu")
for i in range(n):                                      # add_keyframe (:91-105): one row at a time
    retriever.add_to_database(desc[i].reshape(1, -1))

out_idx = np.full((len(queries), top_k), -1, np.int64)
out_dist = np.full((len(queries), top_k), np.inf, np.float64)
n_valid = []
for qi, q in enumerate(queries):
    valid_indices = []                                  # :157-170
    for i in range(n):
        if has_pose[q] and has_pose[i]:
            if euclidean_distance(poses[q], poses[i]) < thr:
                continue
        valid_indices.append(i)
    n_valid.append(len(valid_indices))
    k = min(top_k, len(valid_indices))                  # :176
    indices, distances = retriever.query(desc[q], top_k=n)      # :182-185 "get all, will filter manually"
    valid = set(valid_indices)
    c = 0
    for idx, d in zip(indices, distances):              # :190-200
        if idx in valid:
            out_idx[qi, c], out_dist[qi, c] = idx, d
            c += 1
            if c >= k:
                break
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "two_stage.npz"), desc=desc, poses=poses,
                    has_pose=has_pose, queries=np.asarray(queries), top_k=top_k, thr=thr, idx=out_idx, dist=out_dist,
                    n_valid=np.asarray(n_valid))
print(out_idx, out_dist[:, :3], n_valid)
