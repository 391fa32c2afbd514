"""CPU restatement (numpy, literal loops) of GNNTrainer._compute_recall_loop_closure.

TEST INFRASTRUCTURE ONLY.  Parity status: PARITY UNPINNED BY REFERENCE OUTPUTS -- the method lives in
src/gnn/trainer.py, whose module import needs torch_geometric (absent here), and the reference has no tests
for it.  The restatement follows trainer.py:306-387 statement by statement and is checked on
hand-constructed cases (tests/test_recall.py).
"""
import numpy as np


def recall_loop_closure(embeddings, poses, k, distance_threshold, skip_frames=30):
    n = len(embeddings)
    positions = poses[:, :3, 3]                                                     # :334
    pose_d = np.linalg.norm(positions[:, None, :] - positions[None, :, :], axis=2)  # :337-340
    queries = []
    for i in range(n):                                                              # :344-348
        for j in range(i + skip_frames, n):
            if pose_d[i, j] < distance_threshold:
                queries.append((j, i))
                break
    if not queries:
        return 0.0, 0
    e = np.asarray(embeddings, np.float64)
    emb_d = np.sqrt(((e[:, None, :] - e[None, :, :]) ** 2).sum(-1))                 # cdist euclidean :355
    correct = 0
    for q, _ in queries:                                                            # :360-383
        cand = [(i, emb_d[q, i], pose_d[q, i]) for i in range(n) if abs(i - q) > skip_frames]
        if not cand:
            continue
        cand.sort(key=lambda x: x[1])                                               # stable: ties by index
        for _, _, g in cand[:k]:
            if g < distance_threshold:
                correct += 1
                break
    return correct / len(queries), len(queries)
