#!/usr/bin/env python3
"""tests/golden/miner.npz from the reference's own TripletMiner (build container only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
from gnn.triplet_miner import TripletMiner   # noqa: E402

rng = np.random.default_rng(0)
# two sequences; each drives a loop twice so revisits (positives) exist
def loop_traj(n, radius, seed):
    r = np.random.default_rng(seed)
    t = np.linspace(0, 4 * np.pi, n)          # two laps
    xy = radius * np.stack([np.cos(t), np.sin(t)], 1) + r.normal(0, 0.4, (n, 2))
    poses = np.tile(np.eye(4), (n, 1, 1))
    poses[:, 0, 3], poses[:, 1, 3] = xy[:, 0], xy[:, 1]
    return poses
poses = np.concatenate([loop_traj(260, 40.0, 1), loop_traj(200, 25.0, 2)], 0)
seq = np.concatenate([np.zeros(260, int), np.ones(200, int)])
desc = (rng.random((460, 800)) ** 4).astype(np.float32)
desc /= desc.sum(1, keepdims=True)
np.random.seed(0)
trip = np.array(TripletMiner().mine_triplets(desc, poses, 1, seq))
np.random.seed(1)
trip2 = np.array(TripletMiner().mine_triplets(desc, poses, 2, seq))
np.random.seed(2)
trip_semi = np.array(TripletMiner(mining_strategy="semi-hard").mine_triplets(desc, poses, 1, seq))   # :352-357
print(trip.shape, trip2.shape, trip_semi.shape, trip[:5])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "miner.npz"), poses=poses, seq=seq, desc=desc,
                    triplets=trip, triplets2=trip2, triplets_semi=trip_semi)
