#!/usr/bin/env python3
"""One-off soak (not part of the test suite): GPU hard-negative mining (TripletMiner, nsc_mine_triplets) on random looping
tracks -- 1-3 sequences of 60-900 keyframes, descriptor widths 50 / 800, 1-3 triplets per anchor -- against
oracle/miner_oracle.py: the same anchors produce triplets, the hard negative is the oracle's argmin-W1 candidate (or a candidate
within 2e-4 of it -- W1 is a float32 sum over up to 800 CDF bins, ~1e-7 each: the bound tests/test_retrieval.py uses), every positive lies in the oracle's candidate set.
usage: fuzz_miner.py [n_cases]"""
import os
import sys
import time

import numpy as np

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "oracle"))
import miner_oracle as mo                                                       # noqa: E402
from neural_spectral_codec_amd.gnn.triplet_miner import TripletMiner           # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(21)
t0 = time.time()
total = near = 0
for ci in range(n_cases):
    nseq, dim, per = int(rng.integers(1, 4)), int(rng.choice([50, 800])), int(rng.integers(1, 4))
    descs, poses, seq = [], [], []
    for s in range(nseq):
        n = int(rng.integers(60, 900))
        laps, radius = float(rng.uniform(1.5, 4.0)), float(rng.uniform(15, 45))
        t = np.linspace(0, 2 * np.pi * laps, n)
        pos = np.stack([radius * np.cos(t), radius * np.sin(t), 0.2 * rng.normal(0, 1, n)], 1) + rng.normal(0, 0.4, (n, 3)) + 200.0 * s
        P = np.tile(np.eye(4), (n, 1, 1))
        P[:, :3, 3] = pos
        base = rng.random((1, dim)) ** 3
        d = (base + 0.3 * rng.random((n, dim)) ** 3 + 0.02 * np.abs(np.sin(t))[:, None]).astype(np.float32)
        descs.append(d / d.sum(1, keepdims=True)); poses.append(P); seq.append(np.full(n, s))
    desc, P, seq = np.concatenate(descs), np.concatenate(poses), np.concatenate(seq)
    np.random.seed(ci)
    trip = TripletMiner().mine_triplets(desc, P, per, seq if nseq > 1 or ci % 2 else None)
    got = {}
    for a, p, n_ in trip:
        got.setdefault(int(a), []).append((int(p), int(n_)))
    cand = {}
    for s in range(nseq):
        idx = np.where(seq == s)[0]
        res = mo.mine_sequence(desc[idx], P[idx][:, :3, 3])
        for la, r in enumerate(res):
            if r is not None:
                cand[int(idx[la])] = (set(idx[r[0]].tolist()), idx[r[1]], int(idx[r[2]]))
    assert set(got) == set(cand), f"case {ci}: anchors differ ({len(got)} vs {len(cand)})"
    for a, pairs in got.items():
        assert len(pairs) == per, (ci, a)
        for p, n_ in pairs:
            assert p in cand[a][0], f"case {ci}: positive of anchor {a} outside the candidate set"
            if n_ != cand[a][2]:                                 # a float32 near-tie of two candidates' W1 distances
                dw = abs(mo.w1_numpy(desc[a], desc[n_]) - mo.w1_numpy(desc[a], desc[cand[a][2]]))
                assert n_ in set(cand[a][1].tolist()) and dw <= 2e-4, f"case {ci}: negative of anchor {a}: {n_} vs {cand[a][2]} (dW1 {dw:.2e})"
                near += 1
            total += 1
    if ci % 10 == 9:
        print(f"{ci + 1} cases, {total} triplets, {near} near-ties ({time.time() - t0:.0f} s)", flush=True)
print(f"TOTAL {n_cases} cases, {total} triplets: same anchors, every positive in the candidate set, hard negative = the oracle's argmin "
      f"({near} float32 near-ties within 2e-4)")
