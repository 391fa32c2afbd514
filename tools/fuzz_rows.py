#!/usr/bin/env python3
"""One-off soak (not part of the test suite) of the rows either side of the path: the 16-bit wire format, the voxel-IoU
novelty test, the validation recall and the chain-graph builder on RANDOM shapes against their CPU restatements
(oracle/keyframe_oracle.py, oracle/recall_oracle.py) -- bit-exact for uint16 / counts / indices / ranks.
usage: fuzz_rows.py [n_cases]"""
import os
import sys
import time

import numpy as np
import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "oracle"))
import keyframe_oracle as ko                                                    # noqa: E402
import recall_oracle as ro                                                      # noqa: E402
from neural_spectral_codec_amd.data import pose_utils as pu                    # noqa: E402
from neural_spectral_codec_amd.encoding import quantization as qz              # noqa: E402
from neural_spectral_codec_amd.gnn.trainer import GNNTrainer                   # noqa: E402
from neural_spectral_codec_amd.keyframe.graph_manager import build_chain_graph  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rand_pose(rng, scale):
    a = rng.normal(0, 1, 3)
    a /= np.linalg.norm(a) + 1e-12
    th = rng.uniform(-0.6, 0.6)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T[:3, 3] = rng.normal(0, scale, 3)
    return T


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(11)
t0 = time.time()
for ci in range(n_cases):
    # --- wire format: widths 1-4096, any row count, normalised / raw / empty rows
    nb, n = int(rng.integers(1, 4097)) if ci % 3 == 0 else int(rng.choice([50, 800, 181, 16])), int(rng.integers(1, 300))
    h = (rng.random((n, nb)) ** 4).astype(np.float32)
    h[: n // 2] /= np.maximum(h[: n // 2].sum(1, keepdims=True), 1e-30)
    if n > 2:
        h[-1] = 0.0
    q = qz.quantize_batch(torch.from_numpy(h).cuda()).cpu().numpy()
    want = np.stack([ko.quantize(r) for r in h])
    assert (q == want).all(), f"case {ci}: quantize ({n} x {nb})"
    arb = rng.integers(0, 65536, (n, nb)).astype(np.uint16)
    d = qz.dequantize_batch(torch.from_numpy(arb).cuda()).cpu().numpy()
    assert (bits(d) == bits(np.stack([ko.dequantize(r) for r in arb]))).all(), f"case {ci}: dequantize ({n} x {nb})"
    # --- voxel IoU: cloud sizes 0-5 000 (the reference down-samples above that, with an unseeded RNG), 3 or 4 columns, random rigid transform, voxel 0.1-1.0, NaN rows now and then
    cols, vox = int(rng.choice([3, 4])), float(rng.choice([0.1, 0.2, 0.25, 0.5, 1.0]))
    pairs = int(rng.integers(1, 6))
    p1 = [rng.uniform(-8, 8, (int(rng.integers(0, 5001)), cols)).astype(np.float32) * np.array([1, 1, 0.1, 1][:cols], np.float32) for _ in range(pairs)]
    p2 = [rng.uniform(-8, 8, (int(rng.integers(0, 5001)), cols)).astype(np.float32) * np.array([1, 1, 0.1, 1][:cols], np.float32) for _ in range(pairs)]
    for a in p1 + p2:
        if len(a) > 10 and rng.random() < 0.3:
            a[rng.integers(0, len(a), 3), rng.integers(0, 3)] = np.nan
    Ts = np.stack([rand_pose(rng, 2.0) for _ in range(pairs)])
    iou, counts = pu.compute_overlap_batch(p1, p2, Ts, vox, return_counts=True)
    for i in range(pairs):
        wi, wc = ko.voxel_overlap(p1[i], p2[i], Ts[i], vox)
        assert float(iou[i]) == wi and counts[i].cpu().numpy().tolist() == wc.tolist(), f"case {ci}: voxel IoU pair {i} ({len(p1[i])} / {len(p2[i])} points, voxel {vox})"
    # --- chain graph: 1-400 nodes, 1-9 temporal neighbours, loop closures (some out of range), with / without poses
    n, m = int(rng.integers(1, 400)), int(rng.integers(1, 10))
    poses = np.stack([rand_pose(rng, 5.0) for _ in range(n)])
    loops = [(int(a), int(b)) for a, b in rng.integers(-2, n + 2, (int(rng.integers(0, 8)), 2))]
    ei, ea = ko.chain_graph_loop(n, m, poses, loops)
    g = build_chain_graph(torch.zeros((n, 8)), m, "cuda", poses, loops)
    assert tuple(g.edge_index.shape) == (2, ei.shape[1]) and (g.edge_index.cpu().numpy() == ei).all(), f"case {ci}: chain graph ({n}, {m})"
    if ea is not None:
        got = g.edge_attr.cpu().numpy()
        ulp = np.abs(bits(got).astype(np.int64) - bits(ea).astype(np.int64))
        assert ((np.abs(got - ea) <= 1e-7) | (ulp <= 2)).all(), f"case {ci}: edge features"
    # --- validation recall: 40-900 poses on a looping track, embedding widths 8-800
    n, dim = int(rng.integers(40, 900)), int(rng.choice([8, 32, 100, 800]))
    t = np.linspace(0, 2 * np.pi * 3, n)
    pos = np.stack([30 * np.cos(t), 30 * np.sin(t), 0.1 * rng.normal(0, 1, n)], 1) + rng.normal(0, 0.3, (n, 3))
    poses = np.tile(np.eye(4), (n, 1, 1))
    poses[:, :3, 3] = pos
    emb = np.concatenate([pos * 0.05 + rng.normal(0, 0.08, (n, 3)), rng.normal(0, 0.05, (n, dim - 3))], 1).astype(np.float32)
    tr = GNNTrainer.__new__(GNNTrainer)
    tr.device = "cuda"
    et = torch.from_numpy(emb).cuda()
    for k in (1, 5, 10):
        got, ref = tr._compute_recall_loop_closure(et, poses, k, 5.0, 30), ro.recall_loop_closure(emb, poses, k, 5.0, 30)
        assert got[1] == ref[1] and abs(got[0] - ref[0]) < 1e-12, f"case {ci}: recall@{k} ({n} poses, dim {dim}): {got} vs {ref}"
    if ci % 10 == 9:
        print(f"{ci + 1} cases ({time.time() - t0:.0f} s)", flush=True)
print(f"TOTAL {n_cases} cases: wire format (widths 1-4 096) and voxel-IoU counts bit-exact, chain-graph indices exact and edge features within "
      f"2 ulp, recall@1/5/10 equal to the restatement")
