#!/usr/bin/env python3
"""Timings of the keyframe-side rows: quantizer, record packing, chain graph, voxel IoU."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_spectral_codec_amd import synth                                    # noqa: E402
from neural_spectral_codec_amd.data import pose_utils as pu                    # noqa: E402
from neural_spectral_codec_amd.encoding import quantization as qz              # noqa: E402
from neural_spectral_codec_amd.keyframe.graph_manager import chain_graph_device, chain_edges, edge_features  # noqa: E402


def bench(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


h = torch.rand((1024, 800), device="cuda") ** 4
h = h / h.sum(1, keepdim=True)
q = qz.quantize_batch(h)
print(f"quantize 1024x800   : {bench(lambda: qz.quantize_batch(h)):8.1f} us")
print(f"dequantize 1024x800 : {bench(lambda: qz.dequantize_batch(q)):8.1f} us")
p7 = torch.rand((1024, 7), device="cuda")
ts = torch.rand(1024, device="cuda", dtype=torch.float64)
ids = torch.arange(1024, device="cuda")
hs = torch.randint(0, 256, (1024, 20), device="cuda", dtype=torch.uint8)
print(f"pack 1024 records   : {bench(lambda: qz.pack_records(q, p7, ts, ids, hs)):8.1f} us")

n = 4541
poses = synth.make_pose_chain(n, 0)
pd = torch.from_numpy(poses).cuda()
print(f"chain graph N=4541  : {bench(lambda: chain_graph_device(n, 5, 'cuda', pd)):8.1f} us (device poses)")
t = time.perf_counter()
for _ in range(5):
    e = chain_edges(n, 5)
    edge_features(poses, e)
print(f"  host numpy builder: {(time.perf_counter() - t) / 5 * 1e6:8.1f} us")

rng = np.random.default_rng(0)
P = 64
c1 = [torch.from_numpy((rng.uniform(-8, 8, (5000, 3)) * [1, 1, 0.1]).astype(np.float32)).cuda() for _ in range(P)]
c2 = [torch.from_numpy((rng.uniform(-8, 8, (5000, 3)) * [1, 1, 0.1]).astype(np.float32)).cuda() for _ in range(P)]
T = np.tile(np.eye(4), (P, 1, 1))
print(f"voxel IoU 1 pair    : {bench(lambda: pu.compute_overlap_batch(c1[:1], c2[:1], T[:1]), 20):8.1f} us (incl. Python packing)")
print(f"voxel IoU 64 pairs  : {bench(lambda: pu.compute_overlap_batch(c1, c2, T), 20):8.1f} us (incl. Python packing)")
