#!/usr/bin/env python3
"""The GAT half of the path as a stand-alone workload for rocprofv3 (BASELINE configs[2]): 3-layer GAT forward,
800 -> 256 -> GATx3 -> 800, edge_dim = 2, eval mode, over a KITTI-00-shaped temporal chain (N keyframes, 5 temporal
neighbours).  usage: gat_workload.py [N=4541] [reps=50] [kernel set: 0 default (banded layers), 1 co-resident, 2 generic stand-alone]
Prints the average forward time measured with HIP events (un-profiled runs) -- under the profiler use the trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import gat_oracle as go
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cores = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.manual_seed(0)
m = create_spectral_gnn(edge_dim=2)
go.randomize_bn_stats(m)
m = m.to("cuda").eval()
m.gnn.coresident = {0: False, 1: True, 2: "generic"}[cores]
g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
with torch.no_grad():
    for _ in range(10):
        out = m(g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = m(g)
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
flop = 2.0 * n * (800 * 256 + 3 * 256 * 256 + 256 * 800)
print(f"N={n} kernel set {cores}: {us:.1f} us per forward = {flop / us / 1e6:.1f} TFLOP/s of f32 MFMA work "
      f"({flop / us / 1e6 / 157.3 * 100:.1f} % of 157.3 TF)", flush=True)
