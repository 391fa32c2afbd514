#!/usr/bin/env python3
"""What would a split-K = 2 form of input_proj cost?  The same tile count and per-tile K as the split halves, through the
shipped kernel: input_dim 400 (6.25 chunks of 64) on 9 082 nodes = 1 136 tiles of 32 x 64, against the shipped shape
(input_dim 800 on 4 541 nodes = 568 tiles of 12.5 chunks).  Run under rocprofv3 --kernel-trace --stats and read the
duration of gemm_nt_kernel<*, 1, *> (the input projection).  usage: gemm_shape_probe.py IN_DIM N"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm

in_dim, n = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
m = create_spectral_gnn(input_dim=in_dim, output_dim=800, edge_dim=2)
synth.randomize_bn_stats(m)
m = m.to("cuda").eval()
g = gm.synthetic_chain_graph(n, device="cuda", seed=1, features=torch.rand(n, in_dim))
with torch.no_grad():
    for _ in range(60):
        m(g)
torch.cuda.synchronize()
print("done", in_dim, n)
