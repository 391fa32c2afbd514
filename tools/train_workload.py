"""The captured training step as a stand-alone workload for rocprofv3 (BASELINE configs[4] shape): N keyframes (default
4 541), 1 024-triplet batches, 4 batches per optimizer step.  usage: train_workload.py [N=4541] [optimizer steps=12]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.gnn.trainer import GNNTrainer
from neural_spectral_codec_amd.keyframe import graph_manager as gm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(0)
m = create_spectral_gnn(edge_dim=2, dropout=0.1)
g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
tr = GNNTrainer(m, device="cuda", batch_size=1024, accumulation_steps=4)
trip = np.random.default_rng(0).integers(0, n, (4096, 3))
for _ in range(steps):
    tr.train_batches(g, trip)
torch.cuda.synchronize()
print("captured:", bool(tr._captured), "failed:", tr._capture_failed)
