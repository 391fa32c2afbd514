"""Timing of the training step (BASELINE configs[4] shape): N keyframes, 1024 triplets per batch."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import gat_oracle as go
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.gnn.trainer import GNNTrainer
from neural_spectral_codec_amd.keyframe import graph_manager as gm
for n in (1024, 4541):
    torch.manual_seed(0)
    m = create_spectral_gnn(edge_dim=2, dropout=0.1)
    g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
    tr = GNNTrainer(m, device="cuda", batch_size=1024, accumulation_steps=4)
    rng = np.random.default_rng(0)
    trip = rng.integers(0, n, (4096, 3))
    tr.train_batches(g, trip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        tr.train_batches(g, trip)           # 4 batches of 1024 triplets + 1 Adam step
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n}: {dt*1e3:.2f} ms per optimizer step (4 x [forward+loss+backward] + Adam) = {dt/4*1e3:.2f} ms per 1024-triplet batch", flush=True)
