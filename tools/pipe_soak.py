#!/usr/bin/env python3
"""Soak of the pipelined step (two overlapping encoder streams, GNN replayed as a captured hipGraph, four rotating
buffers): thousands of steps over a rotation of different batches, EVERY step's gathered matrix and embedding compared on
the device with the one-stream path's result for that batch (bit for bit), reading the result only after the path's own
event -- a buffer handed out too early, a stale capture or a missing dependency shows up as a mismatch.
usage: pipe_soak.py [steps=6000] [clouds=1024] [points=20000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neural_spectral_codec_amd import distributed as nd, synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
dev = torch.device("cuda", 0)
enc = SpectralEncoder(n_elevation=16).to(dev)
torch.manual_seed(0)
model = create_spectral_gnn(edge_dim=2)
synth.randomize_bn_stats(model)
model = model.to(dev).eval()
poses = synth.make_pose_chain(n, 0)
NB = 5                                             # batches in rotation: coprime to the 4 buffers and the 2 encoder streams
batches = [synth.make_clouds_device(n, npts + 64 * k, dev, seed=100 + k) for k in range(NB)]
serial = nd.ShardedDescriptorPath(enc, model, n, poses)
with torch.no_grad():
    model.gnn.coresident = False
    want = [tuple(t.clone() for t in serial.step(b)) for b in batches]
piped = nd.ShardedDescriptorPath(enc, model, n, poses, pipeline=True)
bad = torch.zeros(2, dtype=torch.int64, device=dev)
t0 = time.time()
with torch.no_grad():
    for k in range(steps):
        d, e = piped.step(batches[k % NB], inputs_ready=True)
        torch.cuda.current_stream(dev).wait_event(piped.last_event)       # the caller's stream reads the result
        wd, we = want[k % NB]
        bad[0] += (d != wd).any().long()
        bad[1] += (e != we).any().long()
    piped.synchronize()
    torch.cuda.synchronize()
b = bad.cpu().tolist()
print(f"{steps} pipelined steps ({n} clouds x ~{npts} points, {NB} batches in rotation, encoder streams {piped.encoder_streams}, "
      f"gnn_graph {piped.gnn_graph}, captures {len(piped._gnn_graphs)}): steps with a differing gathered matrix {b[0]}, "
      f"with a differing embedding {b[1]}  ({time.time() - t0:.1f} s)")
sys.exit(1 if (b[0] or b[1]) else 0)
