"""How long does a fresh box take to reach its steady step time?  The pipelined step (two encoder streams) in blocks
of 50 steps, per-block time printed; run as the first GPU program on a box and again right after."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from neural_spectral_codec_amd import distributed as nd
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn

t_start = time.perf_counter()
dev = torch.device("cuda", 0)
enc = SpectralEncoder(n_elevation=16).to(dev)
torch.manual_seed(0)
model = create_spectral_gnn(edge_dim=2)
synth.randomize_bn_stats(model)
model = model.to(dev).eval()
n = 1024
pts, off = synth.make_clouds_device(n, 120000, dev, seed=1234)
poses = synth.make_pose_chain(n, 0)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 30
path = nd.ShardedDescriptorPath(enc, model, n, poses, pipeline=True, encoder_streams=streams)
torch.cuda.synchronize()
print(f"setup {time.perf_counter() - t_start:.2f} s", flush=True)
out = []
with torch.no_grad():
    for b in range(blocks):
        t0 = time.perf_counter()
        for _ in range(50):
            path.step((pts, off), inputs_ready=True)
        path.synchronize()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 50 * 1e3)
print("ms/step per block of 50:", " ".join(f"{v:.4f}" for v in out), flush=True)
