"""Per-call latency of the reference-style API: encode_points(one 120 k cloud) and gnn(graph)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import gat_oracle as go
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm
enc = SpectralEncoder(n_elevation=16).to("cuda")
cloud = synth.make_cloud(0, 120000)
dev_cloud = torch.from_numpy(cloud).cuda()
def t(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print(f"encode_points(numpy (120000,4)) incl. H2D + D2H like pipeline.py:245 : {t(lambda: enc.encode_points(cloud).detach().cpu().numpy()):8.1f} us")
print(f"encode_points(device tensor), async                                  : {t(lambda: enc.encode_points(dev_cloud)):8.1f} us")
off = torch.tensor([0, 120000], dtype=torch.int64, device='cuda')
out = torch.empty((1, 800), device='cuda')
print(f"encode_points_batch((pts, offsets), out=...) one cloud                : {t(lambda: enc.encode_points_batch((dev_cloud, off), out=out)):8.1f} us")
m = create_spectral_gnn(edge_dim=None); go.randomize_bn_stats(m); m = m.to('cuda').eval()
for n in (100, 1000):
    g = gm.Data(x=torch.rand(n, 800, device='cuda'), edge_index=torch.from_numpy(gm.chain_edges(n).T.copy()).cuda(), num_nodes=n)
    with torch.no_grad():
        print(f"gnn(graph) {n} nodes (online window, pipeline.py:253-256)              : {t(lambda: m(g)):8.1f} us")
