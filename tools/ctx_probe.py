"""Does the encoder kernel run slower when interleaved with the GAT kernels? (development probe)"""
import sys, os, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import gat_oracle as go
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm
n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
out = torch.empty((n, 800), device="cuda")
m = create_spectral_gnn(edge_dim=2); go.randomize_bn_stats(m); m = m.to("cuda").eval()
g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
g.x = out
def run(mode, reps=20):
    evs = []
    with torch.no_grad():
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); enc.encode_points_batch((pts, off), out=out); e1.record()
            if mode == "gat": m(g)
            elif mode == "small": torch.zeros(16, device="cuda").add_(1)
            elif mode == "sleep": torch.cuda._sleep(200000)
            evs.append((e0, e1))
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in evs) * 1e3
for mode in ("none", "gat", "small", "sleep", "none", "gat"):
    run(mode, 3)
    print(mode, f"{run(mode):.1f} us (median encoder kernel)", flush=True)
