"""cProfile of the host side of the pipelined step (what the 0.13 ms of issue time per step is spent on)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neural_spectral_codec_amd import distributed as nd, synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn

dev = torch.device("cuda", 0)
enc = SpectralEncoder(n_elevation=16).to(dev)
torch.manual_seed(0)
model = create_spectral_gnn(edge_dim=2)
synth.randomize_bn_stats(model)
model = model.to(dev).eval()
n = 1024
pts, off = synth.make_clouds_device(n, 120000, dev, seed=1)
path = nd.ShardedDescriptorPath(enc, model, n, synth.make_pose_chain(n, 0), pipeline=True)
with torch.no_grad():
    for _ in range(50):
        path.step((pts, off), inputs_ready=True)
    path.synchronize(); torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        path.step((pts, off), inputs_ready=True)
    pr.disable()
    path.synchronize(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
