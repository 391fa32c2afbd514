"""Development probe: GAT forward timing in isolation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import gat_oracle as go
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm
for n in (1024, 4541):
    torch.manual_seed(0)
    m = create_spectral_gnn(edge_dim=2); go.randomize_bn_stats(m); m = m.to("cuda").eval()
    g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
    with torch.no_grad():
        for _ in range(5): m(g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): m(g)
        e1.record(); torch.cuda.synchronize()
    print(f"N={n}: {e0.elapsed_time(e1)/100*1e3:.1f} us per forward", flush=True)
sys.exit(0)
for n in (1024, 4541):
    torch.manual_seed(0)
    m = create_spectral_gnn(edge_dim=2); go.randomize_bn_stats(m); m = m.to("cuda").eval()
    g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
    with torch.no_grad():
        for _ in range(5): m(g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): m(g)
        e1.record(); torch.cuda.synchronize()
    print(f"unfused N={n}: {e0.elapsed_time(e1)/100*1e3:.1f} us per forward", flush=True)
