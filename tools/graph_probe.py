"""hipGraph probe: the GNN forward (8 launches, launch-bound at 1 024 nodes) captured with torch.cuda.CUDAGraph
(the C ABI only enqueues kernels on the given stream, so a capturing stream records them) vs issued eagerly;
and the training step (forward_train + triplet loss + backward) the same way, dropout 0."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.gnn.trainer import TripletLoss
from neural_spectral_codec_amd.keyframe import graph_manager as gm

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = create_spectral_gnn(edge_dim=2, dropout=0.0)
synth.randomize_bn_stats(model)
model = model.to(dev).eval()


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    t_host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, t_host


for n in (1024, 4541):
    g = gm.synthetic_chain_graph(n, device=dev, seed=1)
    with torch.no_grad():
        ref = model(g)
        e_dev, e_host = timeit(lambda: model(g))
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3):
                model(g)
        torch.cuda.current_stream(dev).wait_stream(s)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            out = model(g)
        cg.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref), "graph replay differs from the eager forward"
        g_dev, g_host = timeit(cg.replay)
    print(f"GNN forward N={n}: eager {e_dev:.1f} us device / {e_host:.1f} us host issue; "
          f"hipGraph replay {g_dev:.1f} us device / {g_host:.1f} us host issue", flush=True)

# training step, dropout 0 (a seed baked into a captured launch would repeat the mask)
for n in (1024, 4541):
    g = gm.synthetic_chain_graph(n, device=dev, seed=1)
    model.train()
    crit = TripletLoss(0.1)
    rng = np.random.default_rng(0)
    trip = torch.from_numpy(rng.integers(0, n, (1024, 3))).to(dev)
    ia, ip, in_ = trip[:, 0].contiguous(), trip[:, 1].contiguous(), trip[:, 2].contiguous()
    params = [p for p in model.parameters()]

    def step():
        emb = model(g)
        loss = crit.forward_indexed(emb, ia, ip, in_, scale=0.25)
        loss.backward()
        return loss

    for p in params:
        p.grad = None
    step()
    for p in params:
        p.grad = torch.zeros_like(p)
    e_dev, e_host = timeit(step, 50)
    try:
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(s)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            loss = step()
        cg.replay()
        torch.cuda.synchronize()
        g_dev, g_host = timeit(cg.replay, 50)
        print(f"train step N={n} (forward + loss + backward, grads accumulated): eager {e_dev:.1f} us device / {e_host:.1f} us host; "
              f"hipGraph replay {g_dev:.1f} us device / {g_host:.1f} us host; loss {float(loss):.5f}", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"train step N={n}: eager {e_dev:.1f} us device / {e_host:.1f} us host; capture failed: {type(ex).__name__}: {ex}", flush=True)
    model.eval()
