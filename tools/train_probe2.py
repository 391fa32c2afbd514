"""Where the captured training step's time goes: replays alone, replays + index/seed updates, whole train_batches."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.gnn.trainer import GNNTrainer
from neural_spectral_codec_amd.keyframe import graph_manager as gm

def T(fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for n in (1024, 4541):
    torch.manual_seed(0)
    m = create_spectral_gnn(edge_dim=2, dropout=0.1)
    g = gm.synthetic_chain_graph(n, device="cuda", seed=1)
    tr = GNNTrainer(m, device="cuda", batch_size=1024, accumulation_steps=4)
    rng = np.random.default_rng(0)
    trip = rng.integers(0, n, (4096, 3))
    tr.train_batches(g, trip); tr.train_batches(g, trip)
    assert tr._captured and not tr._capture_failed
    cg, idx, seed, loss, grads = next(iter(tr._captured.values()))[:5]
    dev_trip = torch.from_numpy(np.ascontiguousarray(trip.T)).cuda()
    print(f"N={n}: replay only {T(cg.replay, 40):.3f} ms; ", end="")
    def upd():
        for j in range(3): idx[j].copy_(dev_trip[j, :1024], non_blocking=True)
        seed.fill_(12345); cg.replay()
    print(f"+ index/seed updates {T(upd, 40):.3f} ms; ", end="")
    print(f"_captured_step {T(lambda: tr._captured_step(g, trip[:1024], 0.25, dev_trip[:, :1024]), 40):.3f} ms; ", end="")
    print(f"optimizer.step+zero_grad {T(lambda: (tr.optimizer.step(), tr.optimizer.zero_grad(set_to_none=False)), 20):.3f} ms; ", end="")
    print(f"train_batches/4 {T(lambda: tr.train_batches(g, trip), 5) / 4:.3f} ms; ", end="")
    tr2 = GNNTrainer(create_spectral_gnn(edge_dim=2, dropout=0.1), device="cuda", batch_size=1024, accumulation_steps=4, use_graph=False)
    tr2.train_batches(g, trip)
    print(f"eager train_batches/4 {T(lambda: tr2.train_batches(g, trip), 5) / 4:.3f} ms", flush=True)
