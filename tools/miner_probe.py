"""Mining time on a KITTI-00-sized sequence (4 541 keyframes, loops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
n = 4541
rng = np.random.default_rng(0)
t = np.linspace(0, 6 * np.pi, n)
xy = 150.0 * np.stack([np.cos(t), np.sin(2 * t) * 0.6], 1) + rng.normal(0, 0.4, (n, 2))
poses = np.tile(np.eye(4), (n, 1, 1)); poses[:, 0, 3], poses[:, 1, 3] = xy[:, 0], xy[:, 1]
desc = (rng.random((n, 800)) ** 4).astype(np.float32); desc /= desc.sum(1, keepdims=True)
seq = np.zeros(n, int)
if len(sys.argv) > 1 and sys.argv[1] == "ref":
    sys.path.insert(0, "/root/reference/src")
    from gnn.triplet_miner import TripletMiner
    t0 = time.perf_counter(); tr = TripletMiner().mine_triplets(desc, poses, 1, seq); dt = time.perf_counter() - t0
    print(f"reference (CPU, 1 core): {len(tr)} triplets in {dt:.2f} s")
else:
    import torch
    from neural_spectral_codec_amd.gnn.triplet_miner import TripletMiner
    m = TripletMiner()
    m.mine_triplets(desc, poses, 1, seq); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr = m.mine_triplets(desc, poses, 1, seq); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"MI355X: {len(tr)} triplets in {dt*1e3:.1f} ms (incl. H2D of descriptors and the Python list build)")
