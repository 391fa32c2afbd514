"""Multi-GPU layout of the descriptor path: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).

The reference is single-process; this is the MI355X-side design:

* keyframes shard CONTIGUOUSLY over ranks -- every cloud is independent, so the encoder needs no
  communication;
* ONE exchange step: an all-gather of the (N_r, 800) float32 descriptor shards (3 200 B per keyframe),
  after which every rank holds the full descriptor matrix (what stage-1 retrieval consumes,
  reference src/retrieval/two_stage_retrieval.py:145-202);
* the GNN runs on each rank's own node range plus a halo: the temporal graph is a chain with
  |i - j| <= M//2 (src/keyframe/graph_manager.py:520-532), so L GAT layers see at most L*(M//2)
  nodes beyond the shard (6 for the reference's L = 3, M = 5).  With eval-mode BatchNorm (pointwise)
  the owned rows are exactly the rows of the full-graph forward.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .keyframe.graph_manager import Data, chain_edges, edge_features


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank: the first n_total % world ranks get one extra keyframe."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_descriptors(local: torch.Tensor, n_total: Optional[int] = None,
                           group=None) -> torch.Tensor:
    """All-gather (N_r, D) shards laid out by shard_range() into the full (N, D) matrix.

    Equal shards use a single all_gather_into_tensor (one RCCL collective, no copies); ragged
    shards are padded to the largest shard and trimmed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n_local, d = int(local.shape[0]), int(local.shape[1])
    if n_total is None:
        n_total = n_local * world
    base, rem = divmod(n_total, world)
    local = local.contiguous()
    if rem == 0:
        out = torch.empty((n_total, d), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    cap = base + 1
    padded = torch.zeros((cap, d), dtype=local.dtype, device=local.device)
    padded[:n_local] = local
    buf = torch.empty((world * cap, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(buf[r * cap:r * cap + (hi - lo)])
    return torch.cat(parts, 0)


def halo_window(n_total: int, lo: int, hi: int, n_layers: int = 3, temporal_neighbors: int = 5
                ) -> Tuple[int, int]:
    """Node window [wlo, whi) a rank needs so rows [lo, hi) of an n_layers-deep GAT are exact."""
    halo = n_layers * (temporal_neighbors // 2)
    return max(0, lo - halo), min(n_total, hi + halo)


def shard_graph(desc_all: torch.Tensor, lo: int, hi: int, poses=None, n_layers: int = 3,
                temporal_neighbors: int = 5):
    """Sub-chain graph over the shard + halo window (node ids relative to the window start).
    Returns (Data, first owned row inside the window)."""
    n_total = int(desc_all.shape[0])
    wlo, whi = halo_window(n_total, lo, hi, n_layers, temporal_neighbors)
    n = whi - wlo
    edges = chain_edges(n, temporal_neighbors)
    dev = desc_all.device
    edge_index = (torch.from_numpy(edges.T.copy()).to(dev) if len(edges)
                  else torch.empty((2, 0), dtype=torch.long, device=dev))
    edge_attr = None
    if poses is not None and len(edges):
        edge_attr = torch.from_numpy(edge_features(poses[wlo:whi], edges)).to(dev)
    g = Data(x=desc_all[wlo:whi], edge_index=edge_index, edge_attr=edge_attr, num_nodes=n)
    return g, lo - wlo


class ShardedDescriptorPath:
    """encode (local shard) -> all-gather descriptors -> GNN on shard + halo (local rows out).

    ``encoder`` needs ``encode_points_batch(clouds)``; ``gnn`` is called as ``gnn(data)``.  The graph
    of a fixed (n_total, poses) layout is built once and reused: only ``x`` changes per step."""

    def __init__(self, encoder, gnn, n_total: int, poses=None, temporal_neighbors: int = 5,
                 n_layers: int = 3, group=None):
        self.encoder, self.gnn, self.group = encoder, gnn, group
        self.n_total, self.poses = n_total, poses
        self.M, self.L = temporal_neighbors, n_layers
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.lo, self.hi = shard_range(n_total, self.rank, self.world)
        self._graph = None
        self._own0 = 0

    def step(self, clouds):
        """clouds: this rank's shard (list of arrays or (points, offsets) device tensors).
        Returns (all descriptors (n_total, D), enhanced embeddings of the owned rows (hi-lo, D))."""
        local = self.encoder.encode_points_batch(clouds)
        desc_all = all_gather_descriptors(local, self.n_total, self.group)
        if self._graph is None:
            self._graph, self._own0 = shard_graph(desc_all, self.lo, self.hi, self.poses, self.L, self.M)
            self._wlo = self.lo - self._own0
        else:
            self._graph.x = desc_all[self._wlo:self._wlo + self._graph.num_nodes]
        emb = self.gnn(self._graph)
        return desc_all, emb[self._own0:self._own0 + (self.hi - self.lo)]


def all_reduce_gradients(params, group=None, average: bool = False):
    """Sum (or average) the .grad of ``params`` over the ranks with ONE collective: gradients are
    flattened into a single bucket (the GNN has 613 920 float32 parameters = 2.46 MB, far below the
    size where several buckets would overlap anything), all-reduced, and copied back."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n


def split_triplets(triplets, rank: int, world: int):
    """Contiguous slice of a triplet batch for this rank + the weight local/total that makes the sum of
    the per-rank mean losses equal the reference's global mean (trainer.py:68 loss.mean())."""
    n = len(triplets)
    lo, hi = shard_range(n, rank, world)
    return triplets[lo:hi], ((hi - lo) / n if n else 0.0)
