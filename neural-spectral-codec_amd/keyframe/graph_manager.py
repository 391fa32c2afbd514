"""Graph container + offline temporal-graph builder feeding the GNN.

Mirrors reference src/keyframe/graph_manager.py:471-606 (``build_graph_from_keyframes_batch``):
chain graph over the flat keyframe list, offsets +-1..+-(M//2), edge_attr = [log1p(d)/5, theta/pi].
The reference's O(N*M) Python loop is vectorised here (same arithmetic, float64 -> float32 at the
same place).  ``Data`` stands in for ``torch_geometric.data.Data`` (reference :19,599) when PyG is
not installed; a real PyG ``Data`` is accepted everywhere as well.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

try:  # pragma: no cover - PyG is optional
    from torch_geometric.data import Data as _PygData
except Exception:  # noqa: BLE001
    _PygData = None


class _Data:
    """Minimal duck-typed stand-in for torch_geometric.data.Data (x, edge_index, edge_attr, num_nodes)."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, num_nodes=None, **kw):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self._num_nodes = num_nodes
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        if self._num_nodes is not None:
            return self._num_nodes
        return None if self.x is None else int(self.x.shape[0])

    @property
    def num_edges(self):
        return 0 if self.edge_index is None else int(self.edge_index.shape[1])

    def to(self, device):
        for k, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self

    def __repr__(self):
        def sh(t):
            return None if t is None else list(t.shape)
        return f"Data(x={sh(self.x)}, edge_index={sh(self.edge_index)}, edge_attr={sh(self.edge_attr)})"


Data = _PygData if _PygData is not None else _Data


def chain_edges(n_nodes: int, temporal_neighbors: int = 5) -> np.ndarray:
    """Edges [i, i+off] for off in -M//2..M//2, off != 0, in the reference's order (node-major,
    ascending offset), as an (E,2) int64 array.   graph_manager.py:520-532"""
    half = temporal_neighbors // 2
    offs = np.array([o for o in range(-half, half + 1) if o != 0], dtype=np.int64)
    i = np.repeat(np.arange(n_nodes, dtype=np.int64), len(offs))
    j = i + np.tile(offs, n_nodes)
    ok = (j >= 0) & (j < n_nodes)
    return np.stack([i[ok], j[ok]], 1)


def edge_features(poses: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """[log1p(||t_i - t_j||)/5, arccos(clip((clip(tr(R_j R_i^T),-1,3)-1)/2,-1,1))/pi] per edge, float32.
    graph_manager.py:535-550,583-596"""
    i, j = edges[:, 0], edges[:, 1]
    d = np.linalg.norm(poses[i, :3, 3] - poses[j, :3, 3], axis=1)
    rel = np.einsum("eab,ecb->eac", poses[j, :3, :3], poses[i, :3, :3])       # R_j @ R_i.T
    tr = np.clip(np.trace(rel, axis1=1, axis2=2), -1.0, 3.0)
    ang = np.arccos(np.clip((tr - 1.0) / 2.0, -1.0, 1.0))
    d32 = d.astype(np.float32)
    a32 = ang.astype(np.float32)
    return np.stack([np.log1p(d32) / 5.0, a32 / np.pi], axis=1).astype(np.float32)


def chain_graph_device(n_nodes: int, temporal_neighbors: int, device, poses=None, loop_closures=None):
    """edge_index (2,E) int64 and edge_attr (E,2) float32 (or None) built on the device by
    nsc_build_chain_graph: same edge order and arithmetic as graph_manager.py:520-596."""
    from .. import _lib
    device = torch.device(device)
    loops = None
    if loop_closures:                                                     # range check as graph_manager.py:555
        ok = [[int(q), int(m)] for q, m in loop_closures if 0 <= q < n_nodes and 0 <= m < n_nodes]
        if ok:
            loops = torch.tensor(ok, dtype=torch.int64, device=device)
    n_loops = 0 if loops is None else int(loops.shape[0])
    L = _lib.lib()
    E = int(L.nsc_chain_graph_num_edges(n_nodes, temporal_neighbors, n_loops))
    edge_index = torch.empty((2, E), dtype=torch.int64, device=device)
    pd = None
    edge_attr = None
    if poses is not None and E > 0:
        pd = torch.as_tensor(np.ascontiguousarray(np.asarray(poses, dtype=np.float64))
                             if not isinstance(poses, torch.Tensor) else poses)
        pd = pd.to(device=device, dtype=torch.float64).reshape(n_nodes, 16).contiguous()
        edge_attr = torch.empty((E, 2), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        st = L.nsc_build_chain_graph(_lib.ptr(pd), n_nodes, temporal_neighbors, _lib.ptr(loops), n_loops,
                                     _lib.ptr(edge_index), _lib.ptr(edge_attr), _lib.stream_ptr(device))
    _lib.check(st, "nsc_build_chain_graph")
    return edge_index, edge_attr


def build_chain_graph(features: torch.Tensor, temporal_neighbors: int = 5, device="cpu",
                      poses: Optional[np.ndarray] = None,
                      loop_closures: Optional[List[Tuple[int, int]]] = None):
    """Same graph as build_graph_from_keyframes_batch, from a stacked (N,D) feature tensor.  On a HIP
    device the edges and edge features are produced by nsc_build_chain_graph; for device='cpu' (the
    reference's default, graph moved later with .to()) the vectorised host builder is used."""
    n = int(features.shape[0])
    if torch.device(device).type == "cuda":
        edge_index, edge_attr = chain_graph_device(n, temporal_neighbors, device, poses, loop_closures)
        return Data(x=features.to(device), edge_index=edge_index, edge_attr=edge_attr, num_nodes=n)
    edges = chain_edges(n, temporal_neighbors)
    if loop_closures:                                                     # graph_manager.py:553-558
        extra = [[q, m] for q, m in loop_closures if 0 <= q < n and 0 <= m < n]
        if extra:
            ex = np.asarray(extra, dtype=np.int64)
            edges = np.concatenate([edges, np.stack([ex, ex[:, ::-1]], 1).reshape(-1, 2)], 0)
    if len(edges):
        edge_index = torch.from_numpy(np.ascontiguousarray(edges.T)).to(device)
    else:
        edge_index = torch.empty((2, 0), dtype=torch.long, device=device)
    edge_attr = None
    if poses is not None and len(edges):
        edge_attr = torch.from_numpy(edge_features(np.asarray(poses), edges)).to(device)
    g = Data(x=features.to(device), edge_index=edge_index, edge_attr=edge_attr, num_nodes=n)
    return g.to(device)


def build_graph_from_keyframes_batch(keyframes, temporal_neighbors: int = 5, device: str = 'cpu',
                                     poses: np.ndarray = None,
                                     loop_closures: List[Tuple[int, int]] = None):
    """graph_manager.py:471-606.  ``keyframes``: objects with a ``.descriptor`` ndarray."""
    if len(keyframes) == 0:
        return None
    feats = torch.stack([torch.from_numpy(np.asarray(kf.descriptor)).float() for kf in keyframes], 0)
    return build_chain_graph(feats, temporal_neighbors, device, poses, loop_closures)


def build_graph_from_keyframes(keyframes, temporal_neighbors: int = 5, device: str = 'cpu'):
    """graph_manager.py:443-470: the graph TemporalGraphManager builds when every keyframe is active -- the same
    chain edges in the same order (:130-165), no edge_attr."""
    return build_graph_from_keyframes_batch(keyframes, temporal_neighbors, device, poses=None)


def synthetic_chain_graph(n_nodes: int, device="cpu", seed: int = 0, temporal_neighbors: int = 5,
                          features: Optional[torch.Tensor] = None):
    """KITTI-00-shaped synthetic graph: positive rows summing to 1
    (or given descriptors), random-walk poses, chain edges with edge_attr."""
    from .. import synth
    if features is None:
        g = torch.Generator().manual_seed(seed)
        features = torch.rand((n_nodes, 800), generator=g) ** 4
        features = features / features.sum(1, keepdim=True)
    return build_chain_graph(features, temporal_neighbors, device, synth.make_pose_chain(n_nodes, seed))
