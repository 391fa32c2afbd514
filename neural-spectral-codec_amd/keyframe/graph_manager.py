"""Graph container, offline temporal-graph builder and the online sliding-window manager feeding the GNN.

Mirrors reference src/keyframe/graph_manager.py:471-606 (``build_graph_from_keyframes_batch``):
chain graph over the flat keyframe list, offsets +-1..+-(M//2), edge_attr = [log1p(d)/5, theta/pi].
The reference's O(N*M) Python loop is vectorised here (same arithmetic, float64 -> float32 at the
same place).  ``Data`` stands in for ``torch_geometric.data.Data`` (reference :19,599) when PyG is
not installed; a real PyG ``Data`` is accepted everywhere as well.
"""
from typing import List, Optional, Set, Tuple

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


class TemporalGraphManager:
    """Online sliding-window graph of the inference loop -- same methods and behaviour as the reference's
    TemporalGraphManager (graph_manager.py:24-441, driven by pipeline.py:250-256), different bookkeeping:

    * descriptors live in a preallocated device buffer; ``add_keyframe`` copies one 800-float row instead of
      re-stacking every active descriptor (:114-118), the window slides by moving the start of the view;
    * the temporal edges of an n-node window depend on n only and are cached per n (built by
      nsc_build_chain_graph on a HIP device);
    * as in the reference, every ``add_keyframe`` rebuilds the graph from the temporal edges alone, so an edge
      added by ``add_loop_closure_edge`` lives until the next keyframe arrives (:100, :104-128), and the online
      graph carries no ``edge_attr`` (:124-128)."""

    def __init__(self, temporal_neighbors: int = 5, max_active_nodes: int = 1000, feature_dim: int = 800,
                 device: str = 'cpu'):
        self.temporal_neighbors = temporal_neighbors
        self.max_active_nodes = max_active_nodes
        self.feature_dim = feature_dim
        self.device = device
        self._edge_cache = {}
        self.reset()

    def reset(self):
        self.graph = None
        self.keyframes = []
        self.frozen_keyframes = []
        self.frozen_embeddings = None
        self.keyframe_id_to_node_idx = {}
        self._buf = None
        self._start = 0

    # -- internals ----------------------------------------------------------------------------
    def _edges(self, n: int) -> torch.Tensor:
        e = self._edge_cache.get(n)
        if e is None:
            if torch.device(self.device).type == "cuda":
                e, _ = chain_graph_device(n, self.temporal_neighbors, self.device)
            else:
                ce = chain_edges(n, self.temporal_neighbors)
                e = (torch.from_numpy(np.ascontiguousarray(ce.T)) if len(ce)
                     else torch.zeros((2, 0), dtype=torch.long)).to(self.device)
            if len(self._edge_cache) > 2 * self.max_active_nodes + 8:
                self._edge_cache.clear()
            self._edge_cache[n] = e
        return e

    def _append_row(self, descriptor: np.ndarray):
        row = torch.from_numpy(np.ascontiguousarray(descriptor, dtype=np.float32))
        d = int(row.numel())
        n = len(self.keyframes)                       # the new keyframe is already in the list
        if self._buf is None:
            cap = 2 * max(self.max_active_nodes, 1) + 2
            self._buf = torch.empty((cap, d), dtype=torch.float32, device=self.device)
            self._start = 0
        if self._start + n > self._buf.shape[0]:      # slide the window back to the front of the buffer
            self._buf[:n - 1] = self._buf[self._start:self._start + n - 1].clone()
            self._start = 0
        self._buf[self._start + n - 1] = row.to(self.device)

    def _rebuild_graph(self):
        """:104-128 -- temporal edges only, x = the active rows (a view of the buffer)."""
        n = len(self.keyframes)
        if n == 0:
            self.graph = None
            return
        self.graph = Data(x=self._buf[self._start:self._start + n], edge_index=self._edges(n), num_nodes=n)

    def _freeze_oldest_node(self):
        """:166-202"""
        if not self.keyframes:
            return
        oldest = self.keyframes.pop(0)
        self.frozen_keyframes.append(oldest)
        del self.keyframe_id_to_node_idx[oldest.keyframe_id]
        for k in self.keyframe_id_to_node_idx:
            self.keyframe_id_to_node_idx[k] -= 1
        if getattr(oldest, "embedding", None) is not None:
            emb = torch.from_numpy(np.asarray(oldest.embedding)).float().to(self.device).unsqueeze(0)
            self.frozen_embeddings = emb if self.frozen_embeddings is None else torch.cat([self.frozen_embeddings, emb], 0)
        self._start += 1
        self._rebuild_graph()

    # -- reference API ------------------------------------------------------------------------
    def add_keyframe(self, keyframe) -> int:
        """:75-102 -> node index in the active graph"""
        if keyframe.descriptor is None:
            raise ValueError("Keyframe must have descriptor computed before adding to graph")
        self.keyframes.append(keyframe)
        node_idx = len(self.keyframes) - 1
        self.keyframe_id_to_node_idx[keyframe.keyframe_id] = node_idx
        self._append_row(keyframe.descriptor)
        self._rebuild_graph()
        if len(self.keyframes) > self.max_active_nodes:
            self._freeze_oldest_node()
        return node_idx

    def get_graph(self):
        return self.graph

    def add_loop_closure_edge(self, query_keyframe_id: int, match_keyframe_id: int, pose_query: np.ndarray = None,
                              pose_match: np.ndarray = None) -> bool:
        """:208-272 -- bidirectional edge between two ACTIVE keyframes (edge features only if the graph has any)."""
        qi = self.keyframe_id_to_node_idx.get(query_keyframe_id)
        mi = self.keyframe_id_to_node_idx.get(match_keyframe_id)
        if qi is None or mi is None or self.graph is None:
            return False
        new = torch.tensor([[qi, mi], [mi, qi]], dtype=torch.long, device=self.device)
        self.graph.edge_index = torch.cat([self.graph.edge_index, new], 1)
        if pose_query is not None and pose_match is not None and getattr(self.graph, "edge_attr", None) is not None:
            f = edge_features(np.stack([pose_query, pose_match]), np.array([[0, 1], [0, 1]]))
            self.graph.edge_attr = torch.cat([self.graph.edge_attr, torch.from_numpy(f).to(self.device)], 0)
        return True

    def get_node_index(self, keyframe_id: int) -> Optional[int]:
        return self.keyframe_id_to_node_idx.get(keyframe_id, None)

    def get_k_hop_neighbors(self, node_idx: int, k: int) -> Set[int]:
        """:286-320 -- breadth-first over outgoing edges"""
        if self.graph is None or k <= 0:
            return {node_idx}
        ei = self.graph.edge_index.cpu().numpy()
        seen, layer = {node_idx}, {node_idx}
        for _ in range(k):
            nxt = set(ei[1, np.isin(ei[0], list(layer))].tolist()) if layer else set()
            seen |= nxt
            layer = nxt
            if not layer:
                break
        return seen

    def get_local_subgraph(self, node_idx: int, k_hops: int = 3):
        """:322-375 -> (subgraph, {original index: subgraph index})"""
        if self.graph is None:
            raise ValueError("Graph is empty")
        nodes = sorted(self.get_k_hop_neighbors(node_idx, k_hops))
        mapping = {old: new for new, old in enumerate(nodes)}
        ei = self.graph.edge_index.cpu().numpy()
        keep = np.isin(ei[0], nodes) & np.isin(ei[1], nodes)
        lut = np.full(int(self.graph.num_nodes), -1, dtype=np.int64)
        lut[nodes] = np.arange(len(nodes))
        sub = lut[ei[:, keep]]
        sub_ei = (torch.from_numpy(np.ascontiguousarray(sub)) if sub.size
                  else torch.zeros((2, 0), dtype=torch.long)).to(self.device)
        idx = torch.as_tensor(nodes, dtype=torch.long, device=self.graph.x.device)
        return Data(x=self.graph.x[idx], edge_index=sub_ei, num_nodes=len(nodes)), mapping

    def update_embeddings(self, embeddings: torch.Tensor):
        """:377-393"""
        if len(embeddings) != len(self.keyframes):
            raise ValueError(f"Embedding count ({len(embeddings)}) != keyframe count ({len(self.keyframes)})")
        emb = embeddings.detach().cpu().numpy()
        for i, kf in enumerate(self.keyframes):
            kf.embedding = emb[i]

    def get_all_keyframes(self) -> list:
        return self.frozen_keyframes + self.keyframes

    def get_all_descriptors(self) -> np.ndarray:
        return np.array([kf.descriptor for kf in self.get_all_keyframes()])

    def get_all_embeddings(self) -> Optional[np.ndarray]:
        kfs = self.get_all_keyframes()
        if getattr(kfs[0], "embedding", None) is None:
            return None
        return np.array([kf.embedding for kf in kfs])

    def get_statistics(self) -> dict:
        ne = int(self.graph.edge_index.shape[1]) if self.graph is not None else 0
        return {'num_active_nodes': len(self.keyframes), 'num_frozen_nodes': len(self.frozen_keyframes),
                'total_nodes': len(self.keyframes) + len(self.frozen_keyframes), 'num_edges': ne,
                'avg_degree': ne / len(self.keyframes) if self.graph is not None and self.keyframes else 0.0}


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
