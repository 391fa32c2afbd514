"""Hard-negative triplet mining on MI355X -- mirror of reference src/gnn/triplet_miner.py ``TripletMiner``
(:27-359), per-sequence path (``mine_triplets`` with ``sequence_ids``, :88-117 -> ``_mine_sequence_triplets``
:141-229 -> ``_select_hard_negative`` :314-359).

The reference walks the anchors in Python, asks a cKDTree three radius questions per anchor and calls
``wasserstein_distance_1d_numpy`` once per negative candidate.  Here one kernel (nsc_mine_triplets, one
wavefront per anchor) evaluates the same inclusive-radius / temporal-gap predicates by brute force with
float64 distances and picks argmin W1 over the candidates from pre-normalised CDF rows ("hard"; "semi-hard" takes the
candidate at position len // 2 of the W1 order, :352-357, "random" a uniform one).
The positive is a uniform random candidate (``np.random.choice`` in the reference, unseeded): chosen
here by a counter-based hash seeded from numpy's global RNG, so ``np.random.seed`` still controls it.
"""
import ctypes as C
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from ..retrieval.wasserstein import _cdf


class TripletMiner:
    def __init__(self, positive_distance_max: float = 5.0, positive_temporal_min: int = 30,
                 negative_distance_min: float = 10.0, negative_distance_max: float = 50.0,
                 negative_temporal_min: int = 30, mining_strategy: str = "hard", device: str = "cuda"):
        self.positive_distance_max = positive_distance_max
        self.positive_temporal_min = positive_temporal_min
        self.negative_distance_min = negative_distance_min
        self.negative_distance_max = negative_distance_max
        self.negative_temporal_min = negative_temporal_min
        self.mining_strategy = mining_strategy
        self.device = torch.device(device)
        if mining_strategy not in ("hard", "random", "semi-hard"):
            raise ValueError(f"Unknown mining strategy: {mining_strategy}")                  # :359

    def _params(self, per_anchor: int) -> _lib.MineParams:
        p = _lib.MineParams()
        p.positive_distance_max = float(self.positive_distance_max)
        p.negative_distance_min = float(self.negative_distance_min)
        p.negative_distance_max = float(self.negative_distance_max)
        p.positive_temporal_min = int(self.positive_temporal_min)
        p.negative_temporal_min = int(self.negative_temporal_min)
        p.strategy = {"hard": 0, "random": 1, "semi-hard": 2}[self.mining_strategy]
        p.triplets_per_anchor = int(per_anchor)
        p.seed = int(np.random.randint(0, 2 ** 62, dtype=np.int64))
        return p

    def mine_sequence(self, seq_indices: np.ndarray, descriptors: torch.Tensor, positions: torch.Tensor,
                      n_triplets_per_anchor: int = 1):
        """One sequence: returns (triplets (T,3) int64 global indices, counts (n,2))."""
        dev = self.device
        idx = torch.as_tensor(np.asarray(seq_indices), dtype=torch.int64, device=dev)
        n = int(idx.numel())
        desc = descriptors.index_select(0, idx).contiguous()
        pos = positions.index_select(0, idx).contiguous()
        cdf = _cdf(desc, 1e-8, True)                       # wasserstein.py:37-47: h / sum, cumsum
        out_pos = torch.empty((n, n_triplets_per_anchor), dtype=torch.int32, device=dev)
        out_neg = torch.empty_like(out_pos)
        counts = torch.empty((n, 2), dtype=torch.int32, device=dev)
        p = self._params(n_triplets_per_anchor)
        L = _lib.lib()
        nbytes = L.nsc_mine_workspace_bytes(n, p.strategy)       # semi-hard: one row of candidate distances per anchor
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
        with torch.cuda.device(dev):
            st = L.nsc_mine_triplets_ws(_lib.ptr(pos), _lib.ptr(cdf), n, int(desc.shape[1]), C.byref(p),
                                        _lib.ptr(out_pos), _lib.ptr(out_neg), _lib.ptr(counts), _lib.ptr(ws), nbytes,
                                        _lib.stream_ptr(dev))
        _lib.check(st, "nsc_mine_triplets_ws")
        ok = (out_pos >= 0) & (out_neg >= 0)
        a_loc = torch.arange(n, device=dev).unsqueeze(1).expand_as(out_pos)[ok]
        trip = torch.stack([idx[a_loc], idx[out_pos[ok].long()], idx[out_neg[ok].long()]], 1)
        return trip, counts

    def mine_triplets(self, descriptors: np.ndarray, poses: np.ndarray, n_triplets_per_anchor: int = 1,
                      sequence_ids: Optional[np.ndarray] = None) -> List[Tuple[int, int, int]]:
        """triplet_miner.py:66-139.  Without ``sequence_ids`` the whole set is mined as one sequence
        (the reference's O(n^2) branch applies the same predicates on global indices)."""
        dev = self.device
        desc = torch.as_tensor(np.asarray(descriptors), dtype=torch.float32).to(dev)
        pos = torch.as_tensor(np.asarray(poses)[:, :3, 3].astype(np.float64)).to(dev)   # :161-163
        n = int(desc.shape[0])
        if sequence_ids is None:
            groups = [np.arange(n)]
        else:
            sequence_ids = np.asarray(sequence_ids)
            groups = [np.where(sequence_ids == s)[0] for s in np.unique(sequence_ids)]          # :91-97
        out = []
        for g in groups:
            if len(g) < 3:                                                                      # :99-100
                continue
            trip, _ = self.mine_sequence(g, desc, pos, n_triplets_per_anchor)
            out.append(trip)
        if not out:
            return []
        allt = torch.cat(out, 0).cpu().numpy()
        return [tuple(int(v) for v in row) for row in allt]


def create_triplet_miner(positive_distance_max: float = 5.0, positive_temporal_min: int = 30,
                         negative_distance_min: float = 10.0, negative_distance_max: float = 50.0,
                         negative_temporal_min: int = 30, mining_strategy: str = "hard",
                         device: str = "cuda") -> TripletMiner:
    """triplet_miner.py:512-541"""
    return TripletMiner(positive_distance_max=positive_distance_max, positive_temporal_min=positive_temporal_min,
                        negative_distance_min=negative_distance_min, negative_distance_max=negative_distance_max,
                        negative_temporal_min=negative_temporal_min, mining_strategy=mining_strategy, device=device)
