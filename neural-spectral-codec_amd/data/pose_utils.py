"""Voxel-IoU overlap on MI355X -- mirror of ``compute_overlap`` (reference src/data/pose_utils.py:323-389),
the geometric-novelty test of the keyframe selector (src/keyframe/criteria.py:98-136).

The voxelisation, set union and intersection run in nsc_voxel_overlap (one workgroup per cloud pair, an
LDS hash set); results are bit-identical to the reference on the same sampled points.  As in the
reference, clouds above ``max_points`` are randomly down-sampled first (:340-347; the reference draws from
numpy's unseeded global RNG, so no stream is matched -- host arrays use numpy's RNG, device tensors
torch's).  Additive: ``compute_overlap_batch`` scores many pairs in one launch.  No CPU fallback.
"""
from typing import Sequence

import numpy as np
import torch

from .. import _lib

MAX_PAIR_POINTS = 12288          # capacity of the kernel's hash set (reference: 2 x max_points = 10 000)


def _sample(points, max_points):
    if len(points) <= max_points:
        return points
    if isinstance(points, torch.Tensor):
        return points[torch.randperm(len(points), device=points.device)[:max_points]]
    return points[np.random.choice(len(points), max_points, replace=False)]          # pose_utils.py:341-347


def _as_dev(points, device):
    t = torch.from_numpy(np.ascontiguousarray(points)) if isinstance(points, np.ndarray) else points
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def compute_overlap_batch(points1: Sequence, points2: Sequence, transforms, voxel_size: float = 0.2,
                          max_points: int = 5000, device=None, return_counts: bool = False):
    """IoU of voxelised cloud pairs: pair k = (points1[k] moved by transforms[k], points2[k]).
    Returns a float64 tensor (n_pairs,) on the device (and the (n_pairs,3) counts
    [|voxels1|, |voxels2|, |intersection|] if asked)."""
    if len(points1) != len(points2):
        raise _lib.NscError("compute_overlap_batch: points1 and points2 must pair up")
    if device is None:
        first = next((p for p in list(points1) + list(points2) if isinstance(p, torch.Tensor) and p.is_cuda), None)
        device = first.device if first is not None else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.NscError("compute_overlap runs on a HIP device (no CPU fallback)")
    P = len(points1)
    c1 = [_as_dev(_sample(p, max_points), device) for p in points1]
    c2 = [_as_dev(_sample(p, max_points), device) for p in points2]
    cols = {int(c.shape[1]) for c in c1 + c2 if c.ndim == 2}
    if len(cols) > 1 or (cols and next(iter(cols)) not in (3, 4)) or any(c.ndim != 2 for c in c1 + c2):
        raise _lib.NscError("compute_overlap: clouds must all be (N,3) or all (N,4)")
    stride = next(iter(cols)) if cols else 3
    n1 = [len(c) for c in c1]
    n2 = [len(c) for c in c2]
    max_pair = max([a + b for a, b in zip(n1, n2)], default=0)
    if max_pair > MAX_PAIR_POINTS:
        raise _lib.NscError(f"compute_overlap: {max_pair} points in one pair exceeds {MAX_PAIR_POINTS}; "
                            "lower max_points")
    off1 = torch.tensor(np.concatenate([[0], np.cumsum(n1)]), dtype=torch.int64, device=device)
    off2 = torch.tensor(np.concatenate([[0], np.cumsum(n2)]), dtype=torch.int64, device=device)
    p1 = torch.cat(c1, 0) if c1 else torch.empty((0, stride), device=device)
    p2 = torch.cat(c2, 0) if c2 else torch.empty((0, stride), device=device)
    T = torch.as_tensor(np.asarray(transforms, dtype=np.float64).reshape(P, 16)).to(device).contiguous()
    counts = torch.empty((P, 3), dtype=torch.int32, device=device)
    iou = torch.empty((P,), dtype=torch.float64, device=device)
    L = _lib.lib()
    t1, t2 = int(sum(n1)), int(sum(n2))
    nbytes = L.nsc_voxel_overlap_workspace_bytes(t1, t2)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        st = L.nsc_voxel_overlap(_lib.ptr(p1), _lib.ptr(off1), _lib.ptr(p2), _lib.ptr(off2), P, t1, t2, max_pair,
                                 stride, _lib.ptr(T), float(voxel_size), _lib.ptr(counts), _lib.ptr(iou),
                                 _lib.ptr(ws), nbytes, _lib.stream_ptr(device))
    _lib.check(st, "nsc_voxel_overlap")
    return (iou, counts) if return_counts else iou


def compute_overlap(points1, points2, T_12, voxel_size: float = 0.2, max_points: int = 5000) -> float:
    """IoU ratio of two clouds; same signature and meaning as pose_utils.py:323-389."""
    return float(compute_overlap_batch([points1], [points2], np.asarray(T_12)[None], voxel_size, max_points)[0].item())
