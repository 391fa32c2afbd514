"""16-bit descriptor wire format on MI355X -- mirror of the reference's src/encoding/quantization.py.

``HistogramQuantizer`` (:113-191) and ``CompressedDescriptor`` (:22-110) keep their names and meaning;
the arithmetic runs in nsc_quantize_descriptors / nsc_dequantize_descriptors (bit-exact, including
numpy's pairwise float32 summation order) and the records are laid out by nsc_pack_records /
nsc_unpack_records.  The reference still assumes 50-bin descriptors here (quantization.py:117,141,
SURVEY.md section 9 quirk 10); this mirror takes any ``n_bins`` (800 on the path), so a record is
``2 * n_bins + 120`` bytes -- 220 for 50 bins, 1 720 for 800.

Additive: ``quantize_batch`` / ``dequantize_batch`` / ``pack_records`` / ``unpack_records`` work on whole
(n, n_bins) device tensors.  There is no CPU fallback.
"""
import hashlib
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

from .. import _lib


def _device(device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None:
        raise _lib.NscError("the quantizer runs on a HIP device; none is available (no CPU fallback)")
    return torch.device(device)


def quantize_batch(histograms: torch.Tensor, epsilon: float = 1e-8) -> torch.Tensor:
    """(n, n_bins) float32 device tensor -> (n, n_bins) uint16      quantization.py:131-168 per row"""
    _lib.require_cuda(histograms, "histograms")
    h = histograms.detach().to(torch.float32).contiguous()
    out = torch.empty(h.shape, dtype=torch.uint16, device=h.device)
    with torch.cuda.device(h.device):
        st = _lib.lib().nsc_quantize_descriptors(_lib.ptr(h), int(h.shape[0]), int(h.shape[1]), float(epsilon),
                                                 _lib.ptr(out), _lib.stream_ptr(h.device))
    _lib.check(st, "nsc_quantize_descriptors")
    return out


def dequantize_batch(quantized: torch.Tensor, epsilon: float = 1e-8) -> torch.Tensor:
    """(n, n_bins) uint16 device tensor -> (n, n_bins) float32      quantization.py:170-191 per row"""
    _lib.require_cuda(quantized, "quantized")
    q = quantized.contiguous()
    if q.dtype != torch.uint16:
        raise _lib.NscError(f"quantized must be uint16 (got {q.dtype})")
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        st = _lib.lib().nsc_dequantize_descriptors(_lib.ptr(q), int(q.shape[0]), int(q.shape[1]), float(epsilon),
                                                   _lib.ptr(out), _lib.stream_ptr(q.device))
    _lib.check(st, "nsc_dequantize_descriptors")
    return out


def record_bytes(n_bins: int) -> int:
    return 2 * int(n_bins) + 120


def pack_records(quantized: torch.Tensor, poses7: torch.Tensor, timestamps: torch.Tensor,
                 keyframe_ids: torch.Tensor, hashes: torch.Tensor) -> torch.Tensor:
    """Device tensors (n,n_bins) uint16, (n,7) float32, (n,) float64, (n,) uint32/int64, (n,20) uint8
    -> (n, 2*n_bins+120) uint8 records in the layout of CompressedDescriptor.to_bytes (:41-72)."""
    _lib.require_cuda(quantized, "quantized")
    dev = quantized.device
    n, dim = int(quantized.shape[0]), int(quantized.shape[1])
    q = quantized.contiguous()
    p7 = poses7.to(device=dev, dtype=torch.float32).contiguous()
    ts = timestamps.to(device=dev, dtype=torch.float64).contiguous()
    ids = keyframe_ids.to(device=dev)
    if ids.dtype != torch.uint32:                       # ids above 2^31 arrive as int64: keep the low 32 bits
        ids = ids.to(torch.int64).to(torch.int32).view(torch.uint32)
    ids = ids.contiguous()
    hs = hashes.to(device=dev, dtype=torch.uint8).contiguous()
    if p7.shape != (n, 7) or ts.shape != (n,) or ids.shape != (n,) or hs.shape != (n, 20):
        raise _lib.NscError("pack_records: metadata shapes must be (n,7), (n,), (n,), (n,20)")
    rec = torch.empty((n, record_bytes(dim)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().nsc_pack_records(_lib.ptr(q), _lib.ptr(p7), _lib.ptr(ts), _lib.ptr(ids), _lib.ptr(hs),
                                         n, dim, _lib.ptr(rec), _lib.stream_ptr(dev))
    _lib.check(st, "nsc_pack_records")
    return rec


def unpack_records(records: torch.Tensor, n_bins: int):
    """(n, 2*n_bins+120) uint8 device tensor -> (quantized, poses7, timestamps, keyframe_ids, hashes)."""
    _lib.require_cuda(records, "records")
    dev = records.device
    rec = records.contiguous()
    n = int(rec.shape[0])
    if rec.dtype != torch.uint8 or int(rec.shape[1]) != record_bytes(n_bins):
        raise _lib.NscError(f"records must be (n, {record_bytes(n_bins)}) uint8")
    q = torch.empty((n, n_bins), dtype=torch.uint16, device=dev)
    p7 = torch.empty((n, 7), dtype=torch.float32, device=dev)
    ts = torch.empty((n,), dtype=torch.float64, device=dev)
    ids = torch.empty((n,), dtype=torch.uint32, device=dev)
    hs = torch.empty((n, 20), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().nsc_unpack_records(_lib.ptr(rec), n, int(n_bins), _lib.ptr(q), _lib.ptr(p7), _lib.ptr(ts),
                                           _lib.ptr(ids), _lib.ptr(hs), _lib.stream_ptr(dev))
    _lib.check(st, "nsc_unpack_records")
    return q, p7, ts, ids, hs


@dataclass
class CompressedDescriptor:
    """One keyframe record (quantization.py:22-39): uint16 histogram, 7-DOF pose [x,y,z,qw,qx,qy,qz],
    timestamp, keyframe id and the 20-byte SHA-1 of the cloud."""
    histogram: np.ndarray
    pose: np.ndarray
    timestamp: float
    keyframe_id: int
    point_cloud_hash: bytes

    def to_bytes(self) -> bytes:
        """2*n_bins + 120 bytes (220 for the reference's 50 bins)          quantization.py:41-72"""
        dev = _device()
        rec = pack_records(
            torch.from_numpy(np.ascontiguousarray(self.histogram, dtype=np.uint16)).to(dev).reshape(1, -1),
            torch.from_numpy(np.asarray(self.pose, dtype=np.float32)).reshape(1, 7),
            torch.tensor([self.timestamp], dtype=torch.float64),
            torch.tensor([int(self.keyframe_id)], dtype=torch.int64),
            torch.frombuffer(bytearray(self.point_cloud_hash), dtype=torch.uint8).reshape(1, 20))
        return rec[0].cpu().numpy().tobytes()

    @staticmethod
    def from_bytes(data: bytes) -> 'CompressedDescriptor':
        """quantization.py:74-110; n_bins follows from the record length."""
        if len(data) < 122 or (len(data) - 120) % 2:
            raise ValueError(f"not a descriptor record: {len(data)} bytes")
        n_bins = (len(data) - 120) // 2
        rec = torch.frombuffer(bytearray(data), dtype=torch.uint8).reshape(1, -1).to(_device())
        q, p7, ts, ids, hs = unpack_records(rec, n_bins)
        return CompressedDescriptor(histogram=q[0].cpu().numpy(), pose=p7[0].cpu().numpy(),
                                    timestamp=float(ts.cpu().numpy()[0]), keyframe_id=int(ids.cpu().numpy()[0]),
                                    point_cloud_hash=hs[0].cpu().numpy().tobytes())


class HistogramQuantizer:
    """quantization.py:113-191.  ``quantize`` / ``dequantize`` keep the reference's per-histogram numpy
    signatures (host array in, host array out, computed on the device)."""

    def __init__(self, n_bins: int = 50, epsilon: float = 1e-8, device=None):
        self.n_bins = n_bins
        self.epsilon = epsilon
        self.max_value = 65535
        self.device = device

    def quantize(self, histogram: np.ndarray) -> np.ndarray:
        assert len(histogram) == self.n_bins, f"Expected {self.n_bins} bins, got {len(histogram)}"
        h = torch.from_numpy(np.ascontiguousarray(histogram, dtype=np.float32)).to(_device(self.device))
        return quantize_batch(h.reshape(1, -1), self.epsilon)[0].cpu().numpy()

    def dequantize(self, quantized: np.ndarray) -> np.ndarray:
        assert len(quantized) == self.n_bins, f"Expected {self.n_bins} bins, got {len(quantized)}"
        q = torch.from_numpy(np.ascontiguousarray(quantized, dtype=np.uint16)).to(_device(self.device))
        return dequantize_batch(q.reshape(1, -1), self.epsilon)[0].cpu().numpy()


def compute_point_cloud_hash(points: np.ndarray) -> bytes:
    """SHA-1 of the float32 xyz bytes (quantization.py:194-212).  Host: a hash chain does not parallelise."""
    return hashlib.sha1(np.ascontiguousarray(points[:, :3], dtype=np.float32).tobytes()).digest()


def pose_to_7dof(pose: np.ndarray) -> np.ndarray:
    """(4,4) SE(3) -> [x, y, z, qw, qx, qy, qz]                     quantization.py:215-246"""
    from scipy.spatial.transform import Rotation
    x, y, z, w = Rotation.from_matrix(pose[:3, :3]).as_quat()
    return np.concatenate([pose[:3, 3], [w, x, y, z]])


def pose_from_7dof(pose_7dof: np.ndarray) -> np.ndarray:
    """[x, y, z, qw, qx, qy, qz] -> (4,4) SE(3)                     quantization.py:249-284"""
    from scipy.spatial.transform import Rotation
    w, x, y, z = pose_7dof[3:]
    pose = np.eye(4)
    pose[:3, :3] = Rotation.from_quat([x, y, z, w]).as_matrix()
    pose[:3, 3] = pose_7dof[:3]
    return pose


def compress_descriptor(histogram: np.ndarray, pose: np.ndarray, timestamp: float, keyframe_id: int,
                        points: np.ndarray) -> CompressedDescriptor:
    """quantization.py:287-329 (n_bins follows the histogram instead of the stale default of 50)."""
    q = HistogramQuantizer(n_bins=len(histogram)).quantize(histogram)
    return CompressedDescriptor(histogram=q, pose=pose_to_7dof(pose), timestamp=timestamp,
                                keyframe_id=keyframe_id, point_cloud_hash=compute_point_cloud_hash(points))


def decompress_descriptor(descriptor: CompressedDescriptor) -> Tuple[np.ndarray, np.ndarray, float, int]:
    """quantization.py:332-358"""
    h = HistogramQuantizer(n_bins=len(descriptor.histogram)).dequantize(descriptor.histogram)
    return h, pose_from_7dof(descriptor.pose), descriptor.timestamp, descriptor.keyframe_id
