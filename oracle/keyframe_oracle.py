"""CPU restatement (numpy + the C oracle) of the keyframe-side rows either side of the hot path
(SURVEY.md 8f rank 4): the 16-bit descriptor wire format, the chain-graph edge features and the
voxel-IoU novelty test.

TEST INFRASTRUCTURE ONLY: imported by tests/ and oracle/gen_golden_keyframe.py -- never by the
product package.

Parity status:
* quantize / dequantize / record layout: PINNED against the reference's own
  encoding.quantization.{HistogramQuantizer, CompressedDescriptor} run in this container
  (tests/golden/keyframe.npz, generator oracle/gen_golden_keyframe.py).
* voxel overlap: PINNED against data.pose_utils.compute_overlap (same fixture file) on clouds at or
  below its max_points (above it the reference samples with the unseeded global RNG).
* edge features: restated from keyframe/graph_manager.py:520-596; that module imports torch_geometric
  and cannot run here, so this piece is "parity unpinned" (it is pinned only against a literal
  per-edge Python loop of the same lines in tests/test_keyframe_rows.py).
"""
import ctypes as C
import struct

import numpy as np

import nsc_oracle

MAX_U16 = 65535


# ---------------------------------------------------------------------------------------------
# numpy's float32 add.reduce order (what every `.sum()` below executes): pairwise summation,
# numpy/_core/src/umath/loops_utils.h.src  *_pairwise_sum, valid for contiguous n <= 8192.
# Restated so that the HIP kernel has an order to follow; tests check it against ndarray.sum().
# ---------------------------------------------------------------------------------------------
def pairwise_sum_f32(a):
    f = np.float32
    n = len(a)
    if n < 8:
        r = f(0.0)
        for x in a:
            r = f(r + x)
        return r
    if n <= 128:
        r = [f(a[k]) for k in range(8)]
        m = n - (n % 8)
        for i in range(8, m, 8):
            for k in range(8):
                r[k] = f(r[k] + a[i + k])
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        for i in range(m, n):
            res = f(res + a[i])
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(pairwise_sum_f32(a[:n2]) + pairwise_sum_f32(a[n2:]))


# ---------------------------------------------------------------------------------------------
# HistogramQuantizer (encoding/quantization.py:113-191), any n_bins
# ---------------------------------------------------------------------------------------------
def quantize(histogram, epsilon=1e-8):
    """quantization.py:131-168 for one (n_bins,) float32 histogram -> uint16."""
    h = np.asarray(histogram, dtype=np.float32)
    s = h.sum()                                                   # :144 float32 pairwise
    if s > np.float32(epsilon):
        h = h / (s + np.float32(epsilon))                         # :146 (numpy >= 2: float32 + weak python float)
    q = np.round(h * np.float32(MAX_U16)).astype(np.uint16)       # :150
    qs = int(q.sum())                                             # :154 (uint64 accumulate: exact)
    if qs > 0:
        err = MAX_U16 - qs                                        # :157
        if err != 0:
            i = int(q.argmax())                                   # :161 first maximum
            q[i] = np.uint16(min(max(int(q[i]) + err, 0), MAX_U16))   # :162-166
    return q


def dequantize(quantized, epsilon=1e-8):
    """quantization.py:170-191."""
    h = np.asarray(quantized).astype(np.float32)
    s = h.sum()
    if s > np.float32(epsilon):
        return h / (s + np.float32(epsilon))
    return np.ones(len(h), dtype=np.float32) / np.float32(len(h))


def record_bytes(n_bins):
    """2 bytes per bin + 120 bytes of metadata (quantization.py:26-33; 220 for the reference's 50 bins)."""
    return 2 * n_bins + 120


def pack_record(q, pose7, timestamp, keyframe_id, pc_hash):
    """CompressedDescriptor.to_bytes (quantization.py:41-72) for any n_bins."""
    out = (np.asarray(q).astype(np.uint16).tobytes() + np.asarray(pose7).astype(np.float32).tobytes()
           + struct.pack('d', timestamp) + struct.pack('I', keyframe_id) + bytes(pc_hash) + bytes(60))
    assert len(out) == record_bytes(len(q))
    return out


def unpack_record(data, n_bins):
    """CompressedDescriptor.from_bytes (quantization.py:74-110)."""
    b = 2 * n_bins
    q = np.frombuffer(data[:b], dtype=np.uint16)
    pose = np.frombuffer(data[b:b + 28], dtype=np.float32)
    ts = struct.unpack('d', data[b + 28:b + 36])[0]
    kid = struct.unpack('I', data[b + 36:b + 40])[0]
    return q, pose, ts, kid, data[b + 40:b + 60]


# ---------------------------------------------------------------------------------------------
# chain graph + edge features (keyframe/graph_manager.py:520-596), literal per-edge loop
# ---------------------------------------------------------------------------------------------
def chain_graph_loop(n_nodes, temporal_neighbors, poses=None, loop_closures=None):
    edges, dist, rot = [], [], []
    half = temporal_neighbors // 2

    def feat(i, j):
        d = np.linalg.norm(poses[i, :3, 3] - poses[j, :3, 3])                      # :537-539
        r_rel = poses[j, :3, :3] @ poses[i, :3, :3].T                               # :546
        tr = np.clip(np.trace(r_rel), -1.0, 3.0)                                    # :548
        dist.append(d)
        rot.append(np.arccos(np.clip((tr - 1.0) / 2.0, -1.0, 1.0)))                # :549

    for i in range(n_nodes):
        for off in range(-half, half + 1):
            j = i + off
            if off == 0 or not (0 <= j < n_nodes):
                continue
            edges.append([i, j])
            if poses is not None:
                feat(i, j)
    for q, m in (loop_closures or []):                                              # :553-572
        if 0 <= q < n_nodes and 0 <= m < n_nodes:
            edges += [[q, m], [m, q]]
            if poses is not None:
                feat(q, m)
                feat(q, m)                                                          # both directions reuse (q, m)
    ei = np.asarray(edges, dtype=np.int64).reshape(-1, 2).T
    ea = None
    if poses is not None and dist:
        d32 = np.array(dist, dtype=np.float32)
        r32 = np.array(rot, dtype=np.float32)
        ea = np.stack([np.log1p(d32) / 5.0, r32 / np.pi], axis=1).astype(np.float32)   # :583-596
    return ei, ea


# ---------------------------------------------------------------------------------------------
# voxel IoU (data/pose_utils.py:323-389 after the sampling step) -- C oracle
# ---------------------------------------------------------------------------------------------
def voxel_overlap(points1, points2, T_12, voxel_size=0.2):
    """Returns (iou, [n_unique1, n_unique2, n_intersection])."""
    p1 = np.ascontiguousarray(points1, dtype=np.float32)
    p2 = np.ascontiguousarray(points2, dtype=np.float32)
    assert p1.shape[1] == p2.shape[1] and p1.shape[1] in (3, 4)
    T = np.ascontiguousarray(T_12, dtype=np.float64)
    counts = np.zeros(3, dtype=np.int32)
    fp = C.POINTER(C.c_float)
    iou = nsc_oracle.lib().nsc_oracle_voxel_overlap(
        p1.ctypes.data_as(fp), len(p1), p2.ctypes.data_as(fp), len(p2), p1.shape[1],
        T.ctypes.data_as(C.POINTER(C.c_double)), float(voxel_size),
        counts.ctypes.data_as(C.POINTER(C.c_int32)))
    return float(iou), counts
