#!/usr/bin/env python3
"""tests/golden/keyframe.npz from the reference's own quantizer, record serialiser and voxel-IoU
(build container only: imports /root/reference/src, which does not travel)."""
import math
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
from encoding.quantization import CompressedDescriptor, HistogramQuantizer   # noqa: E402
from data.pose_utils import compute_overlap, transform_points                 # noqa: E402

rng = np.random.default_rng(7)
out = {}

# ---- quantizer: (n_bins,) float32 -> uint16 -> float32 -------------------------------------
def quant_cases(nb):
    c = []
    for _ in range(24):
        h = (rng.random(nb) ** 4).astype(np.float32)
        c.append(h / h.sum())
    c.append(np.full(nb, 1.0 / nb, dtype=np.float32))                 # every bin rounds the same way
    c.append(np.zeros(nb, dtype=np.float32))                          # empty -> all zero / uniform
    z = np.zeros(nb, dtype=np.float32); z[nb // 3] = 1.0; c.append(z)  # one spike
    z = np.zeros(nb, dtype=np.float32); z[:3] = [0.5, 0.25, 0.25]; c.append(z)
    c.append((rng.random(nb) * 3).astype(np.float32))                 # not normalised
    c.append((rng.random(nb) * 1e-12).astype(np.float32))             # sum below epsilon
    h = np.zeros(nb, dtype=np.float32); h[::2] = 1.0; c.append(h)      # ties for argmax
    return np.stack(c)

for nb in (50, 800):
    q = HistogramQuantizer(n_bins=nb)
    hs = quant_cases(nb)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                                # uint64 wrap-around in quantization.py:157-165
        qs = np.stack([q.quantize(h.copy()) for h in hs])
    ds = np.stack([q.dequantize(x) for x in qs])
    # dequantize on arbitrary uint16 rows too (sums above 2^24: the float32 order matters)
    arb = rng.integers(0, 65536, (8, nb)).astype(np.uint16)
    out[f"q{nb}_hist"], out[f"q{nb}_quant"], out[f"q{nb}_deq"] = hs, qs, ds
    out[f"q{nb}_arb"], out[f"q{nb}_arb_deq"] = arb, np.stack([q.dequantize(x) for x in arb])

# ---- 220-byte records (n_bins = 50 only: CompressedDescriptor.to_bytes asserts the length) -
recs, poses7, tss, ids, hashes = [], [], [], [], []
for i in range(6):
    pose7 = rng.normal(size=7)
    ts = float(rng.random() * 1e9)
    kid = int(rng.integers(0, 2 ** 32))
    hsh = rng.integers(0, 256, 20).astype(np.uint8).tobytes()
    d = CompressedDescriptor(histogram=out["q50_quant"][i], pose=pose7, timestamp=ts, keyframe_id=kid,
                             point_cloud_hash=hsh)
    b = d.to_bytes()
    back = CompressedDescriptor.from_bytes(b)
    assert back.keyframe_id == kid and back.timestamp == ts
    recs.append(np.frombuffer(b, dtype=np.uint8))
    poses7.append(pose7); tss.append(ts); ids.append(kid); hashes.append(np.frombuffer(hsh, dtype=np.uint8))
out["rec_bytes"] = np.stack(recs)
out["rec_pose7"], out["rec_ts"] = np.stack(poses7), np.array(tss)
out["rec_id"], out["rec_hash"] = np.array(ids, dtype=np.uint32), np.stack(hashes)

# ---- voxel IoU (clouds <= max_points so the unseeded sampling never runs) ------------------
def rand_T(scale_t):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    th = rng.normal() * 0.3
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K
    T[:3, 3] = rng.normal(size=3) * scale_t
    return T

def scene(n, cols, spread):
    """Ground plane + two walls (dense enough that 0.2 m voxels are shared between two samplings)."""
    p = np.zeros((n, cols), dtype=np.float64)
    k = n // 2
    p[:k, 0], p[:k, 1] = rng.uniform(-spread, spread, k), rng.uniform(-spread, spread, k)
    p[:k, 2] = -1.7 + rng.normal(0, 0.02, k)
    m = n - k
    p[k:, 0] = np.where(rng.random(m) < 0.5, -spread, spread) + rng.normal(0, 0.03, m)
    p[k:, 1], p[k:, 2] = rng.uniform(-spread, spread, m), rng.uniform(-1.7, 1.0, m)
    if cols == 4:
        p[:, 3] = rng.random(n)
    return p.astype(np.float32)

ov = []
base = scene(4000, 3, 6.0)
T = rand_T(0.5)
inv = np.linalg.inv(T)
moved = transform_points(base.astype(np.float64), inv).astype(np.float32)          # cloud 1 such that T maps it onto base
ov.append((moved, base, T))                                                          # high overlap
ov.append((scene(5000, 3, 7.0), scene(5000, 3, 7.0), rand_T(0.05)))                  # partial overlap
ov.append((scene(3000, 4, 4.0), scene(2500, 4, 4.0), rand_T(0.02)))                   # intensity column rides along
p = scene(1500, 3, 3.0); p[::97, 1] = np.nan; p[5::131, 0] = np.inf
q = scene(1400, 3, 3.0); q[::89, 2] = -np.inf
ov.append((p, q, rand_T(0.01)))                                                       # non-finite rows dropped
p = scene(800, 4, 2.0); p[::50, 3] = np.nan                                          # NaN intensity drops the row too
ov.append((p, scene(900, 4, 2.0), np.eye(4)))
p = (rng.normal(size=(600, 3)) * 3.0e6).astype(np.float32)                                                            # clip at +-1e6
ov.append((p, p.copy(), np.eye(4)))
ov.append((np.zeros((0, 3), np.float32), scene(10, 3, 1.0), np.eye(4)))              # empty cloud
ov.append((np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.eye(4)))   # both empty -> 0.0
g = (np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(3)), -1).reshape(-1, 3) * 0.2).astype(np.float32)
ov.append((g, g + np.float32(0.2), np.eye(4)))                                       # points ON voxel faces (f32 vs f64 division)
ov.append((scene(5000, 3, 10.0), scene(5000, 3, 10.0), rand_T(0.5)))
for vs_i, (p1, p2, T) in enumerate(ov):
    for voxel in ((0.2, 0.5) if vs_i < 3 else (0.2,)):
        k = f"ov{vs_i}_{int(voxel * 10)}"
        out[k + "_p1"], out[k + "_p2"], out[k + "_T"] = p1, p2, T
        out[k + "_voxel"] = np.float64(voxel)
        out[k + "_iou"] = np.float64(compute_overlap(p1, p2, T, voxel_size=voxel))

# the dgemm accumulation order the oracle assumes (fma chain), checked here against numpy itself
import ctypes                                                                        # noqa: E402
libm = ctypes.CDLL("libm.so.6"); libm.fma.restype = ctypes.c_double; libm.fma.argtypes = [ctypes.c_double] * 3
p1, _, T = ov[1]
ref = transform_points(p1, T)
bad = 0
for i in range(len(p1)):
    x, y, z = (float(v) for v in p1[i])
    for c in range(3):
        v = libm.fma(T[c, 3], 1.0, libm.fma(T[c, 2], z, libm.fma(T[c, 1], y, T[c, 0] * x)))
        bad += v != ref[i, c]
print("fma-chain vs numpy matmul mismatches:", bad, "of", 3 * len(p1))
assert bad == 0

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "keyframe.npz"), **out)
print({k: v.shape for k, v in out.items() if k.startswith(("q", "rec"))})
print({k: float(v) for k, v in out.items() if k.endswith("_iou")})
