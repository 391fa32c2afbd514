#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

The reference (/root/reference, read-only, Python) is importable here; it never travels to the
GPU box, so its outputs on fixed inputs are committed as small fixtures (inputs + expected
outputs -- data, no reference source).  Run from the repo root:

    python oracle/gen_golden.py

Each encoder case stores
  points      (N,3|4) float32   input cloud, verbatim (guards generator drift)
  ref_raw     (E,360) uint32    bit pattern of RangeImageProjector.project()      range_image.py:129
  ref_interp  (E,360) uint32    bit pattern of interpolate_range_image()          range_image.py:15
  ref_desc    (800,)  float32   SpectralEncoder.encode_points()                   spectral_encoder.py:206
  np_idx      (N,)    int32     per-point row*360+col as numpy computes it (numpy's own float32
                                arctan2; -1 = dropped) -- for itemising atan2-ULP edge points
  edge_pts    (K,)    int64     points where the oracle's correctly rounded atan2 lands in a
                                different pixel than numpy's SVML arctan2 (K is 0..3 per cloud)
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference/src")

import nsc_oracle as orc                                  # noqa: E402
from neural_spectral_codec_amd import synth               # noqa: E402
from encoding.range_image import RangeImageProjector, interpolate_range_image   # noqa: E402
from encoding.spectral_encoder import SpectralEncoder     # noqa: E402

torch.set_num_threads(1)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def numpy_linear_idx(points, E=16, A=360, elev=(-24.8, 2.0), rmin=1.0, rmax=80.0):
    """Per-point pixel index with numpy's own ufuncs (what project() scatters with)."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    ok = np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    xs, ys, zs = (np.clip(v ** 2, 0, 1e10) for v in (x, y, z))
    with np.errstate(invalid="ignore"):
        r = np.sqrt(xs + ys + zs)
        az = (np.arctan2(y, x) + np.pi) % (2 * np.pi)
        el = np.arctan2(z, np.sqrt(xs + ys))
        ok &= (r >= rmin) & (r <= rmax) & np.isfinite(r)
        lo, hi = np.deg2rad(elev[0]), np.deg2rad(elev[1])
        row = np.clip(np.floor(np.nan_to_num((el - lo) / (hi - lo) * E)).astype(int), 0, E - 1)
        col = np.clip(np.floor(np.nan_to_num(az / (2 * np.pi) * A)).astype(int), 0, A - 1)
    return np.where(ok, row * A + col, -1).astype(np.int32)


def encoder_case(name, points, n_elevation=16, elevation_range=(-24.8, 2.0)):
    enc = SpectralEncoder(n_elevation=n_elevation, n_azimuth=360, n_bins=50, alpha=2.0,
                          learnable_alpha=True, target_elevation_bins=16,
                          elevation_range=elevation_range)
    proj = enc.projector
    raw, _ = proj.project(points, keep_intensity=False)
    raw = raw.copy()
    itp = interpolate_range_image(raw, method="linear")
    desc = enc.encode_points(points).detach().cpu().numpy()
    np_idx = numpy_linear_idx(points, n_elevation, 360, elevation_range)

    p = orc.default_params(n_elevation=n_elevation, elevation_range=elevation_range)
    o_raw, o_idx, _ = orc.project(points, p, want_idx=True)
    edge = np.nonzero(o_idx != np_idx)[0].astype(np.int64)
    n_pix_diff = int((o_raw.view(np.uint32) != raw.view(np.uint32)).sum())
    print(f"{name:14s} N={len(points):7d} E={n_elevation:2d} edge_pts={len(edge)} "
          f"pixel_diffs_vs_oracle={n_pix_diff} sum={desc.sum():.8f} "
          f"sha={hashlib.sha256(points.tobytes()).hexdigest()[:12]}")
    np.savez_compressed(
        os.path.join(OUT, f"enc_{name}.npz"), points=points,
        ref_raw=raw.view(np.uint32), ref_interp=itp.view(np.uint32), ref_desc=desc,
        np_idx=np_idx, edge_pts=edge, n_elevation=np.int32(n_elevation),
        elevation_range=np.asarray(elevation_range, np.float64))


def main():
    encoder_case("uniform20k", synth.make_cloud(1, 20000, "uniform"))
    encoder_case("safe20k", synth.make_cloud(2, 20000, "safe"))
    encoder_case("ring32k", synth.make_cloud(3, 32000, "ring"))
    encoder_case("sparse8k", synth.make_cloud(4, 8000, "sparse"))
    encoder_case("empty", np.zeros((0, 4), np.float32))
    encoder_case("adversarial", synth.make_cloud(5, 20000, "adversarial"))
    encoder_case("wide20k", synth.make_cloud(6, 20000, "wide"))
    encoder_case("xyz_only", synth.make_cloud(7, 12000, "uniform")[:, :3].copy())
    encoder_case("single_pt", np.array([[10.0, 3.0, -1.0, 0.5]], np.float32))
    encoder_case("uniform120k", synth.make_cloud(8, 120000, "uniform"))
    encoder_case("e64_ring", synth.make_cloud(9, 32000, "ring"), n_elevation=64)
    encoder_case("vlp_fov", synth.make_cloud(10, 16000, "sparse"), elevation_range=(-15.0, 15.0))

    # encode_range_image / forward on images directly (spectral_encoder.py:160,231), incl. the
    # adaptive_avg_pool2d branch for 64-row input (:171-176)
    rng = np.random.default_rng(11)
    enc = SpectralEncoder(n_elevation=16, n_azimuth=360, n_bins=50, alpha=2.0,
                          target_elevation_bins=16)
    imgs16 = rng.uniform(0.0, 80.0, (6, 16, 360)).astype(np.float32)
    imgs16[1] *= (rng.uniform(0, 1, (16, 360)) > 0.3)           # holes, NOT interpolated by forward()
    imgs16[2] = 0.0                                              # all-empty -> uniform 1/800
    imgs16[3] = 37.5                                             # constant -> DC only
    imgs16[4, :, :] = (20 + 10 * np.sin(np.arange(360) * 2 * np.pi * 7 / 360))[None, :]
    with torch.no_grad():
        d16 = enc.forward(torch.from_numpy(imgs16)).numpy()
    imgs64 = rng.uniform(0.0, 80.0, (2, 64, 360)).astype(np.float32)
    with torch.no_grad():
        d64 = enc.forward(torch.from_numpy(imgs64)).numpy()
        edges = enc._compute_bin_edges(enc.alpha).numpy()
    np.savez_compressed(os.path.join(OUT, "range_images.npz"), imgs16=imgs16, desc16=d16,
                        imgs64=imgs64, desc64=d64)

    # bin edges + 181->50 LUT for several alpha (spectral_encoder.py:93-116,136-145)
    luts = {}
    for alpha in (0.5, 1.0, 2.0, 3.0):
        e = SpectralEncoder(n_elevation=16, n_bins=50, alpha=alpha)
        with torch.no_grad():
            ed = e._compute_bin_edges(e.alpha)
            fi = torch.arange(e.n_freqs, dtype=torch.float32)
            lut = torch.clamp(torch.searchsorted(ed, fi, right=True) - 1, 0, e.n_bins - 1)
        luts[f"edges_{alpha}"] = ed.numpy()
        luts[f"lut_{alpha}"] = lut.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "bin_lut.npz"), **luts)
    print("edges[:4]", edges[:4], "lut sizes", np.bincount(luts["lut_2.0"]).tolist())


if __name__ == "__main__":
    main()
