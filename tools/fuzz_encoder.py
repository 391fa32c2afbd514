#!/usr/bin/env python3
"""One-off soak (not part of the test suite): many full-size random clouds through nsc_encode_clouds vs the C
oracle -- raw and interpolated range images must be bit-identical, descriptors within 1e-6 relative.
usage: fuzz_encoder.py [n_rounds] [clouds_per_round] [points_per_cloud] [config: 0-5 = the fixed sets below, -SEED = a random one]"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "oracle"))
import nsc_oracle                                                               # noqa: E402
from neural_spectral_codec_amd import synth                                    # noqa: E402
from neural_spectral_codec_amd.encoding import SpectralEncoder                 # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ncl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 120000
threads = min(os.cpu_count() or 1, 128)
# parameter sets: the path's shape, the 64-row projector pooled to 16 rows, the float32 row chain of numpy 1.24,
# a wide FOV (no narrow-FOV shortcut), another alpha / bin count, (N,3) input
CFG = [dict(),
       dict(n_elevation=64, target_elevation_bins=16),
       dict(elev_float64=False),
       dict(elevation_range=(-40.0, 35.0)),
       dict(n_bins=37, alpha=1.3),
       dict(xyz_only=True)]
cfg_id = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if cfg_id < 0:
    # a RANDOM parameter set from the range check_params accepts (seed = -cfg_id): projector rows 1-64, target rows 1-16
    # (<= projector rows), 1-176 bins, any alpha, any field of view, either row arithmetic, (N,3) or (N,4) points
    prng = np.random.default_rng(-cfg_id)
    E_ = int(prng.integers(1, 65))
    lo_ = float(prng.uniform(-60, 10))
    CFG.append(dict(n_elevation=E_, target_elevation_bins=int(prng.integers(1, min(E_, 16) + 1)), n_bins=int(prng.integers(1, 177)),
                    alpha=float(prng.uniform(0.3, 4.0)), elevation_range=(lo_, lo_ + float(prng.uniform(2, 80))),
                    elev_float64=bool(prng.integers(0, 2)), xyz_only=bool(prng.integers(0, 2))))
    cfg_id = len(CFG) - 1
kw = dict(CFG[cfg_id])
xyz_only = kw.pop("xyz_only", False)
kw.setdefault("n_elevation", 16)
enc = SpectralEncoder(**kw).to("cuda")
op = nsc_oracle.default_params(n_elevation=enc.n_elevation, target_rows=enc.target_elevation_bins, n_bins=enc.n_bins,
                               elev_f64=int(kw.get("elev_float64", True)))
if "elevation_range" in kw:
    op.elev_min_rad, op.elev_max_rad = np.deg2rad(kw["elevation_range"][0]), np.deg2rad(kw["elevation_range"][1])
olut = nsc_oracle.bin_lut(float(kw.get("alpha", 2.0)), enc.n_bins, 181, 1e-8)[1]
print("config", cfg_id, CFG[cfg_id], flush=True)
tot_pts = bad_raw = bad_itp = 0
worst = 0.0
t0 = time.time()
for rd in range(rounds):
    pts, off = synth.make_clouds_device(ncl, npts, "cuda", seed=9000 + rd)
    if rd % 3 == 1:                                   # lace with non-finite values and out-of-range points
        idx = torch.randint(0, pts.shape[0], (pts.shape[0] // 997,), device="cuda")
        pts[idx, rd % 3] = float("nan")
        idx = torch.randint(0, pts.shape[0], (pts.shape[0] // 1009,), device="cuda")
        pts[idx, 0] = float("inf")
    if rd % 3 == 2:                                   # sparse rows: squeeze elevation so that rows stay empty
        pts[:, 2] *= 0.2
    if xyz_only:
        pts = pts[:, :3].contiguous()
    desc, raw, itp = enc.encode_points_batch((pts, off), return_images=True)
    hp, ho = pts.cpu().numpy(), off.cpu().numpy()
    od, oraw, oitp = nsc_oracle.encode_clouds(hp, ho, p=op, lut=olut, n_threads=threads, want_images=True)
    raw, itp, desc = raw.cpu().numpy(), itp.cpu().numpy(), desc.cpu().numpy()
    bad_raw += int((raw.view(np.uint32) != oraw.view(np.uint32)).sum())
    bad_itp += int((itp.view(np.uint32) != oitp.view(np.uint32)).sum())
    worst = max(worst, float((np.abs(desc - od) / (np.abs(od) + 1e-9)).max()))
    tot_pts += pts.shape[0]
    print(f"round {rd}: {ncl} clouds, raw mismatches {bad_raw}, interpolated mismatches {bad_itp}, "
          f"worst descriptor rel err {worst:.2e}  ({time.time() - t0:.0f} s)", flush=True)
print(f"TOTAL {tot_pts} points, {rounds * ncl} clouds: raw pixel mismatches {bad_raw}, interpolated {bad_itp}, "
      f"descriptor rel err {worst:.2e}")
sys.exit(1 if (bad_raw or bad_itp or worst > 1e-5) else 0)
