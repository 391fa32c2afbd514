#!/usr/bin/env python3
"""tests/golden/intensity.npz from the reference's RangeImageProjector.project(keep_intensity=True)
(build container only: imports /root/reference/src, which does not travel)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from encoding.range_image import RangeImageProjector            # noqa: E402
import nsc_oracle                                                # noqa: E402

rng = np.random.default_rng(11)
out = {}


def cloud(n, seed):
    r = np.random.default_rng(seed)
    az = r.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(r.uniform(-26, 3, n))
    rr = r.uniform(0.5, 90, n)
    p = np.stack([rr * np.cos(el) * np.cos(az), rr * np.cos(el) * np.sin(az), rr * np.sin(el),
                  r.uniform(-0.2, 1.0, n)], 1).astype(np.float32)       # some intensities <= 0
    return p


cases = []
seed = 100
while len(cases) < 4:                                 # clouds on which oracle and reference agree on every pixel
    n = [20000, 8000, 25000, 12000][len(cases)]
    p = cloud(n, seed)
    seed += 1
    if len(cases) == 3:                               # NaN / +inf intensities: np.maximum.at propagates NaN (:225)
        p[::37, 3] = np.nan                           # some of these are a pixel's closest point, some are not
        p[5::101, 3] = np.inf
        p[2000:2300, :3] = p[:300, :3]                # exact range ties: NaN next to a number on the same pixel
        p[2000:2300:2, 3] = np.nan
    if len(cases) == 1:                               # exact range ties inside a pixel: duplicated points, different intensity
        p[1000:2000, :3] = p[:1000, :3]
        p[1000:2000, 3] = rng.uniform(0, 2, 1000).astype(np.float32)
        p[5, 3] = 0.0
    proj = RangeImageProjector(n_elevation=16, n_azimuth=360)
    rimg, iimg = proj.project(p, keep_intensity=True)
    oimg, ointen = nsc_oracle.project_intensity(p)
    if not (oimg.view(np.uint32) == rimg.astype(np.float32).view(np.uint32)).all():
        continue                                      # an ULP-edge point: the atan2 definition differs
    assert (ointen.view(np.uint32) == iimg.astype(np.float32).view(np.uint32)).all(), "oracle intensity != reference"
    k = f"c{len(cases)}"
    out[k + "_pts"], out[k + "_range"], out[k + "_intensity"] = p, rimg.astype(np.float32), iimg.astype(np.float32)
    cases.append(k)
    print(k, n, "seed", seed - 1, "pixels with intensity", int((iimg > 0).sum()), "NaN pixels", int(np.isnan(iimg).sum()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "intensity.npz"), **out)
