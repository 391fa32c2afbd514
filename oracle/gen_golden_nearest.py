#!/usr/bin/env python3
"""tests/golden/interp_nearest.npz: interpolate_range_image(img, method='nearest') of the reference
(src/encoding/range_image.py:66-87), imported in the build container.  Inputs + expected outputs only."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src")
from encoding.range_image import interpolate_range_image          # noqa: E402

imgs = []
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "enc_*.npz"))):
    raw = np.load(f)["ref_raw"].view(np.float32)
    if raw.shape == (16, 360):
        imgs.append(raw)
rng = np.random.default_rng(5)
for density in (0.5, 0.1, 0.02):                                   # random holes, incl. long gaps and exact ties
    im = rng.uniform(1, 80, (16, 360)).astype(np.float32)
    im[rng.random((16, 360)) > density] = 0.0
    im[3] = 0.0                                                    # an empty row (copied from a neighbour, :77-87)
    imgs.append(im)
tie = np.zeros((16, 360), np.float32)                              # hand-made ties: equidistant left / right, also across the wrap
tie[0, [10, 20]] = [5.0, 7.0]                                      # column 15 is 5 from both
tie[1, [3, 359]] = [2.0, 9.0]                                      # column 1 is 2 from both (wrap)
tie[2, [0]] = [4.0]                                                # a single valid pixel
tie[5, [100, 102, 104]] = [1.5, 2.5, 3.5]
imgs.append(tie)
imgs = np.stack(imgs)
out = np.stack([interpolate_range_image(im, "nearest") for im in imgs])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "interp_nearest.npz"), raw=imgs.view(np.uint32),
                    nearest=out.view(np.uint32))
print(imgs.shape, (imgs == 0).mean(), (out == 0).mean())
